"""Data-parallel fusionnet training step (BASELINE.json configs[4]): forward over an 8-frame sub-sequence, L1 loss on
inverse depth at five scales, backward through the HIP ops, one bucketed gradient all-reduce over RCCL/xGMI.

The forward follows /root/reference/dvmvs/fusionnet/run-training.py:184-284 (features of all frames first; frame i uses
frame i-1 as its single measurement frame; the hidden state is warped with the *ground-truth* depth, nearest /32; the
warp is executed for every step including the first) and the loss follows /root/reference/dvmvs/losses.py:48-80 with
``loss_type = "L1-inv"`` and unit weights.  Batch-norm statistics stay per GPU, exactly like the single-GPU reference at
the same per-device batch.

Communication design for MI355X: gradients live in a few large flat fp32 buckets (default 32 MB; parameters are 138.7 MB in
the last unfreezing stage) and ``param.grad`` tensors are views into them, so autograd accumulates straight into the
communication buffer; a bucket's all-reduce is launched (asynchronously, on RCCL's stream) from a post-accumulate hook
as soon as its last gradient lands, overlapping the rest of backward.  xGMI is point-to-point, so fewer, larger
collectives beat many small ones; nothing else on this path communicates.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from dvmvs import pose_algebra as _pose_algebra
from dvmvs.config import Config
from dvmvs.pose_algebra import to_host as _host
from dvmvs.utils import cost_volume_from_matrices, get_warp_grid_for_cost_volume_calculation


# ----------------------------------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------------------------------
def inverse_depth_l1(prediction, groundtruth):
    """Sum over valid pixels of |1/gt - 1/pred| and the number of valid pixels; ``groundtruth`` is nearest-resized to
    the prediction's resolution and pixels with gt == 0 are invalid (losses.py:48-80)."""
    b, h, w = prediction.shape
    gt = F.interpolate(groundtruth.view(b, 1, *groundtruth.shape[-2:]), size=(h, w), mode="nearest").view(b, h, w)
    valid = gt != 0
    count = valid.sum()
    diff = (1.0 / gt[valid] - 1.0 / prediction[valid]).abs().sum()
    return diff, count


def multi_scale_loss(predictions, groundtruth, weights=(1, 1, 1, 1, 1)):
    """sum_j w_j * L1-inv(prediction_j) / valid_j over the decoder's five outputs (coarse to fine or any order)."""
    total = 0.0
    for weight, prediction in zip(weights, predictions):
        s, n = inverse_depth_l1(prediction, groundtruth)
        total = total + weight * (s / n.clamp(min=1))
    return total


# ----------------------------------------------------------------------------------------------------------------------
# forward over one sub-sequence
# ----------------------------------------------------------------------------------------------------------------------
_STAGING = {}      # (device, floats, thread) -> [position, [(pinned buffer, event), ...]]


def subsequence_matrices(poses, half_K, device):
    """The small matrices of a whole sub-sequence -- per frame i >= 1 the sweep's K R K^-1 / K t against frame i - 1 (utils.py:51-56) and
    the ConvLSTM's inverse(pose[i-1]) @ pose[i] (convlstm.py:30) -- evaluated on the host with the reference's fp32 expressions
    (dvmvs.pose_algebra) and sent to ``device`` as ONE block through pinned memory (the per-frame path made ~20 small synchronous copies per
    step).  Returns per frame (Hm [B,1,9], kt [B,1,3], (host Hm, host kt), lstm_T [B,4,4]); entry 0 is None.  In "exact" mode (fp64 on
    the device) there is nothing to batch: the per-frame device ops are used."""
    n = len(poses)
    if _pose_algebra._resolve(None, list(poses) + [half_K]) != "reference":
        out = [None]
        for i in range(1, n):
            Hm, kt, host = _pose_algebra.sweep_matrices(poses[i], [poses[i - 1]], half_K, device, with_host=True)
            out.append((Hm, kt, host, _pose_algebra.relative_pose(poses[i - 1], poses[i], device)))
        return out
    poses = [_host(p) for p in poses]
    half_K = _host(half_K)
    B = poses[0].shape[0]
    host = [None]
    for i in range(1, n):
        Hm, kt = _pose_algebra.sweep_matrices_host(poses[i], [poses[i - 1]], half_K)
        host.append((Hm.contiguous().float(), kt.contiguous().float(), _pose_algebra.relative_pose_host(poses[i - 1], poses[i]).contiguous().float()))
    per_frame = B * (9 + 3 + 16)
    total = per_frame * (n - 1)
    import threading
    key = (str(device), total, threading.get_ident())      # (a ring per thread: the pinned buffers are rewritten by the host)
    ring = _STAGING.get(key)
    if ring is None:
        ring = _STAGING[key] = [0, [(torch.zeros(total, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(3)]]
    pinned, event = ring[1][ring[0]]
    ring[0] = (ring[0] + 1) % len(ring[1])
    event.synchronize()      # (the copy issued from this buffer three steps ago: does not block in practice)
    for i in range(1, n):
        o = (i - 1) * per_frame
        pinned[o:o + 9 * B].copy_(host[i][0].reshape(-1))
        pinned[o + 9 * B:o + 12 * B].copy_(host[i][1].reshape(-1))
        pinned[o + 12 * B:o + 28 * B].copy_(host[i][2].reshape(-1))
    block = torch.empty(total, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        block.copy_(pinned, non_blocking=True)
        event.record(torch.cuda.current_stream(device))
    out = [None]
    for i in range(1, n):
        o = (i - 1) * per_frame
        out.append((block[o:o + 9 * B].view(B, 1, 9), block[o + 9 * B:o + 12 * B].view(B, 1, 3), (host[i][0], host[i][1]),
                    block[o + 12 * B:o + 28 * B].view(B, 4, 4)))
    return out


def fusionnet_subsequence_loss(model, images, depths, poses, K, warp_grid=None):
    """``model`` = [feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion, cost_volume_decoder];
    ``images`` list of [B,3,H,W], ``depths`` list of [B,H,W], ``poses`` list of [B,4,4], ``K`` [B,3,3] (full resolution).
    Returns (loss summed over frames 1..n-1, list of full-resolution predictions)."""
    fe, fs, enc, lstm, dec = model
    B, _, H, W = images[0].shape
    # poses and the sweep's intrinsics are only read by the small pose algebra (dvmvs.pose_algebra), which by default runs on the
    # host: hold them there once per step instead of copying them back once per frame
    poses = [_host(p) for p in poses]
    half_K = _host(K).clone()
    half_K[:, 0:2, :] = half_K[:, 0:2, :] * 0.5
    lstm_K = K.to(images[0].device).clone()
    lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0
    if warp_grid is None:
        warp_grid = get_warp_grid_for_cost_volume_calculation(W // 2, H // 2, images[0].device)

    matrices = subsequence_matrices(poses, half_K, images[0].device)      # every frame's small matrices: one upload per step
    feats = [fs(*fe(img)) for img in images]
    loss = 0.0
    state = None
    full_predictions = []
    for i in range(1, len(images)):
        ref, meas = feats[i], feats[i - 1]
        Hm, kt, host, lstm_T = matrices[i]
        cost_volume = cost_volume_from_matrices(ref[0], [meas[0]], Hm, kt, host, Config.train_min_depth, Config.train_max_depth,
                                                Config.train_n_depth_levels, True)
        skip0, skip1, skip2, skip3, bottom = enc(ref[0], ref[1], ref[2], ref[3], cost_volume)
        depth_estimation = F.interpolate(depths[i].view(B, 1, H, W), scale_factor=1.0 / 32.0, mode="nearest")
        state = lstm(bottom, state, poses[i - 1], poses[i], depth_estimation, lstm_K, transformation=lstm_T)
        full, half, quarter, one_eight, one_sixteen = dec(images[i], skip0, skip1, skip2, skip3, state[0])
        loss = loss + multi_scale_loss([one_sixteen, one_eight, quarter, half, full], depths[i])
        full_predictions.append(full)
    return loss, full_predictions


class TrainingHyperparameters:
    """fusionnet/run-training.py:21-27."""
    Config.train_subsequence_length = Config.train_subsequence_length or 8
    batch_size = 4
    learning_rate = 1e-4
    momentum = 0.9
    beta = 0.999
    weight_decay = 0
    loss_type = "L1-inv"   # "L1", "L1-inv", "L1-rel" or "Huber"
    finetune_epochs = 1
    use_checkpoint = False


def forward_pass(images, depths, poses, K, model, is_training):
    """The fusionnet training scripts' ``forward_pass`` (run-training.py:184-284) with its exact return contract:
    (l1_meter, huber_meter, l1_inv_meter, l1_rel_meter, optimizer_loss, [quarter, half, full] predictions of the last frame,
    their names).  Tensors are moved to the current HIP device; the cost volume, hidden-state warp and gates run as HIP
    kernels with their backward kernels, the convolutions through MIOpen."""
    from dvmvs.losses import LossMeter, update_losses
    fe, fs, enc, lstm, dec = model
    device = next(fe.parameters()).device
    images = [t.to(device) for t in images]
    depths = [t.to(device) for t in depths]
    poses = [_host(t) for t in poses]          # read by the host-side pose algebra only (dvmvs.pose_algebra)
    B, _, H, W = images[0].shape
    half_K = _host(K).clone()
    half_K[:, 0:2, :] = half_K[:, 0:2, :] * 0.5
    lstm_K = K.to(device).clone()
    lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0
    warp_grid = get_warp_grid_for_cost_volume_calculation(W // 2, H // 2, device)
    matrices = subsequence_matrices(poses, half_K, device)
    feats = [fs(*fe(img)) for img in images]
    meters = [LossMeter() for _ in range(4)]
    l1_meter, huber_meter, l1_inv_meter, l1_rel_meter = meters
    optimizer_loss, predictions, state = 0, None, None
    for i in range(1, len(images)):
        ref, meas = feats[i], feats[i - 1]
        Hm, kt, host, lstm_T = matrices[i]
        cost_volume = cost_volume_from_matrices(ref[0], [meas[0]], Hm, kt, host, Config.train_min_depth, Config.train_max_depth,
                                                Config.train_n_depth_levels, True)
        skip0, skip1, skip2, skip3, bottom = enc(ref[0], ref[1], ref[2], ref[3], cost_volume)
        depth_estimation = F.interpolate(depths[i].view(B, 1, H, W), scale_factor=1.0 / 32.0, mode="nearest")
        state = lstm(bottom, state, poses[i - 1], poses[i], depth_estimation, lstm_K, transformation=lstm_T)
        full, half, quarter, one_eight, one_sixteen = dec(images[i], skip0, skip1, skip2, skip3, state[0])
        optimizer_loss = optimizer_loss + update_losses(predictions=[one_sixteen, one_eight, quarter, half, full], weights=[1, 1, 1, 1, 1],
                                                        groundtruth=depths[i], is_training=is_training, l1_meter=l1_meter,
                                                        huber_meter=huber_meter, l1_inv_meter=l1_inv_meter, l1_rel_meter=l1_rel_meter,
                                                        loss_type=TrainingHyperparameters.loss_type)
        predictions = [quarter, half, full]
    return l1_meter, huber_meter, l1_inv_meter, l1_rel_meter, optimizer_loss, predictions, ["prediction_quarter", "prediction_half", "prediction_full"]


# ----------------------------------------------------------------------------------------------------------------------
# bucketed gradient all-reduce
# ----------------------------------------------------------------------------------------------------------------------
class BucketedGradientReducer:
    """Averages gradients over the data-parallel group with a few large asynchronous all-reduces that overlap backward.

    Usage per step:  ``reducer.zero_grad()``; ``loss.backward()``; ``reducer.finish()``; ``optimizer.step()``.
    With no process group (single GPU) it only provides the flat, view-backed gradient storage.
    """

    def __init__(self, parameters, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        params = [p for p in parameters if p.requires_grad]
        # autograd produces gradients roughly in reverse parameter order: fill buckets in that order so that the first
        # bucket to complete is the first to be reduced
        params = list(reversed(params))
        self.buckets = []
        current, size = [], 0
        for p in params:
            current.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(current)
                current, size = [], 0
        if current:
            self.buckets.append(current)
        self.flat, self._pending, self._handles = [], [], []
        for b, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            flat = torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device)
            offset = 0
            for p in bucket:
                p.grad = flat[offset:offset + p.numel()].view_as(p)   # autograd accumulates in place into the bucket
                offset += p.numel()
                p.register_post_accumulate_grad_hook(self._make_hook(b))
            self.flat.append(flat)
            self._pending.append(len(bucket))
        self.launched_during_backward = 0

    def _make_hook(self, b):
        def hook(param):
            self._pending[b] -= 1
            if self._pending[b] == 0 and self.world > 1:
                self._handles.append(dist.all_reduce(self.flat[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.launched_during_backward += 1
        return hook

    def zero_grad(self):
        for b, flat in enumerate(self.flat):
            flat.zero_()
            self._pending[b] = len(self.buckets[b])
        self._handles = []
        self.launched_during_backward = 0

    def finish(self):
        """Waits for the in-flight all-reduces, reduces buckets whose parameters received no gradient this step, and
        turns the sums into means."""
        if self.world == 1:
            return
        for b, pending in enumerate(self._pending):
            if pending != 0:   # some parameter of the bucket was unused in this step: reduce it now (zeros contribute 0)
                self._handles.append(dist.all_reduce(self.flat[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self._handles:
            h.wait()
        for flat in self.flat:
            flat.div_(self.world)


def train_step(model, optimizer, reducer, images, depths, poses, K, warp_grid=None):
    """One optimisation step on this rank's batch of sub-sequences; returns the (local) loss value as a tensor."""
    reducer.zero_grad()
    loss, _ = fusionnet_subsequence_loss(model, images, depths, poses, K, warp_grid)
    loss.backward()
    reducer.finish()
    optimizer.step()
    return loss.detach()
