"""``dvmvs.utils`` -- the reference's geometric hot-path functions, executed by hand-written gfx950 HIP kernels.

Function names, argument order and return shapes are those of /root/reference/dvmvs/utils.py so the reference's
run-testing / run-training scripts can import this module unchanged.  The bodies are not translations: each of
the hot functions is one C-ABI call into ``libdvmvs_hip.so`` (see ``include/dvmvs_hip.h``):

===================================================  ==========================================================
reference function (utils.py)                        what runs here
===================================================  ==========================================================
calculate_cost_volume_by_warping  :45-86             dvmvs_cost_volume_fwd, M = 1 (all planes, one launch)
cost_volume_fusion                :89-107            dvmvs_cost_volume_fwd, all M frames fused, written once
get_non_differentiable_rectangle_depth_estimation    dvmvs_depth_reproject_fwd (atomic z-buffer; no sort, no
                                  :110-154           host round trip)
warp_frame_depth                  :205-258           dvmvs_hidden_warp_fwd
===================================================  ==========================================================

The few 3x3 / 4x4 matrices these functions derive from the poses (utils.py:51-56, :121) are evaluated by
``dvmvs.pose_algebra`` -- by default with the reference's own fp32 expressions, so that the kernels sample at the
reference's positions bit for bit -- and handed to the kernels as device arrays.

There is no CPU implementation in this package; CPU tensors raise.  ``cv2``/``kornia``/``path``/``pytorch3d`` are
not imported.
"""
import os
import time
import zipfile

import numpy as np
import torch

from dvmvs.hip import ops as _ops
from dvmvs import pose_algebra as _pose_algebra

# kernel selector for the cost volume (include/dvmvs_hip.h): 0 = automatic (generic kernel for small maps / SAD, otherwise the
# LDS-tiled sweep in the configuration the host-side plan model picks for the keyframe geometry), 1 = generic reference-order
# kernel, 2 = LDS-tiled sweep, default configuration, 3 = LDS-tiled sweep, wide-baseline configuration, 4 / 5 = 2 / 3 without a second pass
COST_VOLUME_VARIANT = int(os.environ.get("DVMVS_COST_VOLUME_VARIANT", "0"))
# host-planned work list for the tiled sweep (dvmvs_sweep_work_list); DVMVS_SWEEP_WORK_LIST=0: the static (tile, chunk) numbering
SWEEP_WORK_LIST = os.environ.get("DVMVS_SWEEP_WORK_LIST", "1") != "0"


def sweep_variant(host_matrices, height, width, n_depth_levels, min_depth, max_depth, dot_product=True):
    """The ``variant`` argument of dvmvs_cost_volume_fwd for one call: the forced one (DVMVS_COST_VOLUME_VARIANT), or for the
    dot-product sweep of a map the tiled kernel handles, the configuration picked from the HOST copies of the matrices."""
    if COST_VOLUME_VARIANT != 0 or host_matrices is None or not dot_product or height * width < 64 * 64:
        return COST_VOLUME_VARIANT
    return _pose_algebra.sweep_variant_host(host_matrices[0], host_matrices[1], height, width, n_depth_levels, min_depth, max_depth)


# ----------------------------------------------------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------------------------------------------------
def pose_distance(reference_pose, measurement_pose):
    """Combined / rotational / translational distance between two camera-to-world poses (4x4 numpy).

    Keyframe-selection measure of /root/reference/dvmvs/utils.py:17-31.
    """
    relative = np.linalg.inv(reference_pose) @ measurement_pose
    rotation, translation = relative[:3, :3], relative[:3, 3]
    r_measure = np.sqrt(2 * (1 - min(3.0, float(np.trace(rotation))) / 3))
    t_measure = np.linalg.norm(translation)
    return np.sqrt(t_measure ** 2 + r_measure ** 2), r_measure, t_measure


def is_pose_available(pose):
    return bool(np.all(np.isfinite(pose)))


def get_warp_grid_for_cost_volume_calculation(width, height, device):
    """[3, H*W] homogeneous pixel grid (x, y, 1), row-major.

    Kept for call-site compatibility (/root/reference/dvmvs/utils.py:34-42).  The HIP kernel derives pixel
    coordinates from its thread index, so the tensor is only shape-checked by ``cost_volume_fusion``.
    """
    ys, xs = torch.meshgrid(torch.arange(int(height), dtype=torch.float32), torch.arange(int(width), dtype=torch.float32),
                            indexing="ij")
    grid = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(int(height) * int(width))], dim=0)
    return grid.to(device)


def _check_warp_grid(warp_grid, height, width):
    if warp_grid is not None and tuple(warp_grid.shape) != (3, height * width):
        raise ValueError(f"warp_grid must be [3, {height * width}] for a {width}x{height} feature map, "
                         f"got {tuple(warp_grid.shape)}")


_WORK_LIST_RINGS = {}      # (device, words, stream, thread) -> [position, [(pinned host buffer, device buffer, event), ...]]


def _upload_work_list(host, H, W, n_depth_levels, min_depth, max_depth, variant, device):
    """The tiled sweep's work list for these host matrices, on ``device``: planned straight into a pinned staging buffer and sent with an
    asynchronous, stream-ordered copy (a pageable ``.to(device)`` per call was a device synchronisation inside every training step: ADVICE
    r4).  A small ring of (pinned, device) buffer pairs per device; a pair is reused only after the copy AND the sweep launch that read it
    have executed (its event is recorded by the next call on the same stream, i.e. behind that launch)."""
    words = _ops.sweep_work_list_words(host[0].shape[0], H, W, n_depth_levels)
    # One ring per (stream, thread): a slot's guard is recorded by the NEXT call, which is only "behind the sweep that read the slot" when both calls
    # enqueue on the same stream from the same thread (with DDP side streams or two threads a shared ring could be rewritten by the host
    # while its copy was still pending).
    import threading
    key = (str(device), words, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    ring = _WORK_LIST_RINGS.get(key)
    if ring is None:
        ring = _WORK_LIST_RINGS[key] = [0, [(torch.zeros(words, dtype=torch.int32).pin_memory(), torch.zeros(words, dtype=torch.int32, device=device),
                                             torch.cuda.Event()) for _ in range(4)]]
    position, slots = ring
    # the slot used by the PREVIOUS call gets its guard now: everything that call enqueued (copy + sweep) is in front of this record
    previous = slots[(position - 1) % len(slots)]
    with torch.cuda.device(device):
        previous[2].record(torch.cuda.current_stream(device))
    pinned, on_device, event = slots[position]
    ring[0] = (position + 1) % len(slots)
    event.synchronize()      # (three calls back: does not block in practice)
    _ops.sweep_work_list_host(host[0], host[1], H, W, n_depth_levels, min_depth, max_depth, variant, out=pinned)
    on_device.copy_(pinned, non_blocking=True)
    return on_device


def cost_volume_fusion(image1, image2s, pose1, pose2s, K, warp_grid, min_depth, max_depth, n_depth_levels, device,
                       dot_product):
    """Mean plane-sweep cost volume [B,D,H,W] of ``image1`` against every measurement frame in ``image2s``.

    One fused HIP launch for all frames and planes.  ``device`` is accepted for signature compatibility; the
    computation happens where the tensors live (which must be the GPU).
    """
    _check_warp_grid(warp_grid, image1.shape[2], image1.shape[3])
    image2s, pose2s = list(image2s), list(pose2s)
    if len(image2s) == 0 or len(image2s) != len(pose2s):
        raise ValueError("cost_volume_fusion: need as many measurement poses as measurement feature maps (>= 1)")
    Hm, kt, host = _pose_algebra.sweep_matrices(pose1, pose2s, K, image1.device, with_host=True)
    H, W = image1.shape[2], image1.shape[3]
    variant = sweep_variant(host, H, W, n_depth_levels, min_depth, max_depth, dot_product)
    work_list = None
    if host is not None and dot_product and variant in (0, 2, 3, 4, 5) and H * W >= 64 * 64 and SWEEP_WORK_LIST:
        # the tiled sweep's work list, planned on the host copies of the matrices (long workgroups cut into parallel pieces)
        work_list = _upload_work_list(host, H, W, n_depth_levels, min_depth, max_depth, variant, image1.device)
    return _ops.cost_volume(image1, image2s, Hm, kt, float(min_depth), float(max_depth), int(n_depth_levels), bool(dot_product), variant, work_list)


def cost_volume_from_matrices(image1, image2s, Hm, kt, host, min_depth, max_depth, n_depth_levels, dot_product=True):
    """``cost_volume_fusion`` for callers that already hold the sweep matrices (``Hm`` [B,M,9], ``kt`` [B,M,3] on the features' device,
    ``host`` = their host copies or None): a training step evaluates and uploads the matrices of its whole sub-sequence at once
    (dvmvs.training).  Same launch, same host-side choice of the sweep configuration and work list."""
    H, W = image1.shape[2], image1.shape[3]
    variant = sweep_variant(host, H, W, n_depth_levels, min_depth, max_depth, dot_product)
    work_list = None
    if host is not None and dot_product and variant in (0, 2, 3, 4, 5) and H * W >= 64 * 64 and SWEEP_WORK_LIST:
        work_list = _upload_work_list(host, H, W, n_depth_levels, min_depth, max_depth, variant, image1.device)
    return _ops.cost_volume(image1, list(image2s), Hm, kt, float(min_depth), float(max_depth), int(n_depth_levels), bool(dot_product), variant, work_list)


def calculate_cost_volume_by_warping(image1, image2, pose1, pose2, K, warp_grid, min_depth, max_depth, n_depth_levels,
                                     device, dot_product):
    """Cost volume of a single measurement frame (the M = 1 case of ``cost_volume_fusion``)."""
    return cost_volume_fusion(image1, [image2], pose1, [pose2], K, warp_grid, min_depth, max_depth, n_depth_levels,
                              device, dot_product)


def get_non_differentiable_rectangle_depth_estimation(reference_pose_torch, measurement_pose_torch, previous_depth_torch,
                                                      full_K_torch, half_K_torch, original_width, original_height):
    """Previous depth map splatted into the current view at half resolution, farthest surface wins, holes = 0."""
    B, _, H, W = previous_depth_torch.shape
    if (H, W) != (int(original_height), int(original_width)):
        raise ValueError(f"previous depth is {W}x{H} but original size was given as {original_width}x{original_height}")
    with torch.no_grad():
        transformation = _pose_algebra.relative_pose(reference_pose_torch, measurement_pose_torch, previous_depth_torch.device)
        return _ops.depth_reproject(transformation, previous_depth_torch, full_K_torch, half_K_torch)


def warp_frame_depth(image_src, depth_dst, src_trans_dst, camera_matrix, normalize_points=False, sampling_mode="bilinear"):
    """Warp ``image_src`` [B,C,H,W] into the destination view given the destination depth [B,1,H,W]."""
    for name, t in (("image_src", image_src), ("depth_dst", depth_dst), ("src_trans_dst", src_trans_dst),
                    ("camera_matrix", camera_matrix)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"Input {name} type is not a torch.Tensor. Got {type(t)}.")
    if image_src.dim() != 4:
        raise ValueError(f"Input image_src must have a shape (B, D, H, W). Got: {image_src.shape}")
    if depth_dst.dim() != 4 or depth_dst.shape[-3] != 1:
        raise ValueError(f"Input depth_dst must have a shape (B, 1, H, W). Got: {depth_dst.shape}")
    if src_trans_dst.dim() != 3 or tuple(src_trans_dst.shape[-2:]) != (4, 4):
        raise ValueError(f"Input src_trans_dst must have a shape (B, 4, 4). Got: {src_trans_dst.shape}.")
    if camera_matrix.dim() != 3 or tuple(camera_matrix.shape[-2:]) != (3, 3):
        raise ValueError(f"Input camera_matrix must have a shape (B, 3, 3). Got: {camera_matrix.shape}.")
    if normalize_points or sampling_mode != "bilinear":
        raise NotImplementedError("the HIP hidden-state warp implements normalize_points=False, sampling_mode='bilinear' "
                                  "(the only configuration the depth networks use)")
    return _ops.hidden_warp(image_src, depth_dst, src_trans_dst, camera_matrix, False)


# ----------------------------------------------------------------------------------------------------------------------
# training helpers
# ----------------------------------------------------------------------------------------------------------------------
def freeze_batchnorm(module):
    if isinstance(module, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
        module.eval()
        module.weight.requires_grad = False
        module.bias.requires_grad = False


def zip_code(run_directory):
    """Snapshot the *.py files of the working directory and its parent into ``run_directory/code.zip``."""
    with zipfile.ZipFile(os.path.join(run_directory, "code.zip"), "w", zipfile.ZIP_DEFLATED) as handle:
        for folder in ("./", "../"):
            for name in sorted(os.listdir(folder)):
                if name.endswith(".py"):
                    handle.write(os.path.join(folder, name))


def _checkpoint_name(prefix, filename, step, loss):
    return "{}_{}_epoch:{}_l1:{:.4f}_l1-inv:{:.4f}_l1-rel:{:.4f}_huber:{:.4f}".format(prefix, filename, step, *loss[:4])


def save_checkpoint(save_path, models, step, loss, filename="checkpoint.pth.tar"):
    for model in models:
        torch.save(model["state_dict"], os.path.join(str(save_path), _checkpoint_name(model["name"], filename, step, loss)))


def save_optimizer(save_path, optimizer, step, loss, filename="checkpoint.pth.tar"):
    torch.save(optimizer.state_dict(), os.path.join(str(save_path), _checkpoint_name("optimizer", filename, step, loss)))


def print_number_of_trainable_parameters(optimizer):
    count = sum(p.nelement() for group in optimizer.param_groups for p in group["params"] if p.requires_grad)
    print("Number of trainable parameters:", f"{count:,d}")


# ----------------------------------------------------------------------------------------------------------------------
# inference helpers
# ----------------------------------------------------------------------------------------------------------------------
def save_results(predictions, groundtruths, system_name, scene_name, save_folder, max_depth=np.inf):
    from dvmvs.errors import compute_errors
    if groundtruths is not None:
        errors = np.array([compute_errors(groundtruths[i], p, max_depth) for i, p in enumerate(predictions)])
        names = ["abs_error", "abs_relative_error", "abs_inverse_error", "squared_relative_error", "rmse", "ratio_125",
                 "ratio_125_2", "ratio_125_3"]
        print("Metrics of {} for scene {}:".format(system_name, scene_name))
        print(", ".join("{:>25}".format(n) for n in names))
        print(", ".join("{:25.4f}".format(v) for v in np.nanmean(errors, 0)))
        np.savez_compressed(os.path.join(str(save_folder), system_name + "_errors_" + scene_name), errors)
    save_predictions(predictions, system_name, scene_name, save_folder)


def save_predictions(predictions, system_name, scene_name, save_folder):
    np.savez_compressed(os.path.join(str(save_folder), system_name + "_predictions_" + scene_name), np.array(predictions))


def visualize_predictions(*args, **kwargs):
    raise NotImplementedError("interactive visualisation needs OpenCV, which is not part of this stack; "
                              "set Config.test_visualize = False")


class InferenceTimer:
    """HIP-event timer around the per-frame forward; same statistics as the reference (first ``n_skip`` dropped)."""

    def __init__(self, n_skip=20):
        self.times = []
        self.n_skip = n_skip
        # a stop-watch, not a compute path: without a GPU (the multi-process CPU tests drive the runners with a stub engine)
        # it measures host wall-clock milliseconds instead of HIP events
        self.on_gpu = torch.cuda.is_available()
        if self.on_gpu:
            self.forward_pass_start = torch.cuda.Event(enable_timing=True)
            self.forward_pass_end = torch.cuda.Event(enable_timing=True)

    def record_start_time(self):
        if self.on_gpu:
            self.forward_pass_start.record()
        else:
            self._t0 = time.perf_counter()

    def record_end_time_and_elapsed_time(self):
        if self.on_gpu:
            self.forward_pass_end.record()
            torch.cuda.synchronize()
            self.times.append(self.forward_pass_start.elapsed_time(self.forward_pass_end))
        else:
            self.times.append(1e3 * (time.perf_counter() - self._t0))

    def statistics(self):
        times = np.array(self.times[self.n_skip:])
        if len(times) == 0:
            return None
        return {"n": len(times), "mean": float(times.mean()), "std": float(times.std()), "median": float(np.median(times)),
                "min": float(times.min()), "max": float(times.max())}

    def print_statistics(self):
        stats = self.statistics()
        if stats is None:
            print("Not enough time measurements are taken!")
            return
        print("Number of Forward Passes:", stats["n"])
        for label, key in (("Mean", "mean"), ("Std", "std"), ("Median", "median"), ("Min", "min"), ("Max", "max")):
            print(f"--- {label} Inference Time:", stats[key])
