"""CPU restatement of the reference's per-keyframe forward loop.  TEST INFRASTRUCTURE ONLY (see dvmvs_oracle.py).

Order of operations follows /root/reference/dvmvs/fusionnet/run-testing.py:151-204 (fusionnet) and
/root/reference/dvmvs/pairnet/run-testing.py:136-166 (pairnet, ``lstm_fusion=None``): measurement-frame features are
recomputed every frame exactly as the reference does, the hot-path functions are the oracle's.  Used by the GPU
end-to-end parity test and as the ``cpu_baseline`` ("port") leg of bench.py; never by the product package.
"""
import time

import torch

import dvmvs_oracle as orc


class OracleHotPath:
    """The four hot-path operations of a frame as the CPU oracle evaluates them.  ``CpuDepthPipeline`` takes any object with
    these methods: the hybrid parity test (tests/test_hybrid_parity.py) passes one whose methods run the HIP kernels through
    the C ABI instead, so that the dense convolutions of both pipelines are the SAME CPU code and every difference in the
    depth comes from the hot path alone."""

    def __init__(self, planewise=False):
        self.planewise = planewise

    def cost_volume_fusion(self, ref_half, meas_halves, pose, meas_poses, half_K, lo, hi, D):
        return orc.cost_volume_fusion(ref_half, meas_halves, pose, meas_poses, half_K, lo, hi, D, True, planewise=self.planewise)

    def depth_estimate(self, pose, previous_pose, previous_depth, full_K, half_K, width, height):
        return orc.nearest_downsample(orc.reproject_depth(pose, previous_pose, previous_depth, full_K, half_K, width, height), 16)

    def warp_hidden(self, h, depth_estimate, previous_pose, pose, lstm_K):
        return orc.warp_hidden_state(h, depth_estimate, orc.relative_pose(previous_pose, pose), lstm_K, zero_invalid=True)

    def lstm_gates(self, combined_conv, c):
        return orc.lstm_gates(combined_conv, c)


class CpuDepthPipeline:
    def __init__(self, feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion, cost_volume_decoder,
                 min_depth=0.25, max_depth=20.0, n_depth_levels=64, planewise_cost_volume=False, hot_path=None):
        self.fe, self.fs, self.enc, self.lstm, self.dec = (feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion,
                                                           cost_volume_decoder)
        for m in (self.fe, self.fs, self.enc, self.lstm, self.dec):
            if m is not None:
                m.eval()
        self.depth_range = (min_depth, max_depth, n_depth_levels)
        # bench.py: the per-plane grid_sample formulation of the cost volume is the fast one on CPUs
        self.hot = hot_path if hot_path is not None else OracleHotPath(planewise=planewise_cost_volume)
        self.stage_seconds = {}
        self.reset()

    def reset(self):
        self.lstm_state, self.previous_depth, self.previous_pose = None, None, None

    def _timed(self, name, fn):
        t0 = time.perf_counter()
        out = fn()
        self.stage_seconds[name] = self.stage_seconds.get(name, 0.0) + time.perf_counter() - t0
        return out

    @torch.no_grad()
    def step(self, reference_image, reference_pose, measurement_images, measurement_poses, full_K, record=None):
        H, W = reference_image.shape[-2:]
        half_K = full_K.clone()
        half_K[:, 0:2, :] = half_K[:, 0:2, :] / 2.0
        lstm_K = full_K.clone()
        lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0
        lo, hi, D = self.depth_range

        meas_half = self._timed("features", lambda: [self.fs(*self.fe(img))[0] for img in measurement_images])
        ref_feats = self._timed("features", lambda: self.fs(*self.fe(reference_image)))
        cv = self._timed("cost_volume", lambda: self.hot.cost_volume_fusion(ref_feats[0], meas_half, reference_pose, measurement_poses,
                                                                             half_K, lo, hi, D))
        skip0, skip1, skip2, skip3, bottom = self._timed("encoder", lambda: self.enc(*ref_feats, cv))
        de = None
        if self.lstm is not None:
            if self.previous_depth is not None:
                de = self._timed("reprojection", lambda: self.hot.depth_estimate(reference_pose, self.previous_pose, self.previous_depth,
                                                                                 full_K, half_K, W, H))
            else:
                de = torch.zeros(1, 1, H // 32, W // 32)
            h, c = self.lstm_state if self.lstm_state is not None else (torch.zeros_like(bottom), torch.zeros_like(bottom))
            weight = self.lstm.lstm_cell.conv.weight

            def cell():   # /root/reference/dvmvs/convlstm.py:26-59: warp + mask, 3x3 convolution without bias, gates
                hw = h if self.previous_pose is None else self.hot.warp_hidden(h, de, self.previous_pose, reference_pose, lstm_K)
                cc = torch.nn.functional.conv2d(torch.cat([bottom, hw], dim=1), weight, bias=None,
                                                padding=(weight.shape[2] // 2, weight.shape[3] // 2))
                return self.hot.lstm_gates(cc, c)

            self.lstm_state = self._timed("lstm", cell)
            bottom_out = self.lstm_state[0]
        else:
            bottom_out = bottom
        pred = self._timed("decoder", lambda: self.dec(reference_image, skip0, skip1, skip2, skip3, bottom_out)[0])
        if self.lstm is not None:
            self.previous_depth = pred.view(1, 1, H, W)
            self.previous_pose = reference_pose
        if record is not None:
            record(feat_half=ref_feats[0], cost_volume=cv, bottom=bottom, depth_estimation=de,
                   h=None if self.lstm is None else self.lstm_state[0], c=None if self.lstm is None else self.lstm_state[1], depth=pred)
        return pred


def cpu_subsequence_loss(model, images, depths, poses, K, min_depth=0.25, max_depth=20.0, n_depth_levels=64):
    """CPU restatement of the fusionnet TRAINING forward + L1-inv loss over one sub-sequence, differentiable through the
    oracle ops.  Order of operations: /root/reference/dvmvs/fusionnet/run-training.py:184-284; loss:
    /root/reference/dvmvs/losses.py:26-80 with loss_type "L1-inv", weights 1.  Checker for the GPU training step."""
    fe, fs, enc, lstm, dec = model
    B, _, H, W = images[0].shape
    half_K = K.clone()
    half_K[:, 0:2, :] = half_K[:, 0:2, :] * 0.5
    lstm_K = K.clone()
    lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0
    feats = [fs(*fe(img)) for img in images]
    total = 0.0
    state = None
    for i in range(1, len(images)):
        cv = orc.cost_volume(feats[i][0], feats[i - 1][0], poses[i], poses[i - 1], half_K, min_depth, max_depth, n_depth_levels, True)
        skip0, skip1, skip2, skip3, bottom = enc(*feats[i], cv)
        de = torch.nn.functional.interpolate(depths[i].view(B, 1, H, W), scale_factor=1.0 / 32.0, mode="nearest")
        h, c = state if state is not None else (torch.zeros_like(bottom), torch.zeros_like(bottom))
        state = orc.convlstm_cell(lstm.lstm_cell.conv.weight, bottom, h, c, poses[i - 1], poses[i], de, lstm_K)
        outs = dec(images[i], skip0, skip1, skip2, skip3, state[0])
        for pred in outs:
            b, hs, ws = pred.shape
            gt = torch.nn.functional.interpolate(depths[i].view(B, 1, H, W), size=(hs, ws), mode="nearest").view(b, hs, ws)
            valid = gt != 0
            total = total + (1.0 / gt[valid] - 1.0 / pred[valid]).abs().sum() / valid.sum()
    return total
