"""Hot-path operations of a frame executed by the HIP kernels through the C ABI, packaged for ``CpuDepthPipeline``
(oracle/fusionnet_cpu.py).  TEST INFRASTRUCTURE: the product never mixes CPU convolutions with GPU kernels; this exists so
that a test can hold the dense convolutions fixed (the very same CPU code in both pipelines, which is also what produced
the reference goldens) and attribute every difference in the final depth to the hand-written kernels alone.

Reference call sites: /root/reference/dvmvs/fusionnet/run-testing.py:151-204, dvmvs/convlstm.py:26-59.
"""
import torch


class HipHotPath:
    def __init__(self, device, variant=0):
        from dvmvs.hip import ops
        self.ops, self.dev, self.variant = ops, torch.device(device), variant
        self.calls = {"cost_volume": 0, "depth_reproject": 0, "hidden_warp": 0, "lstm_gates": 0}

    def _g(self, t):
        return t.to(self.dev).contiguous()

    def cost_volume_fusion(self, ref_half, meas_halves, pose, meas_poses, half_K, lo, hi, D):
        self.calls["cost_volume"] += 1
        out = self.ops.cost_volume(self._g(ref_half), [self._g(t) for t in meas_halves], self._g(pose), [self._g(p) for p in meas_poses],
                                   self._g(half_K), lo, hi, D, True, self.variant)
        return out.cpu()

    def depth_estimate(self, pose, previous_pose, previous_depth, full_K, half_K, width, height):
        self.calls["depth_reproject"] += 1
        _, low = self.ops.depth_reproject_lowres(self._g(pose), self._g(previous_pose), self._g(previous_depth), self._g(full_K),
                                                 self._g(half_K), 16)
        return low.cpu()

    def warp_hidden(self, h, depth_estimate, previous_pose, pose, lstm_K):
        self.calls["hidden_warp"] += 1
        T = self.ops.relative_pose(self._g(previous_pose), self._g(pose))
        return self.ops.hidden_warp(self._g(h), self._g(depth_estimate), T, self._g(lstm_K), True).cpu()

    def lstm_gates(self, combined_conv, c):
        self.calls["lstm_gates"] += 1
        h_next, c_next = self.ops.lstm_gates(self._g(combined_conv), self._g(c))
        return h_next.cpu(), c_next.cpu()
