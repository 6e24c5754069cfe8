"""Hot-path operations of a frame executed by the HIP kernels through the C ABI, packaged for ``CpuDepthPipeline``
(oracle/fusionnet_cpu.py).  TEST INFRASTRUCTURE: the product never mixes CPU convolutions with GPU kernels; this exists so
that a test can hold the dense convolutions fixed (the very same CPU code in both pipelines, which is also what produced
the reference goldens) and attribute every difference in the final depth to the hand-written kernels alone.

The small pose algebra in front of the kernels is the product's (dvmvs.pose_algebra): ``mode`` None / "reference" = the
reference's fp32 expressions on the host, "exact" = fp64 on the device.

Reference call sites: /root/reference/dvmvs/fusionnet/run-testing.py:151-204, dvmvs/convlstm.py:26-59.
"""
import torch


class HipHotPath:
    def __init__(self, device, variant=0, mode=None):
        from dvmvs import pose_algebra
        from dvmvs.hip import ops
        self.ops, self.algebra, self.dev, self.variant, self.mode = ops, pose_algebra, torch.device(device), variant, mode
        self.calls = {"cost_volume": 0, "depth_reproject": 0, "hidden_warp": 0, "lstm_gates": 0}

    def _g(self, t):
        return t.to(self.dev).contiguous()

    def cost_volume_fusion(self, ref_half, meas_halves, pose, meas_poses, half_K, lo, hi, D):
        self.calls["cost_volume"] += 1
        Hm, kt = self.algebra.sweep_matrices(pose, meas_poses, half_K, self.dev, self.mode)
        out = self.ops.cost_volume(self._g(ref_half), [self._g(t) for t in meas_halves], Hm, kt, lo, hi, D, True, self.variant)
        return out.cpu()

    def depth_estimate(self, pose, previous_pose, previous_depth, full_K, half_K, width, height):
        self.calls["depth_reproject"] += 1
        T = self.algebra.relative_pose(pose, previous_pose, self.dev, self.mode)
        _, low = self.ops.depth_reproject_lowres(T, self._g(previous_depth), self._g(full_K), self._g(half_K), 16)
        return low.cpu()

    def warp_hidden(self, h, depth_estimate, previous_pose, pose, lstm_K):
        self.calls["hidden_warp"] += 1
        T = self.algebra.relative_pose(previous_pose, pose, self.dev, self.mode)
        return self.ops.hidden_warp(self._g(h), self._g(depth_estimate), T, self._g(lstm_K), True).cpu()

    def lstm_gates(self, combined_conv, c):
        self.calls["lstm_gates"] += 1
        h_next, c_next = self.ops.lstm_gates(self._g(combined_conv), self._g(c))
        return h_next.cpu(), c_next.cpu()
