"""Per-wave timeline of the bottleneck convolution kernel (s_memrealtime brackets; needs `make -C deep-video-mvs_amd/csrc trace`).

    python tools/bottleneck_conv_trace.py

For the ConvLSTM layer (1024 -> 2048 channels, 8x10) and two 16x20 layers: the launch's span (first wave's start to last wave's end) and,
averaged over the waves (and for the last one to finish), where a wave spends its time: until its slice of x is staged, waiting for a group's
weights (the instrumented build waits for ALL outstanding loads at the head of every group: s_waitcnt vmcnt(0)), in the group's 180 MFMAs,
storing its partial sums; + how the waves are spread over the SIMDs and when each SIMD's last wave ends."""
import ctypes
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["DVMVS_HIP_LIB"] = os.environ.get("DVMVS_TRACE_LIB", os.path.join(ROOT, "deep-video-mvs_amd", "lib", "libdvmvs_hip_trace.so"))
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvmvs.hip import _capi, ops  # noqa: E402

LAYERS = [(1024, 2048, 8, 10, 1), (512, 256, 16, 20, 1), (512, 512, 8, 10, 1)]


def main():
    lib = _capi.lib()
    lib.dvmvs_debug_bottleneck_conv_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for C_in, C_out, H, W, stride in LAYERS:
        x = torch.randn(1, C_in, H, W, generator=g).to(dev)
        w = (torch.randn(C_out, C_in, 3, 3, generator=g) / (3 * C_in ** 0.5)).to(dev)
        packed = ops.bottleneck_conv_pack(w)
        S = ops.bottleneck_conv_splits(1, C_out, C_in, H, W, stride)
        P = (H // stride) * (W // stride)
        partials = torch.empty(S * C_out * P, device=dev)
        for _ in range(4):      # the last launch's records are read back (caches warm)
            ops.bottleneck_conv_into(x, packed, C_out, stride, partials)
        torch.cuda.synchronize()
        n_waves = ((C_out + 15) // 16 + 3) // 4 * 4 * S * (P // 80)
        buf = np.zeros((8192, 8), dtype=np.uint64)
        assert lib.dvmvs_debug_bottleneck_conv_trace(buf.ctypes.data, 8192) == 0
        t = buf[:min(n_waves, 8192)]
        t = t[t[:, 5] > 0]
        ticks = t[:, :6].astype(np.float64) * 0.01      # 100 MHz -> us
        t0 = ticks[:, 0].min()
        start, staged, wait, mfma, loop, end = (ticks[:, 0] - t0, ticks[:, 1] - t0, ticks[:, 2], ticks[:, 3], ticks[:, 4] - t0, ticks[:, 5] - t0)
        simd = ((t[:, 6] >> np.uint64(32)) << np.uint64(16)) | (t[:, 6] & np.uint64(0xFF30))      # (XCC id, HW_ID bits: SE [15:13], SH [12], CU [11:8], SIMD [5:4])
        simd_end = {}
        for s_, e in zip(simd, end):
            simd_end[int(s_)] = max(simd_end.get(int(s_), 0.0), e)
        per_simd = np.bincount(np.unique(simd, return_inverse=True)[1])
        last = int(np.argmax(end))
        print(f"{C_in} -> {C_out} @ {H}x{W} stride {stride}: {S} splits, {len(t)} waves traced, groups per wave {int(t[0, 7])}; launch span {end.max():.2f} us "
              f"(waves start {start.min():.2f} .. {start.max():.2f} us, end {end.min():.2f} .. {end.max():.2f})")
        print(f"   mean per wave: staging {np.mean(staged - start):.2f} | weights wait {wait.mean():.2f} | MFMAs {mfma.mean():.2f} | rest of the loop "
              f"{np.mean(loop - staged - wait - mfma):.2f} | store {np.mean(end - loop):.2f} | lifetime {np.mean(end - start):.2f} us")
        print(f"   last wave:     staging {staged[last] - start[last]:.2f} | weights wait {wait[last]:.2f} | MFMAs {mfma[last]:.2f} | store {end[last] - loop[last]:.2f} | "
              f"started {start[last]:.2f}, ended {end[last]:.2f}")
        ends = np.array(sorted(simd_end.values()))
        print(f"   {len(simd_end)} SIMDs in use, waves per SIMD {per_simd.min()} .. {per_simd.max()} (mean {per_simd.mean():.2f}); a SIMD's last wave ends at "
              f"{ends.min():.2f} .. {ends.max():.2f} us (mean {ends.mean():.2f})")
        order = np.argsort(start)
        q = [order[int(f * (len(order) - 1))] for f in (0.0, 0.25, 0.5, 0.75, 1.0)]
        print("   start quartiles:", " ".join(f"{start[i]:.2f}" for i in q), "us")


if __name__ == "__main__":
    main()
