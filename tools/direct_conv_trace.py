"""Per-wave timeline of the direct convolution kernel (s_memrealtime brackets; needs `make -C deep-video-mvs_amd/csrc trace`).

    python tools/direct_conv_trace.py

For a few layers of the frame: the launch's span (first wave's start to last wave's end) and, averaged over the waves (and for the
slowest one), where a wave spends its time: until the first patch is staged, issuing the next chunk's requests, in the MFMAs, storing
the next patch + barrier, adding the channel splits, writing the output.
"""
import ctypes
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["DVMVS_HIP_LIB"] = os.path.join(ROOT, "deep-video-mvs_amd", "lib", "libdvmvs_hip_trace.so")
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvmvs.hip import _capi, ops  # noqa: E402

LAYERS = [  # (C_in, H, W, C_out, k, stride)
    (128, 32, 40, 128, 3, 1), (256, 32, 40, 128, 3, 1), (32, 32, 40, 32, 3, 1), (128, 64, 80, 64, 3, 1), (64, 64, 80, 64, 5, 1),
    (96, 128, 160, 32, 5, 1), (32, 128, 160, 32, 3, 1), (36, 256, 320, 32, 5, 1),
]


def main():
    lib = _capi.lib()
    lib.dvmvs_debug_direct_conv_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for C_in, H, W, C_out, k, s in LAYERS:
        x = torch.randn(1, C_in, H, W, generator=g).to(dev)
        w = (torch.randn(C_out, C_in, k, k, generator=g) / (C_in * k * k) ** 0.5).to(dev)
        bias = torch.randn(C_out, generator=g).to(dev)
        n_tile = ops.direct_conv_tile(1, C_in, H, W, C_out, k, s)
        packed = ops.direct_conv_pack(w, n_tile)
        dst = torch.empty(1, C_out, H // s, W // s, device=dev)
        for _ in range(4):      # the last launch's records are read back (caches warm)
            ops.direct_conv_into(x, packed, n_tile, bias, dst, C_out, k, s, 1)
        torch.cuda.synchronize()
        tile_w = 80 if (W // s) % 80 == 0 and n_tile * 0 == 0 else 40
        waves = 16384
        buf = np.zeros((waves, 8), dtype=np.uint64)
        rc = lib.dvmvs_debug_direct_conv_trace(buf.ctypes.data, waves)
        assert rc == 0, rc
        t = buf[buf[:, 7] > 0].astype(np.float64)
        # only this launch's waves: records of earlier (larger) launches may remain behind them -- keep those that started last
        newest = t[:, 0].max()
        t = t[t[:, 0] > newest - 100 * 100]          # within 100 us of the newest start
        t0 = t[:, 0].min()
        us = lambda v: v / 100.0
        span = us(t[:, 7].max() - t0)
        first, req, mfma, store = us(t[:, 1] - t[:, 0]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4])
        red, epi, total = us(t[:, 6] - t[:, 5]), us(t[:, 7] - t[:, 6]), us(t[:, 7] - t[:, 0])
        late = us(t[:, 0] - t0)
        slow = int(np.argmax(t[:, 7]))
        print(f"{C_in:4d}x{H:3d}x{W:3d} -> {C_out:4d} k{k} s{s} (tile {n_tile}): {len(t)} waves, launch span {span:6.2f} us; wave start {late.mean():5.2f} (max {late.max():5.2f}) | "
              f"to first patch {first.mean():5.2f} | requests {req.mean():5.2f} | MFMAs {mfma.mean():5.2f} | store+barrier {store.mean():5.2f} | "
              f"splits {red.mean():5.2f} | output {epi.mean():5.2f} | wave total {total.mean():5.2f} (max {total.max():5.2f})")
        print(f"      last wave to end: started {late[slow]:5.2f}, first patch {first[slow]:5.2f}, requests {req[slow]:5.2f}, MFMAs {mfma[slow]:5.2f}, "
              f"store+barrier {store[slow]:5.2f}, splits {red[slow]:5.2f}, output {epi[slow]:5.2f}")


if __name__ == "__main__":
    main()
