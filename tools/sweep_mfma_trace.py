"""Per-wave timeline of the correlate-then-interpolate sweep (csrc/sweep_mfma.hip; s_memtime brackets; needs `make -C deep-video-mvs_amd/csrc trace`).

    python tools/sweep_mfma_trace.py [--lines 0,99,202] [--layout nhwc] [--variant 96]

For every wave the instrumented build records start / end, the time up to the first frame (reference features, tables), in the sample
positions, in the box reductions, in the tile loops (operand requests + MFMAs + table writes) and in the interpolation, the number of
tiles and strips, and the CU it ran on.  Printed: launch span, mean / slowest wave's phases, waves per CU and per SIMD.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from dvmvs import pose_algebra  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402
from cv_microbench import index_lines, load_library  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", default="0,99,202")
    ap.add_argument("--layout", default="nhwc")
    ap.add_argument("--variant", type=int, default=96)
    ap.add_argument("--waves", type=int, default=5120)
    args = ap.parse_args()
    lib = load_library("trace")
    lib.dvmvs_debug_sweep_mfma_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    B, C, H, W, D, M = 1, 32, 128, 160, 64, 2
    feats = [syn.smooth_noise((1, C, H, W), seed=300 + i).to(dev) for i in range(M + 1)]
    if args.layout == "nhwc":
        feats[1:] = [t.contiguous(memory_format=torch.channels_last) for t in feats[1:]]
    K = syn.scaled_K(syn.full_K(), 2.0)
    allp = torch.from_numpy(syn.sample_poses()).float()
    lines = index_lines(2)
    out = torch.empty(B, D, H, W, device=dev)
    for li in [int(v) for v in args.lines.split(",")]:
        ids = lines[li]
        Hm, kt = pose_algebra.sweep_matrices(allp[ids[0]:ids[0] + 1], [allp[i:i + 1] for i in ids[1:]], K, dev, "reference")
        img_ptrs = _capi.pointer_array([t.data_ptr() for t in feats[1:]])
        for _ in range(4):    # the last launch's records are the ones read back (caches warm)
            rc = lib.dvmvs_cost_volume_fwd(feats[0].data_ptr(), img_ptrs, Hm.data_ptr(), kt.data_ptr(), out.data_ptr(), B, M, C, H, W, D,
                                           0.25, 20.0, 1, args.variant, 1 if args.layout == "nhwc" else 0, None, 0,
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        t = np.zeros((args.waves, 16), dtype=np.uint64)
        assert lib.dvmvs_debug_sweep_mfma_trace(t.ctypes.data, args.waves) == 0
        t = t[t[:, 1] > 0]
        ntiles = (t[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
        nstrips = (t[:, 7] >> np.uint64(32)).astype(np.float64)
        hw, xcc = t[:, 8].astype(np.int64), t[:, 9].astype(np.int64) & 0xf
        chunk = t[:, 13].astype(np.int64)
        t = t.astype(np.float64)
        tick = (t[:, 11] - t[:, 10]).sum() / 100e6 / np.maximum((t[:, 1] - t[:, 0]).sum(), 1)   # seconds per s_memtime tick
        us = tick * 1e6
        real0 = t[:, 10].min()
        start, end = (t[:, 10] - real0) / 100.0, (t[:, 11] - real0) / 100.0     # us on the 100 MHz wall clock
        dur = (t[:, 1] - t[:, 0]) * us
        pro, pos, box, tiles, look = t[:, 2] * us, t[:, 3] * us, t[:, 4] * us, t[:, 5] * us, t[:, 6] * us
        other = dur - pro - pos - box - tiles - look
        cu_key = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 15)
        simd_key = cu_key * 4 + ((hw >> 4) & 3)
        per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
        per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1])
        print(f"\nline {li}: {args.layout} variant {args.variant}; {len(t)} waves; s_memtime tick {tick * 1e9:.3f} ns; launch span {end.max():.1f} us "
              f"(first start {start.min():.2f}, last start {start.max():.2f}, median start {np.median(start):.2f})")
        print(f"  waves per CU: min {per_cu.min()} max {per_cu.max()} on {len(per_cu)} CUs; per SIMD: min {per_simd.min()} max {per_simd.max()} on {len(per_simd)} SIMDs")
        print(f"  tiles per wave: mean {ntiles.mean():.1f} max {ntiles.max():.0f}; strips per wave: mean {nstrips.mean():.2f} max {nstrips.max():.0f}")
        def row(label, sel):
            print(f"  {label:>14}: dur {dur[sel].mean():6.2f} = prologue {pro[sel].mean():5.2f} + positions {pos[sel].mean():5.2f} + boxes {box[sel].mean():5.2f} "
                  f"+ tiles {tiles[sel].mean():5.2f} + interpolation {look[sel].mean():5.2f} + other {other[sel].mean():5.2f} us; tiles {ntiles[sel].mean():5.1f} "
                  f"({tiles[sel].sum() / max(ntiles[sel].sum(), 1) * 1e3:.0f} ns/tile)")
        row("all waves", slice(None))
        order = np.argsort(dur)
        row("slowest 5 %", order[-len(order) // 20:])
        row("fastest 5 %", order[:len(order) // 20])
        for c in range(4):
            row(f"chunk {c}", chunk == c)
        # ---- per SIMD (round 6: where does a SIMD idle?) ----
        simd_index = np.unique(simd_key, return_inverse=True)[1]
        n_simd = simd_index.max() + 1
        s_waves = np.bincount(simd_index, minlength=n_simd)
        s_tiles = np.bincount(simd_index, weights=ntiles, minlength=n_simd)
        s_occupied = np.bincount(simd_index, weights=dur, minlength=n_simd)      # sum of its waves' lifetimes
        s_first = np.full(n_simd, np.inf)
        s_last = np.zeros(n_simd)
        np.minimum.at(s_first, simd_index, start)
        np.maximum.at(s_last, simd_index, end)
        # time with at least one wave resident, and mean number of resident waves while any is
        s_covered = np.zeros(n_simd)
        for k in range(n_simd):
            sel = np.nonzero(simd_index == k)[0]
            order_k = sel[np.argsort(start[sel])]
            cur_s, cur_e, total = None, None, 0.0
            for i in order_k:
                if cur_s is None:
                    cur_s, cur_e = start[i], end[i]
                elif start[i] <= cur_e:
                    cur_e = max(cur_e, end[i])
                else:
                    total += cur_e - cur_s
                    cur_s, cur_e = start[i], end[i]
            s_covered[k] = total + (cur_e - cur_s if cur_s is not None else 0.0)
        span = end.max()
        print(f"  per SIMD ({n_simd} SIMDs): waves per SIMD histogram " + ", ".join(f"{k}: {int((s_waves == k).sum())}" for k in range(int(s_waves.min()), int(s_waves.max()) + 1)))
        print(f"    tiles per SIMD: min {s_tiles.min():.0f} mean {s_tiles.mean():.1f} max {s_tiles.max():.0f}; sum of wave lifetimes per SIMD: min {s_occupied.min():.1f} mean {s_occupied.mean():.1f} "
              f"max {s_occupied.max():.1f} us")
        print(f"    last wave of a SIMD ends: p10 {np.percentile(s_last, 10):.1f} p50 {np.percentile(s_last, 50):.1f} p90 {np.percentile(s_last, 90):.1f} max {s_last.max():.1f} us "
              f"(launch span {span:.1f}); SIMD has no wave resident for {np.mean(span - s_covered):.1f} us on average ({100 * np.mean(span - s_covered) / span:.0f} % of the span), "
              f"worst {np.max(span - s_covered):.1f}, best {np.min(span - s_covered):.1f}")
        print(f"    mean resident waves per SIMD while any is resident: {np.mean(s_occupied / np.maximum(s_covered, 1e-9)):.2f}")
        for k in range(int(s_waves.min()), int(s_waves.max()) + 1):
            m = s_waves == k
            if m.any():
                print(f"    SIMDs with {k} waves ({int(m.sum()):4d}): last end mean {s_last[m].mean():5.1f} max {s_last[m].max():5.1f} us; tiles {s_tiles[m].mean():5.1f}; wave lifetime mean "
                      f"{(s_occupied[m] / k).mean():5.2f} us")
        cc = np.corrcoef(s_waves, s_last)[0, 1], np.corrcoef(s_tiles, s_last)[0, 1]
        print(f"    correlation of a SIMD's last end with its wave count {cc[0]:.2f}, with its tile count {cc[1]:.2f}")
        edges = np.arange(0.0, span + 2.0, 2.0)
        alive = [(np.minimum(end, hi) - np.maximum(start, lo)).clip(min=0).sum() / (hi - lo) for lo, hi in zip(edges[:-1], edges[1:])]
        print("    resident waves on the chip per 2 us bin: " + " ".join(f"{a:.0f}" for a in alive))
        q = np.percentile(start, [50, 75, 90, 100])
        print(f"  start times p50 {q[0]:.1f} p75 {q[1]:.1f} p90 {q[2]:.1f} max {q[3]:.1f} us; end times p50 {np.percentile(end, 50):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f} us")


if __name__ == "__main__":
    main()
