"""Per-workgroup timeline of the fused sweep kernel (s_memtime brackets; needs `make -C deep-video-mvs_amd/csrc trace`).

    python tools/sweep_trace.py [--lines=-1,153,118,177] [--layout nchw] [--variant 0]

For every workgroup the instrumented build records when it started and ended, the time spent on its tables, on the run plan,
in the staging half of the channel passes (stores of the prefetched pieces + late pieces -> barrier), in their tap half
(next stage's requests + taps -> barrier), on switching to the next run (tap addresses), and the hardware CU it ran on.
Printed: launch span, how the slowest / fastest workgroups spent their time, and workgroups per CU.
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
    ap.add_argument("--lines", default="-1,153,118,177")
    ap.add_argument("--layout", default="nchw")
    ap.add_argument("--variant", type=int, default=0)
    args = ap.parse_args()
    lib = load_library("trace")
    lib.dvmvs_debug_sweep_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    B, C, H, W, D, M = 1, 32, 128, 160, 64, 2
    feats = [syn.smooth_noise((1, C, H, W), seed=300 + i).to(dev) for i in range(M + 1)]
    if args.layout == "nhwc":
        feats[1:] = [t.contiguous(memory_format=torch.channels_last) for t in feats[1:]]
    K = syn.scaled_K(syn.full_K(), 2.0)
    allp = torch.from_numpy(syn.sample_poses()).float()
    lines = index_lines(2)
    ws_bytes = lib.dvmvs_cost_volume_workspace_bytes(B, M, H, W, D)
    ws = torch.zeros((ws_bytes + 3) // 4, device=dev)
    out = torch.empty(B, D, H, W, device=dev)
    groups = (W // 32) * (H // 8) * (D // 8)
    for li in [int(v) for v in args.lines.split(",")]:
        if li < 0:
            pose_src, ids = torch.from_numpy(syn.synthetic_trajectory(10, seed=1000)).float(), [8, 7, 6]
        else:
            pose_src, ids = allp, lines[li]
        Hm, kt = pose_algebra.sweep_matrices(pose_src[ids[0]:ids[0] + 1], [pose_src[i:i + 1] for i in ids[1:]], K, dev, "reference")
        img_ptrs = _capi.pointer_array([t.data_ptr() for t in feats[1:]])
        for _ in range(4):    # the last launch's records are the ones read back (caches warm)
            rc = lib.dvmvs_cost_volume_fwd(feats[0].data_ptr(), img_ptrs, Hm.data_ptr(), kt.data_ptr(), out.data_ptr(), B, M, C, H, W, D,
                                           0.25, 20.0, 1, args.variant, 1 if args.layout == "nhwc" else 0, ws.data_ptr(), ws_bytes,
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        t = np.zeros((groups, 16), dtype=np.uint64)
        assert lib.dvmvs_debug_sweep_trace(t.ctypes.data, groups) == 0
        block = (t[:, 12] >> np.uint64(32)).astype(np.int64)
        t[:, 12] &= np.uint64(0xffffffff)
        t = t.astype(np.float64)
        tick = (t[:, 11] - t[:, 10]).sum() / 100e6 / np.maximum((t[:, 2] - t[:, 0]).sum(), 1)   # seconds per s_memtime tick
        us = tick * 1e6
        real0 = t[:, 10].min()
        start, end = (t[:, 10] - real0) / 100.0, (t[:, 11] - real0) / 100.0     # us on the 100 MHz wall clock
        dur = (t[:, 2] - t[:, 0]) * us
        setup, plan, stage, taps, switch = (t[:, 1] - t[:, 0]) * us, t[:, 3] * us, t[:, 4] * us, t[:, 5] * us, t[:, 13] * us
        loop_end = t[:, 15] * us
        tail = dur - loop_end
        other = dur - setup - plan - stage - taps - switch - tail
        hw, xcc = t[:, 8].astype(np.int64), t[:, 9].astype(np.int64) & 0xf
        cu_key = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 15)
        per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
        print(f"\nline {li}: {args.layout}; s_memtime tick {tick * 1e9:.3f} ns; launch span {end.max():.1f} us (first start {start.min():.2f}, last start {start.max():.2f})")
        print(f"  workgroups per CU: {np.bincount(per_cu)[1:].tolist()} CUs with 1,2,3.. ; CUs used {len(per_cu)}")
        order = np.argsort(-end)

        def row(sel, name):
            print(f"  {name:28s} n={len(sel):4d} dur {dur[sel].mean():6.1f}  tables {setup[sel].mean():5.2f} plan {plan[sel].mean():5.2f} stage {stage[sel].mean():6.2f} "
                  f"taps {taps[sel].mean():6.2f} switch {switch[sel].mean():5.2f} tail {tail[sel].mean():5.2f} other {other[sel].mean():5.2f} | passes {t[sel, 6].mean():5.1f} "
                  f"runs {t[sel, 14].mean():4.1f} records {t[sel, 7].mean():7.0f} spilled runs {t[sel, 12].mean():4.1f} end {end[sel].mean():6.1f}")
        row(np.arange(groups), "all workgroups")
        row(order[:32], "last 32 to finish")
        row(order[-160:], "first 160 to finish")
        same = [np.where(cu_key == k)[0] for k in np.unique(cu_key)]
        for n in (2, 3):
            sel = np.concatenate([g for g in same if len(g) == n] or [np.array([], dtype=int)])
            if len(sel):
                row(sel, f"on CUs holding {n}")
        per_pass_stage = stage.sum() / max(t[:, 6].sum(), 1)
        per_pass_taps = taps.sum() / max(t[:, 6].sum(), 1)
        print(f"  per channel pass: staging half {per_pass_stage:.2f} us, tap half {per_pass_taps:.2f} us; records per staged box {t[:, 7].sum() / max(t[:, 14].sum(), 1):.0f}")


if __name__ == "__main__":
    main()
