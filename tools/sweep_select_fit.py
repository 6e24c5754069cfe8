"""Fits / checks the cost model behind dvmvs_sweep_select_variant (csrc/cost_volume.hip: sweep_model_us).

    python tools/sweep_select_fit.py --stats                      # plan statistics of both configurations on all keyframe pairs (CPU only)
    python tools/sweep_select_fit.py --timings gpurun_out/x.json   # + least-squares fit to per-pair timings of variants 2 and 3
                                                                   #   (tools/cv_microbench.py --variants 2,3 --lines all --work-list --out x.json)

The statistics come from the library's HOST-side plan model (dvmvs_sweep_plan_stats: no GPU needed); the timings from an MI355X.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from dvmvs import pose_algebra  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402

H, W, D = 128, 160, 64


def plan_stats(Hm, kt, configuration):
    out = (ctypes.c_longlong * 8)()
    Hm, kt = Hm.contiguous().float(), kt.contiguous().float()
    B, M = Hm.shape[0], Hm.shape[1]
    rc = _capi.lib().dvmvs_sweep_plan_stats(Hm.data_ptr(), kt.data_ptr(), B, M, H, W, D, 0.25, 20.0, configuration, out)
    _capi.check(rc, "dvmvs_sweep_plan_stats")
    return list(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--timings", default="")
    args = ap.parse_args()
    lines = syn.keyframe_index_lines(2)
    poses = torch.from_numpy(syn.sample_poses()).float()
    K = syn.scaled_K(syn.full_K(), 2.0)
    rows, sel = [], []
    for li, (r, ms) in enumerate(lines):
        Hm, kt = pose_algebra.sweep_matrices_host(poses[r:r + 1], [poses[m:m + 1] for m in ms], K)
        rows.append((plan_stats(Hm, kt, 0), plan_stats(Hm, kt, 1)))
        sel.append(_capi.lib().dvmvs_sweep_select_variant(Hm.contiguous().data_ptr(), kt.contiguous().data_ptr(), 1, 2, H, W, D, 0.25, 20.0))
        if args.stats:
            print(f"line {li:3d}: default {rows[-1][0]}  wide {rows[-1][1]}  -> variant {sel[-1]}")
    if not args.timings:
        return
    t = json.load(open(args.timings))
    us = {v: np.array([t[f"{li}/nchw/{v}"]["us"] for li in range(len(lines))]) for v in (2, 3)}
    for cfg, v in ((0, 2), (1, 3)):
        st = np.array([r[cfg] for r in rows], dtype=np.float64)
        # the model of csrc/cost_volume.hip: sweep_model_us (launches WITH the work list: chains are cut to <= 3 staged runs)
        A = np.stack([np.ones(len(st)), np.minimum(st[:, 6], 3), (st[:, 4] > 0) * 1.0, st[:, 7], st[:, 4], st[:, 1]], 1)
        coef, *_ = np.linalg.lstsq(A, us[v], rcond=None)
        pred = A @ coef
        print(f"variant {v}: base {coef[0]:.4f} us, per staged run of the longest work item (<= 3) {coef[1]:.4f}, non-empty second pass {coef[2]:.4f}, "
              f"per queued plane of the worst workgroup {coef[3]:.4f}, per queued plane {coef[4]:.5f}, per staged record {coef[5]:.3e};  "
              f"rms residual {np.sqrt(np.mean((pred - us[v]) ** 2)):.2f} us, max {np.abs(pred - us[v]).max():.1f}")
        us[f"pred{v}"] = pred
    best = np.minimum(us[2], us[3])
    chosen = np.where(us["pred2"] <= us["pred3"], us[2], us[3])
    print(f"mean us over {len(lines)} keyframe pairs: default everywhere {us[2].mean():.2f} (worst {us[2].max():.1f}), wide everywhere {us[3].mean():.2f} "
          f"(worst {us[3].max():.1f}), the faster of the two {best.mean():.2f}, model's choice {chosen.mean():.2f} (worst {chosen.max():.1f}; wide on "
          f"{int((us['pred2'] > us['pred3']).sum())} pairs)")
    lib_choice = np.array([us[2][i] if sel[i] == 2 else us[3][i] for i in range(len(lines))])
    print(f"the library's current dvmvs_sweep_select_variant on these timings: mean {lib_choice.mean():.2f}, worst {lib_choice.max():.1f}, wide on {sel.count(3)} pairs")


if __name__ == "__main__":
    main()
