"""EXPERIMENT: the sweep with adaptive plane counts and claimable half units (csrc/sweep_experimental.hip) against the shipped
sweep, on keyframe geometries of the sample scene (needs `make -C deep-video-mvs_amd/csrc trace`, or `trace-timeline` for
--timeline: the timeline's bookkeeping spills registers, so timings are taken without it; GPU box).

    python tools/sweep_adaptive_bench.py [--lines=-1,0,35,118,153,177,201]

Per line: microseconds per operation of both (hipGraph of back-to-back launches, HIP events; the experimental entry point
clears its two scratch buffers inside the timed region, timed separately as "clears"), whether the volumes are bit-identical,
the difference to the generic kernel, how many half units were published, and the first passes alone (dvmvs_debug_mode).

Round-2 result (MI355X): bit-identical volumes on all 7 lines; no overhead when nothing is published (36.4 vs 37.2 us); on the
worst line 177 halves are published and 171 claimed, but the first pass does not get shorter (63.9 vs 64.7 us incl. 4 us of
clears).  --timeline shows why: a claimed half takes 36 us on a chip that is still full, so the claimers end last instead of
the owners; lines 118 and 201 publish (almost) nothing because their heavy frame is the second one.  DESIGN.md section 7.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402
from cv_microbench import index_lines  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


def timeline(lib, groups, li, head):
    import numpy as np
    wg = np.zeros((groups, 8), dtype=np.uint64)
    half = np.zeros((groups, 4), dtype=np.uint64)
    lib.dvmvs_debug_adaptive_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert lib.dvmvs_debug_adaptive_trace(wg.ctypes.data, half.ctypes.data, groups) == 0
    wg, half = wg.astype(np.float64), half.astype(np.float64)
    t0 = wg[:, 0].min()
    start, first, end = (wg[:, 0] - t0) / 100.0, (wg[:, 1] - t0) / 100.0, (wg[:, 2] - t0) / 100.0
    tasks, runs = wg[:, 3], wg[:, 4]
    published, light = (wg[:, 6].astype(np.int64) & 1) == 1, (wg[:, 6].astype(np.int64) & 2) == 2
    pub_t = np.where(published, (wg[:, 5] - t0) / 100.0, np.nan)
    half[half[:, 0] < t0, 1] = 0   # records of earlier launches
    claimed = half[:, 1] > 0
    claim_t, done_t = (half[:, 0] - t0) / 100.0, (half[:, 2] - t0) / 100.0
    print(f"line {li}: first pass spans {end.max():.1f} us; published {head[0]} halves (at {np.nanmedian(pub_t) if published.any() else float('nan'):.1f} us median); "
          f"claimed by owner {int((half[:, 1] == 1).sum())}, by another workgroup {int((half[:, 1] == 2).sum())}")
    if claimed.any():
        for who, name in ((1, "owner"), (2, "other")):
            sel = half[:, 1] == who
            if sel.any():
                print(f"   halves done by {name}: claimed at {np.median(claim_t[sel]):.1f} us median ({claim_t[sel].min():.1f}..{claim_t[sel].max():.1f}), "
                      f"done at {np.median(done_t[sel]):.1f} median, {done_t[sel].max():.1f} max; duration {np.median((done_t - claim_t)[sel]):.1f} us median")
    order = np.argsort(-end)[:32]
    print(f"   last 32 workgroups to end: end {end[order].mean():.1f} us, own unit done at {first[order].mean():.1f}, tasks {tasks[order].mean():.2f}, "
          f"staged runs of the own unit {runs[order].mean():.1f}, published {published[order].mean():.2f}, light {light[order].mean():.2f}")
    for name, sel in (("publishers", published), ("light workgroups", light), ("the rest", ~published & ~light)):
        if sel.any():
            print(f"   {name:17s} n={int(sel.sum()):4d}: own unit done at {first[sel].mean():5.1f} us (max {first[sel].max():5.1f}), end {end[sel].mean():5.1f} (max {end[sel].max():5.1f}), "
                  f"tasks {tasks[sel].mean():.2f}, staged runs {runs[sel].mean():.1f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", default="-1,0,35,118,153,177,201")
    ap.add_argument("--timeline", action="store_true", help="print who ends last (one launch per line, no timing)")
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, "deep-video-mvs_amd", "lib", "libdvmvs_hip_trace.so"))
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    P, I, D_, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.dvmvs_debug_sweep_adaptive.argtypes = [P, ctypes.POINTER(P), P, ctypes.POINTER(P), P, P, I, I, I, I, I, I, D_, D_, P, Z, P, Z, P]
    lib.dvmvs_debug_sweep_adaptive.restype = I
    dev = torch.device("cuda:0")
    B, C, H, W, D, M = 1, 32, 128, 160, 64, 2
    feats = [syn.smooth_noise((1, C, H, W), seed=300 + i).to(dev) for i in range(M + 1)]
    K = syn.scaled_K(syn.full_K(), 2.0).to(dev)
    allp = torch.from_numpy(syn.sample_poses()).float()
    lines = index_lines(2)
    ws_bytes = lib.dvmvs_cost_volume_workspace_bytes(B, M, H, W, D)
    ws = torch.zeros((ws_bytes + 3) // 4, device=dev)
    ws2 = torch.zeros_like(ws)
    groups = (W // 32) * (H // 8) * (D // 8) * B
    steal = torch.zeros(16 + 2 * groups, dtype=torch.int32, device=dev)
    out_p, out_a, out_g = (torch.empty(B, D, H, W, device=dev) for _ in range(3))
    img_ptrs = _capi.pointer_array([t.data_ptr() for t in feats[1:]])
    stream = lambda: torch.cuda.current_stream().cuda_stream
    rows = []
    for li in [int(v) for v in args.lines.split(",")]:
        if li < 0:
            pose_src, ids = torch.from_numpy(syn.synthetic_trajectory(10, seed=1000)).float(), [8, 7, 6]
        else:
            pose_src, ids = allp, lines[li]
        pose1 = pose_src[ids[0]:ids[0] + 1].to(dev)
        pose2s = [pose_src[i:i + 1].to(dev) for i in ids[1:]]
        pose_ptrs = _capi.pointer_array([t.data_ptr() for t in pose2s])

        def product(variant=0, dst=out_p):
            rc = lib.dvmvs_cost_volume_fwd(feats[0].data_ptr(), img_ptrs, pose1.data_ptr(), pose_ptrs, K.data_ptr(), dst.data_ptr(), B, M, C, H, W, D,
                                           0.25, 20.0, 1, variant, 0, ws.data_ptr(), ws_bytes, stream())
            assert rc == 0, rc

        def adaptive():
            rc = lib.dvmvs_debug_sweep_adaptive(feats[0].data_ptr(), img_ptrs, pose1.data_ptr(), pose_ptrs, K.data_ptr(), out_a.data_ptr(), B, M, C, H, W, D,
                                                0.25, 20.0, ws2.data_ptr(), ws_bytes, steal.data_ptr(), steal.numel() * 4, stream())
            assert rc == 0, rc

        def clears():
            ws2.zero_()
            steal.zero_()

        out_a.fill_(float("nan"))
        product(1, out_g)
        product()
        adaptive()
        torch.cuda.synchronize()
        head = steal[:3].tolist()
        if args.timeline:
            for _ in range(4):   # warm caches: the records read back are those of the last launch
                adaptive()
            torch.cuda.synchronize()
            timeline(lib, groups, li, steal[:3].tolist())
            continue
        same = bool(torch.equal(out_p, out_a))
        diff_generic, diff_shipped = (out_a - out_g).abs().max().item(), (out_a - out_p).abs().max().item()
        mode = ctypes.c_int.in_dll(lib, "dvmvs_debug_mode")
        extra = {}
        for name, value in (("adaptive first pass", 1), ("shipped first pass", 3), ("plain (uninstrumented shipped) op", 16), ("positions-in-tap-blocks op", 8),
                            ("plain op, no clears", 48), ("instrumented shipped op, no clears", 34)):
            mode.value = value
            extra[name] = timed(adaptive)
        mode.value = 0
        rows.append((li, timed(product), timed(adaptive), timed(clears), same, diff_generic, diff_shipped, head))
        r = rows[-1]
        print(f"line {li:3d}: shipped {r[1]:7.2f} us | adaptive {r[2]:7.2f} us (of which clears {r[3]:5.2f}) | bit-identical {r[4]} "
              f"max|adaptive - shipped| {r[6]:.1e}  max|adaptive - generic| {r[5]:.1e} | halves published {r[7][0]}, claim cursor {r[7][1]}, error {r[7][2]}", flush=True)
        print("          incl. clears: " + ", ".join(f"{k} {v:.2f}" for k, v in extra.items()), flush=True)
    if rows:
      print("mean: shipped %.2f us, adaptive %.2f us (%.2f without the clears)" % (
        sum(r[1] for r in rows) / len(rows), sum(r[2] for r in rows) / len(rows), sum(r[2] - r[3] for r in rows) / len(rows)))


if __name__ == "__main__":
    main()
