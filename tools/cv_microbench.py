"""Times the cost-volume kernel variants on the BASELINE.json shape (B=1, C=32, 128x160, D=64, M=2) on real keyframe
geometry (lines of the sample scene's nmeas+2 index) and cross-checks every variant against the generic kernel.
Run on the GPU box:

    python tools/cv_microbench.py [--lines 0,40,117,202 | all] [--variants 0,32,33 | auto] [--work-list] [--layouts nchw,nhwc] [--batch 1] [--lib tuning]

Each (geometry, layout, variant) is captured into a hipGraph of REPS back-to-back ops and timed with HIP events (no host gaps).
Variant numbers: include/dvmvs_hip.h (0-3; 2 / 3 = the two configurations of the LDS-tiled sweep); 32 + k = tuning configuration k of csrc/sweep_tiled.hip (+ 16 / 32 / 48: second-pass
grid of 512 / 1024 / 2048 workgroups), which exist only in the tools-only library built by `make -C deep-video-mvs_amd/csrc tuning`
(--lib tuning) -- the product library answers them with "invalid argument".
Prints one table (us per op) and writes it as JSON when --out is given.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from dvmvs import pose_algebra  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402


def index_lines(nmeas=2):
    return [[r] + list(ms) for r, ms in syn.keyframe_index_lines(nmeas)]


def load_library(which):
    """The product library, or one of the tools-only builds (same C ABI) next to it."""
    if which == "product":
        return _capi.lib()
    lib = ctypes.CDLL(os.path.join(ROOT, "deep-video-mvs_amd", "lib", f"libdvmvs_hip_{which}.so"))
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--layouts", default="nchw")
    ap.add_argument("--lines", default="0,40,80,117,170,202,250", help="index lines; -1 = the synthetic sideways trajectory")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default="product", choices=["product", "tuning", "trace"])
    ap.add_argument("--small-workspace", action="store_true", help="no spill workspace (single-pass sweep, inline gather)")
    ap.add_argument("--work-list", action="store_true", help="launch with the host-planned work list (dvmvs_sweep_work_list) of each geometry")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = load_library(args.lib)
    B, C, H, W, D, M = args.batch, 32, 128, 160, 64, args.m
    AUTO = -1      # "auto": per geometry the tiled configuration + work list of dvmvs_sweep_plan on the host matrices
    ENGINE = -2    # "engine": what DepthEngine launches -- dvmvs_sweep_plan6 (the MFMA sweep where its estimate takes it, else the tiled plan)
    variants = [AUTO if v == "auto" else ENGINE if v == "engine" else int(v) for v in args.variants.split(",")]
    layouts = args.layouts.split(",")
    feats = [torch.cat([syn.smooth_noise((1, C, H, W), seed=300 + 10 * b + i) for b in range(B)]).to(dev) for i in range(M + 1)]
    feats_cl = [t.contiguous(memory_format=torch.channels_last) for t in feats[1:]]
    K = syn.scaled_K(syn.full_K(), 2.0).repeat(B, 1, 1)
    allp = torch.from_numpy(syn.sample_poses()).float()
    lines = index_lines(2)
    ws_bytes = 0 if args.small_workspace else lib.dvmvs_cost_volume_workspace_bytes(B, M, H, W, D)
    ws = torch.zeros(max(1, (ws_bytes + 3) // 4), device=dev)   # header zero-filled once (contract in include/dvmvs_hip.h)
    out = torch.empty(B, D, H, W, device=dev)
    ref_out = torch.empty_like(out)
    alg_bytes = (1 + M) * B * C * H * W * 4 + B * D * H * W * 4
    print(f"shape B={B} C={C} {H}x{W} D={D} M={M}; algorithmic bytes {alg_bytes}; library: {args.lib}")
    results = {}
    for li in (range(len(lines)) if args.lines == "all" else [int(v) for v in args.lines.split(",")]):
        if li < 0:
            traj = torch.from_numpy(syn.synthetic_trajectory(10, seed=1000)).float()
            ids, pose_src = [8, 7, 6], traj
        else:
            ids, pose_src = lines[li], allp
        pose1 = pose_src[ids[0]:ids[0] + 1].repeat(B, 1, 1)
        pose2s = [pose_src[i:i + 1].repeat(B, 1, 1) for i in (ids[1:] * M)[:M]]
        Hm, kt, host = pose_algebra.sweep_matrices(pose1, pose2s, K, dev, "reference", with_host=True)
        lists, planned = {}, {}
        for pseudo in (AUTO, ENGINE):
            if pseudo in variants:
                from dvmvs.hip import ops as _ops
                plan = torch.zeros(_ops.sweep_work_list_words(B, H, W, D), dtype=torch.int32)
                planned[pseudo] = _ops.sweep_plan_host(host[0], host[1], H, W, D, 0.25, 20.0, 0, plan, allow_mfma=pseudo == ENGINE)
                lists[pseudo] = plan.to(dev)
        if args.work_list:
            from dvmvs.hip import ops as _ops
            for v in variants:
                if v in (AUTO, ENGINE):
                    continue
                if v in (0, 2, 3) or v >= 32:      # (tuning variants use the 32x8x8 tiling of the default configuration)
                    lists[v] = _ops.sweep_work_list_host(host[0], host[1], H, W, D, 0.25, 20.0, v if v < 32 else 2).to(dev)

        def launch(variant, dst, layout):
            meas = feats_cl if layout == "nhwc" else feats[1:]
            img_ptrs = _capi.pointer_array([t.data_ptr() for t in meas])
            rc = lib.dvmvs_cost_volume_planned_fwd(feats[0].data_ptr(), img_ptrs, Hm.data_ptr(), kt.data_ptr(), dst.data_ptr(),
                                                   B, M, C, H, W, D, 0.25, 20.0, 1, planned.get(variant, variant),
                                                   1 if layout == "nhwc" else 0,
                                                   ws.data_ptr() if ws_bytes else None, ws_bytes,
                                                   lists[variant].data_ptr() if variant in lists else None, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"variant {variant}: code {rc}: {lib.dvmvs_error_string(rc).decode()}")

        launch(1, ref_out, "nchw")      # generic kernel on NCHW maps = the cross-check
        torch.cuda.synchronize()
        for layout in layouts:
            for variant in variants:
                out.fill_(float("nan"))
                try:
                    launch(variant, out, layout)
                except RuntimeError as e:
                    print(f"line {li:3d} {layout} variant {variant:3d}: {e}")
                    continue
                torch.cuda.synchronize()
                err = (out - ref_out).abs().max().item()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(args.reps):
                        launch(variant, out, layout)
                g.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    g.replay()
                    e.record()
                    torch.cuda.synchronize()
                    best = min(best, s.elapsed_time(e) * 1e3 / args.reps)
                results[(li, layout, variant)] = (best, err)
                label = f"auto={planned[variant]}" if variant == AUTO else f"engine={planned[variant]}" if variant == ENGINE else f"{variant:3d}"
                print(f"line {li:3d} {layout} variant {label}: {best:8.2f} us  {100 * alg_bytes / best / 1e3 / 8000:5.2f} % of 8 TB/s  "
                      f"max|diff vs generic| {err:.2e}", flush=True)
    print("\nmean over geometries (us):")
    for layout in layouts:
        for variant in variants:
            ts = [v[0] for (li, lo, va), v in results.items() if lo == layout and va == variant]
            es = [v[1] for (li, lo, va), v in results.items() if lo == layout and va == variant]
            if ts:
                print(f"  {layout} variant {'auto' if variant == AUTO else 'engine' if variant == ENGINE else variant}: mean {sum(ts) / len(ts):8.2f}  min {min(ts):8.2f}  max {max(ts):8.2f}   worst diff {max(es):.2e}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({f"{li}/{lo}/{va}": {"us": v[0], "max_diff_vs_generic": v[1]} for (li, lo, va), v in results.items()}, f, indent=1)


if __name__ == "__main__":
    main()
