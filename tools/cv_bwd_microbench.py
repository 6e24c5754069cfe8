"""Times dvmvs_cost_volume_bwd (the measurement-feature gradient; gather kernel, scatter kernels for comparison) on the training step's own
geometry: B=4, C=32, 128x128 features, 64 planes, one measurement frame, pose pairs three frames apart on the sample scene
(the pairs bench.py --mode train feeds; BASELINE.json configs[4]).  Run on the GPU box:

    python tools/cv_bwd_microbench.py [--lib tuning --configs 5,0,1,2,3,4,9] [--frames 1,2,3,4,5,6,7] [--out gpurun_out/x.json]

Each (frame, configuration) is captured into a hipGraph of REPS back-to-back calls and timed with HIP events; the time of the
reference-feature gather kernel alone (no measurement gradient requested) is measured the same way and subtracted.  With
--lib tuning the configurations of csrc/cost_volume_bwd.hip are selected through dvmvs_tuning_set_bwd_config (5 = the
product's gather kernel, 0-4 = the LDS-privatised scatter of rounds 1-3 with different channel chunks / LDS windows, 9 = the
plain global-atomic scatter, which is also the cross-check).  The product library has only configuration 5 and no
cross-check here: its parity tests are tests/test_hip_parity.py (float64 autograd).
"""
import argparse
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from cv_microbench import load_library  # noqa: E402
from dvmvs import pose_algebra  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402


def timed(graph, reps):
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="product", choices=["product", "tuning"])
    ap.add_argument("--configs", default="5")
    ap.add_argument("--frames", default="1,2,3,4,5,6,7")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = load_library(args.lib)
    if args.lib == "tuning":
        lib.dvmvs_tuning_set_bwd_config.restype, lib.dvmvs_tuning_set_bwd_config.argtypes = None, [_capi.ctypes.c_int]
    configs = [int(v) for v in args.configs.split(",")]
    if args.lib != "tuning" and configs != [5]:
        raise SystemExit("the product library has only configuration 5; use --lib tuning")
    B, C, H, W, D = args.batch, 32, 128, 128, 64
    g = torch.Generator().manual_seed(11)
    f1 = torch.randn(B, C, H, W, generator=g).to(dev)
    f2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, D, H, W, generator=g).to(dev)
    g1, g2, g2_ref = torch.zeros_like(f1), torch.zeros_like(f2), torch.zeros_like(f2)
    K = syn.scaled_K(syn.full_K(width=256, height=256), 2.0).repeat(B, 1, 1)
    allp = torch.from_numpy(syn.sample_poses()).float()
    img2 = _capi.pointer_array([f2.data_ptr()])
    print(f"shape B={B} C={C} {H}x{W} D={D} M=1; library: {args.lib}; {4 * B * H * W * D * C / 1e6:.0f} M tap-channel additions per call")
    results = {}
    for i in [int(v) for v in args.frames.split(",")]:
        pose1 = torch.stack([allp[(40 * b + 3 * i) % len(allp)] for b in range(B)])
        pose2 = torch.stack([allp[(40 * b + 3 * (i - 1)) % len(allp)] for b in range(B)])
        Hm, kt = pose_algebra.sweep_matrices(pose1, [pose2], K, dev, "reference")

        def call(dst2):
            rc = lib.dvmvs_cost_volume_bwd(go.data_ptr(), f1.data_ptr(), img2, Hm.data_ptr(), kt.data_ptr(), g1.data_ptr(),
                                           _capi.pointer_array([dst2.data_ptr() if dst2 is not None else None]),
                                           B, 1, C, H, W, D, 0.25, 20.0, torch.cuda.current_stream().cuda_stream)   # the capturing stream inside a graph
            if rc != 0:
                raise RuntimeError(f"code {rc}: {lib.dvmvs_error_string(rc).decode()}")

        def graph_of(dst2):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(args.reps):
                    call(dst2)
            gr.replay()
            torch.cuda.synchronize()
            return gr

        gather_us = timed(graph_of(None), args.reps)
        if args.lib == "tuning":
            lib.dvmvs_tuning_set_bwd_config(9)
            g2_ref.zero_()
            call(g2_ref)
            torch.cuda.synchronize()
        for cfg in configs:
            if args.lib == "tuning":
                lib.dvmvs_tuning_set_bwd_config(cfg)
            g2.zero_()
            call(g2)
            torch.cuda.synchronize()
            err = ((g2 - g2_ref).abs().max() / g2_ref.abs().max()).item() if args.lib == "tuning" else float("nan")
            us = timed(graph_of(g2), args.reps) - gather_us
            results[(i, cfg)] = (us, err)
            print(f"frame {i} config {cfg}: scatter {us:9.1f} us  (gather kernel {gather_us:7.1f} us)   max|diff vs global-atomic scatter| / max|g| {err:.2e}",
                  flush=True)
    print("\nmean over frames (us):")
    for cfg in configs:
        ts = [v[0] for (i, c), v in results.items() if c == cfg]
        es = [v[1] for (i, c), v in results.items() if c == cfg]
        print(f"  config {cfg}: mean {sum(ts) / len(ts):9.1f}  min {min(ts):9.1f}  max {max(ts):9.1f}   worst diff {max(es):.2e}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({f"{i}/{c}": {"scatter_us": v[0], "rel_diff_vs_global_atomics": v[1]} for (i, c), v in results.items()}, f, indent=1)


if __name__ == "__main__":
    main()
