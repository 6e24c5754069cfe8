"""Times the cost-volume kernel variants on the BASELINE.json shape (B=1, C=32, 128x160, D=64) with realistic geometry
and cross-checks every variant against the generic kernel.  Run on the GPU box:

    python tools/cv_microbench.py [--m 2] [--variants 1,2,16,17,...] [--batch 1]

Each variant is captured into a hipGraph of REPS back-to-back launches and timed with HIP events (no host gaps).
"""
import argparse
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import synthetic as syn  # noqa: E402
from dvmvs.hip import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variants", default="1,2,16,17,18,19,20,21,22,23,24,25")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-workspace", action="store_true")
    ap.add_argument("--baseline-step", type=int, default=1, help="measurement frame m is keyframe k-(m+1)*step")
    ap.add_argument("--keyframe", type=int, default=8, help="position along the synthetic trajectory (changes the epipolar geometry)")
    ap.add_argument("--two-pass", action="store_true", help="give the sweep the large workspace (spill list, second gather pass)")
    ap.add_argument("--nhwc", action="store_true", help="measurement maps channels-last (DVMVS_LAYOUT_NHWC)")
    ap.add_argument("--real-line", type=int, default=-1, help="use the poses of this line of the sample scene's nmeas+2 keyframe index")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _capi.lib()
    B, C, H, W, D, M = args.batch, 32, 128, 160, 64, args.m
    feats = [torch.cat([syn.smooth_noise((1, C, H, W), seed=300 + 10 * b + i) for b in range(B)]).to(dev) for i in range(M + 1)]
    k = args.keyframe
    traj = torch.from_numpy(syn.synthetic_trajectory(k + 2, seed=1000)).float()
    pose1 = traj[k:k + 1].repeat(B, 1, 1).to(dev)
    pose2s = [traj[k - (i + 1) * args.baseline_step:k - (i + 1) * args.baseline_step + 1].repeat(B, 1, 1).to(dev) for i in range(M)]
    if args.real_line >= 0:
        names = {n: i for i, n in enumerate(syn.sample_image_names())}
        lines = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "indices", "keyframe+hololens-dataset+000+nmeas+2"))]
        ids = [names[x] for x in [l for l in lines if len(l) == 3][args.real_line]]
        allp = torch.from_numpy(syn.sample_poses()).float()
        pose1 = allp[ids[0]:ids[0] + 1].repeat(B, 1, 1).to(dev)
        pose2s = [allp[i:i + 1].repeat(B, 1, 1).to(dev) for i in (ids[1:] * M)[:M]]
    K = syn.scaled_K(syn.full_K(), 2.0).repeat(B, 1, 1).to(dev)
    out = torch.empty(B, D, H, W, device=dev)
    ref_out = torch.empty_like(out)
    ws_bytes = lib.dvmvs_cost_volume_workspace_bytes_two_pass(B, M, H, W, D) if args.two_pass else lib.dvmvs_cost_volume_workspace_bytes(B, M)
    ws = torch.empty((ws_bytes + 3) // 4, device=dev)
    layout = 1 if args.nhwc else 0
    meas = [t.contiguous(memory_format=torch.channels_last) if args.nhwc else t for t in feats[1:]]
    img_ptrs = _capi.pointer_array([t.data_ptr() for t in meas])
    pose_ptrs = _capi.pointer_array([t.data_ptr() for t in pose2s])

    def launch(variant, dst):
        rc = lib.dvmvs_cost_volume_fwd(feats[0].data_ptr(), img_ptrs, pose1.data_ptr(), pose_ptrs, K.data_ptr(), dst.data_ptr(),
                                       B, M, C, H, W, D, 0.25, 20.0, 1, variant, layout, None if args.no_workspace else ws.data_ptr(),
                                       0 if args.no_workspace else ws_bytes, torch.cuda.current_stream().cuda_stream)
        _capi.check(rc, f"variant {variant}")

    layout_saved, layout = layout, 0
    img_ptrs_saved, img_ptrs = img_ptrs, _capi.pointer_array([t.data_ptr() for t in feats[1:]])
    launch(1, ref_out)      # generic kernel on NCHW maps = the cross-check
    torch.cuda.synchronize()
    layout, img_ptrs = layout_saved, img_ptrs_saved
    alg_bytes = (1 + M) * B * C * H * W * 4 + B * D * H * W * 4
    print(f"shape B={B} C={C} {H}x{W} D={D} M={M}; algorithmic bytes {alg_bytes}; |cv| mean {ref_out.abs().mean().item():.4f}")
    for variant in [int(v) for v in args.variants.split(",")]:
        out.zero_()
        try:
            launch(variant, out)
        except RuntimeError as e:
            print(f"variant {variant:3d}: {e}")
            continue
        torch.cuda.synchronize()
        err = (out - ref_out).abs().max().item()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(args.reps):
                launch(variant, out)
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
        print(f"variant {variant:3d}: {best:8.2f} us/launch  {alg_bytes / best / 1e3:8.1f} GB/s  ({100 * alg_bytes / best / 1e3 / 8000:5.2f} % of 8 TB/s)  "
              f"max|diff vs generic| {err:.2e}")


if __name__ == "__main__":
    main()
