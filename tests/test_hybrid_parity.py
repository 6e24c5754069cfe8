"""Hybrid end-to-end parity (GPU): the north star's number -- depth within 1e-4 rel-L1 of the reference forward -- with the
hand-written kernels isolated from everything else.

All pipelines below are oracle/fusionnet_cpu.py's per-frame loop (a restatement of
/root/reference/dvmvs/fusionnet/run-testing.py:151-204) with ALL dense convolutions on the CPU, i.e. the very code (torch
CPU convolutions, same weights) that produced the reference goldens.  The "hybrid" one swaps only the four hot-path
operations -- plane-sweep cost volume, depth re-projection, hidden-state warp, ConvLSTM gates -- for the HIP kernels, called
through the C ABI (tests/hybrid.py).  Whatever separates two depth maps is therefore the hot path's doing and nothing else:
no MIOpen summation order, no BN folding, no graph replay.

What the numbers mean (measured on MI355X, written to gpurun_out/hybrid_parity.json, copied to profiles/):
* The network with the seeded test weights amplifies fp32 round-off: the REFERENCE's own float32 forward is 0.95e-4 / 1.17e-4
  (frames 0 / 1) away from the float64 evaluation of the same network (PINNING_REPORT.json), and from frame 2 on a z-buffer
  pixel decides differently in float32 and float64 (2.4e-3).
* The one piece of reference arithmetic the kernels do NOT reproduce rounding for rounding is the small pose algebra
  (inverse(pose2) @ pose1, K R K^-1, K t): float32 LAPACK on the host in the reference, float64 on the device here
  (csrc/plane_sweep.h).  In float32 the relative translation carries ~5e-7 m of cancellation error, up to ~3e-4 px at the
  0.25 m plane -- 15x the round-off of everything else in the sweep.  ``orc.exact_pose_algebra()`` evaluates the oracle
  with those few matrices in float64, everything else unchanged.

Assertions:
1. hybrid vs the REFERENCE golden depth: <= 1e-4 on frame 0 (the north-star bound as stated), and on every golden frame the
   hybrid is (a) no farther from the reference than the reference is from float64 (+5 %) and (b) closer to float64 than the
   reference is -- i.e. what separates the two is the reference's own round-off.
2. hybrid vs the oracle with exact pose algebra, 9 keyframes of the sample scene's index (3 openings, a tracking loss, lines
   117-118, the wide-baseline / spilling lines 202-204, line 250): <= 1e-5 on every frame whose re-projected low-resolution
   depth estimate (a discrete z-buffer + nearest-sample decision, utils.py:136-154) agrees, <= 1e-4 where a pixel flipped.
   This is the kernels' own deviation.
3. hybrid vs the faithful oracle (reference arithmetic throughout): <= 2e-4 on frames with agreeing estimates -- two float32
   evaluations that differ in the pose algebra only, through a network that amplifies; reported, loosely bounded.
"""
import json
import os

import numpy as np
import pytest
import torch

import dvmvs_oracle as orc
import synthetic as syn

pytestmark = pytest.mark.gpu

REL_L1_NORTH_STAR = 1e-4
REL_L1_HOT_PATH = 1e-5


def rel_l1(d, ref):
    d, ref = np.asarray(d, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.mean(np.abs(d - ref) / ref))


def index_lines():
    names = {n: i for i, n in enumerate(syn.sample_image_names())}
    out = []
    with open(os.path.join(syn.GOLDEN_DIR, "indices", "keyframe+hololens-dataset+000+nmeas+2")) as f:
        for line in f:
            parts = line.split()
            if len(parts) == 3 and all(p in names for p in parts):
                out.append((names[parts[0]], (names[parts[1]], names[parts[2]])))
    return out


def build(hot_path=None):
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    from fusionnet_cpu import CpuDepthPipeline
    ctors = (FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder)
    return CpuDepthPipeline(*syn.build_e2e_modules(ctors), hot_path=hot_path)


def write_report(name, payload):
    folder = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(folder):
        path = os.path.join(folder, "hybrid_parity.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = payload
        with open(path, "w") as f:
            json.dump(data, f, indent=1)


def estimates_agree(a, b):
    """The low-resolution depth estimate is a z-buffer + nearest-sample result: it either agrees to round-off or a pixel
    took its value from a different source point (a visible jump)."""
    a, b = a.numpy(), b.numpy()
    return int(np.sum(np.abs(a - b) > 1e-3 * np.maximum(np.maximum(a, b), 1e-3)))


def test_hybrid_pipeline_matches_the_reference_goldens(hip_device, golden_dir):
    """3 golden frames: hot path on the GPU, convolutions as in the reference run -> depth vs the reference's depth."""
    from hybrid import HipHotPath
    z = np.load(os.path.join(golden_dir, "fusionnet_e2e.npz"))
    hot = HipHotPath(hip_device)
    hybrid = build(hot)
    fullK = syn.full_K()
    rows = []
    for n, (r, ms) in enumerate(syn.E2E_FRAMES):
        depth = hybrid.step(syn.e2e_image(r), syn.pose(r), [syn.e2e_image(i) for i in ms], [syn.pose(i) for i in ms], fullK)
        d = depth[0, ::4, ::4].numpy()
        rows.append({"frame": n, "hybrid_vs_reference": rel_l1(d, z[f"f{n}_depth_sub4"]),
                     "hybrid_vs_float64": rel_l1(d, z[f"f{n}_depth64_sub4"]),
                     "reference_fp32_vs_float64": rel_l1(z[f"f{n}_depth_sub4"], z[f"f{n}_depth64_sub4"])})
        print("golden frame %d: hybrid (HIP hot path + CPU convolutions) depth rel-L1 vs reference %.3e, vs float64 %.3e   "
              "(the reference's own fp32-vs-float64 distance: %.3e)"
              % (n, rows[-1]["hybrid_vs_reference"], rows[-1]["hybrid_vs_float64"], rows[-1]["reference_fp32_vs_float64"]))
    write_report("goldens", rows)
    assert hot.calls["cost_volume"] == 3 and hot.calls["lstm_gates"] == 3 and hot.calls["hidden_warp"] == 2 and hot.calls["depth_reproject"] == 2
    assert rows[0]["hybrid_vs_reference"] <= REL_L1_NORTH_STAR, rows[0]
    for row in rows:
        assert row["hybrid_vs_reference"] <= 1.05 * row["reference_fp32_vs_float64"], row
        assert row["hybrid_vs_float64"] <= row["reference_fp32_vs_float64"], row


def test_hybrid_pipeline_matches_the_oracle_over_index_lines(hip_device):
    """9 keyframes incl. a tracking loss and the spilling wide-baseline lines: hybrid vs all-CPU oracle, frame by frame."""
    from hybrid import HipHotPath
    hot = HipHotPath(hip_device)
    faithful, exact, hybrid = build(), build(), build(hot)
    lines = index_lines()
    schedule = [0, 1, 2, None, 117, 118, 202, 203, 204, 250]   # None = "TRACKING LOST" (run-testing.py:97-101)
    fullK = syn.full_K()
    rows = []
    for item in schedule:
        if item is None:
            for p in (faithful, exact, hybrid):
                p.reset()
            continue
        r, ms = lines[item]
        args = (syn.e2e_image(r), syn.pose(r), [syn.e2e_image(i) for i in ms], [syn.pose(i) for i in ms], fullK)
        rec_f, rec_e, rec_h = {}, {}, {}
        d_f = faithful.step(*args, record=lambda **kw: rec_f.update(kw))
        with orc.exact_pose_algebra():
            d_e = exact.step(*args, record=lambda **kw: rec_e.update(kw))
        d_h = hybrid.step(*args, record=lambda **kw: rec_h.update(kw))
        scale = float(rec_e["cost_volume"].abs().max())
        rows.append({"index_line": item,
                     "hybrid_vs_oracle_exact_poses": rel_l1(d_h.numpy(), d_e.numpy()),
                     "hybrid_vs_oracle_faithful": rel_l1(d_h.numpy(), d_f.numpy()),
                     "cost_volume_rel_diff_exact_poses": float((rec_e["cost_volume"] - rec_h["cost_volume"]).abs().max()) / scale,
                     "cost_volume_rel_diff_faithful": float((rec_f["cost_volume"] - rec_h["cost_volume"]).abs().max()) / scale,
                     "hidden_state_max_abs_diff_exact_poses": float((rec_e["h"] - rec_h["h"]).abs().max()),
                     "flipped_estimate_pixels_exact_poses": estimates_agree(rec_e["depth_estimation"], rec_h["depth_estimation"]),
                     "flipped_estimate_pixels_faithful": estimates_agree(rec_f["depth_estimation"], rec_h["depth_estimation"])})
        row = rows[-1]
        print("index line %3d: hybrid depth rel-L1 vs oracle with exact pose algebra %.3e (cost volume %.1e of its max, %d flipped "
              "estimate pixels) | vs faithful oracle %.3e (cost volume %.1e, %d flipped)"
              % (item, row["hybrid_vs_oracle_exact_poses"], row["cost_volume_rel_diff_exact_poses"], row["flipped_estimate_pixels_exact_poses"],
                 row["hybrid_vs_oracle_faithful"], row["cost_volume_rel_diff_faithful"], row["flipped_estimate_pixels_faithful"]))
    write_report("index_lines", rows)
    assert hot.calls["cost_volume"] == 9
    for row in rows:
        bound = REL_L1_HOT_PATH if row["flipped_estimate_pixels_exact_poses"] == 0 else REL_L1_NORTH_STAR
        assert row["hybrid_vs_oracle_exact_poses"] <= bound, row
        if row["flipped_estimate_pixels_faithful"] == 0:
            assert row["hybrid_vs_oracle_faithful"] <= 2 * REL_L1_NORTH_STAR, row
    assert sum(r["flipped_estimate_pixels_exact_poses"] == 0 for r in rows) >= len(rows) - 2, "the z-buffer decisions should almost always agree"
