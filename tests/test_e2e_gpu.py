"""GPU end-to-end parity: the frame engine (HIP ops + MIOpen convs, BN folded, feature cache, hipGraph replay) against
(a) the per-stage pins captured from the REFERENCE forward on the same inputs/weights (tests/golden/fusionnet_e2e.npz,
pairnet_e2e.npz) and (b) the CPU oracle pipeline.  Headline criterion (BASELINE.json): depth rel-L1
mean(|d - d_ref| / d_ref) <= 1e-4."""
import os

import numpy as np
import pytest
import torch

import hipcall
import synthetic as syn

pytestmark = pytest.mark.gpu


def rel_l1(d, ref):
    return float(np.mean(np.abs(d - ref) / ref))


def pins_close(t, z, prefix, rtol):
    """Sampled entries + sums vs the reference pins; tolerance relative to the tensor's mean magnitude."""
    t = t.detach().float().cpu()
    assert list(t.shape) == list(z[f"{prefix}_shape"]), prefix
    exp = z[f"{prefix}_samples"]
    got = t.reshape(-1)[syn.sample_indices(t.numel())].numpy()
    scale = float(z[f"{prefix}_abs_sum"]) / t.numel() + 1e-12
    err = np.abs(got - exp)
    assert err.mean() <= rtol * scale, (prefix, err.mean(), scale)
    assert err.max() <= 50 * rtol * max(scale, float(np.abs(exp).max())), (prefix, err.max(), scale)


def build(dev, fusion=True, **kw):   # kw: DepthEngine options (fold_bn, cache_features, use_graphs, sequences, ...)
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    ctors = [FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder]
    if not fusion:
        ctors.pop(3)
    mods = syn.build_e2e_modules(tuple(ctors))
    if not fusion:
        mods.insert(3, None)
    return mods, DepthEngine(*mods, device=dev, **kw)


# Tolerances (written here as the north star asks).  Target: depth rel-L1 <= 1e-4 against the reference forward.
#
# What the kernels contribute is pinned in tests/test_hybrid_parity.py (convolutions held fixed on the CPU): ~1e-6, 0 flipped
# z-buffer pixels, on every frame of the 3-frame and the 14-frame reference runs -- since ABI 3 the kernels sample exactly where
# the reference samples.  What is left for the full-GPU engine is the float32 SUMMATION ORDER OF THE CONVOLUTIONS (MIOpen here,
# oneDNN in the reference run; ~50 layers, reductions of up to 9216 terms): given the same inputs and the same recurrent state,
# one frame's depth lands 0.8-1.2e-4 from the reference's (measured; the reference's own float32 forward is 0.95-1.2e-4 from
# the float64 evaluation of the same network, tests/golden/PINNING_REPORT.json).  Two consequences for what can be asserted:
# * frame by frame, from the REFERENCE's state (teacher forcing: previous depth, h, c of the reference run installed before
#   each step): <= ENGINE_VS_REFERENCE on every frame of both golden runs -- the defensible per-frame statement;
# * free-running: the same bound until the low-resolution depth estimate -- a discrete z-buffer + nearest-sample decision fed
#   by the previous DEPTH (utils.py:136-154) -- takes one pixel from a different source point than the reference did, which a
#   1e-4 perturbation of the previous depth does on some frames; from there on the two runs see different inputs and only a
#   sanity bound applies.  Which frame that is depends on MIOpen's algorithm choice; it is reported, not asserted.
REL_L1_TARGET = 1e-4              # the north-star bound: asserted frame by frame from the reference's state (measured 3.7e-5 .. 7.0e-5)
ENGINE_VS_REFERENCE = 1.5e-4      # free-running, while on the reference's inputs: the per-frame difference compounds through h, c (measured <= 8.4e-5)
AFTER_A_FLIPPED_PIXEL = 0.2       # sanity bound once the discrete depth estimate differs (measured up to 0.10 after 60 of 80 pixels flipped)


def flipped_pixels(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return int(np.sum(np.abs(a - b) > 1e-3 * np.maximum(np.maximum(a, b), 1e-3)))


def reference_pipeline(mods):
    """The all-CPU oracle pipeline with the engine's weights: the hybrid tests show it reproduces the reference goldens to ~1e-6
    with the same discrete decisions, so its (full-resolution) state stands in for the reference's where the fixtures only hold
    sub-sampled depth."""
    from fusionnet_cpu import CpuDepthPipeline
    return CpuDepthPipeline(*syn.build_e2e_modules(tuple(type(m) for m in mods)))


@pytest.mark.parametrize("mode", ["eager_unfolded", "graphs_folded_cached"])
def test_fusionnet_three_frames_match_the_reference(hip_device, golden_dir, mode, fixture_host_algebra):
    """Free-running engine over the 3 golden frames (twice in the graph mode, so that both frame kinds are replayed graphs)."""
    from dvmvs.hip import ops
    z = np.load(os.path.join(golden_dir, "fusionnet_e2e.npz"))
    dev = hip_device
    fast = mode == "graphs_folded_cached"
    mods, engine = build(dev, fusion=True, fold_bn=fast, cache_features=fast, use_graphs=fast)
    fullK = syn.full_K()          # poses and intrinsics stay on the host: that is where the engine evaluates the small matrices
    report = []
    for sweep in range(2 if fast else 1):
        engine.reset()
        clean = True
        for n, (r, ms) in enumerate(syn.E2E_FRAMES):
            images = [syn.e2e_image(i).to(dev) for i in ms]
            flipped = 0
            if n > 0:
                _, low = hipcall.depth_reproject(ops, syn.pose(r), syn.pose(syn.E2E_FRAMES[n - 1][0]), prev_depth, fullK.to(dev),
                                                 syn.scaled_K(fullK, 2.0).to(dev), 16)
                flipped = flipped_pixels(low.cpu().numpy(), z[f"f{n}_depth_estimation_full"])
            clean = clean and flipped == 0
            depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), images, [syn.pose(i) for i in ms], fullK,
                                frame_id=r, measurement_ids=list(ms))
            s = engine._static
            pins_close(s["ref_half"], z, f"f{n}_feat_half", 2e-5)
            d = depth[0, ::4, ::4].cpu().numpy().astype(np.float64)
            ref32, ref64 = z[f"f{n}_depth_sub4"].astype(np.float64), z[f"f{n}_depth64_sub4"]
            prev_depth = depth.clone().view(1, 1, 256, 320)
            h_got = s["h"].detach().float().cpu().reshape(-1)[syn.sample_indices(s["h"].numel())].numpy()
            h_err = float(np.abs(h_got - z[f"f{n}_h_samples"]).mean() / (np.abs(z[f"f{n}_h_samples"]).mean() + 1e-12))
            report.append((sweep, n, rel_l1(d, ref32), rel_l1(d, ref64), rel_l1(ref32, ref64), flipped, clean, h_err))
    for row in report:
        print("%s sweep %d frame %d: rel-L1 vs reference %.3e, vs float64 %.3e (reference vs float64 %.3e), flipped estimate pixels: %d "
              "(run still on the reference's inputs: %s), hidden state rel err %.2e" % ((mode,) + row))
    for sweep, n, vs_ref, vs_f64, ref_vs_f64, flipped, clean, h_err in report:
        if clean:
            assert h_err <= 1e-3, f"sweep {sweep} frame {n} ({mode}): hidden state differs from the reference by {h_err:.2e}"
            assert vs_ref <= ENGINE_VS_REFERENCE, f"sweep {sweep} frame {n} ({mode}): depth rel-L1 {vs_ref:.3e} vs the reference"
        else:
            assert vs_ref <= AFTER_A_FLIPPED_PIXEL, f"sweep {sweep} frame {n} ({mode}): depth rel-L1 {vs_ref:.3e} vs the reference"
    assert report[0][2] <= ENGINE_VS_REFERENCE


def run_teacher_forced(engine, dev, tag, frames, state, golden_estimate):
    """Every frame from the REFERENCE's own state: (h, c, previous depth) of tests/golden/fusionnet_state.npz -- the full-resolution
    tensors the reference's modules returned in the golden run -- and the previous pose are installed in the engine before the step.
    ``frames``: (reference pose index, measurement indices) or None = tracking loss.  Returns [(step, engine-vs-reference rel-L1 at
    full resolution, pixels of the 8x10 depth estimate on which the engine's z-buffer decision differs from the reference's)]."""
    from dvmvs.hip import ops
    fullK = syn.full_K()
    fullK_dev, halfK_dev = fullK.to(dev), syn.scaled_K(fullK, 2.0).to(dev)
    rows, previous = [], None
    for n, item in enumerate(frames):
        if item is None:
            engine.reset()
            previous = None
            continue
        r, ms = item
        flipped = 0
        if previous is not None:
            k, r_prev = previous
            prev_depth = torch.from_numpy(state[f"{tag}{k}_depth"]).to(dev).view(1, 1, 256, 320)
            engine.load_state(torch.from_numpy(state[f"{tag}{k}_h"]).to(dev), torch.from_numpy(state[f"{tag}{k}_c"]).to(dev), prev_depth, syn.pose(r_prev))
            _, low = hipcall.depth_reproject(ops, syn.pose(r), syn.pose(r_prev), prev_depth, fullK_dev, halfK_dev, 16)
            flipped = flipped_pixels(low.cpu().numpy(), golden_estimate(n))
        depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                            frame_id=r, measurement_ids=list(ms))
        if previous is not None and engine._direct_buffers:      # the estimate the engine's own frame body fed its ConvLSTM
            flipped = max(flipped, flipped_pixels(engine._direct_buffers["estimate"].cpu().numpy(), golden_estimate(n)))
        rows.append((n, rel_l1(depth[0].cpu().numpy().astype(np.float64), state[f"{tag}{n}_depth"].astype(np.float64)), flipped))
        previous = (n, r)
    return rows


@pytest.mark.parametrize("mode", ["eager_unfolded", "graphs_folded_cached"])
def test_fusionnet_frames_from_the_reference_state(hip_device, golden_dir, mode, fixture_host_algebra):
    """Teacher forcing over the 3 golden frames and the 14-keyframe reference run (tracking loss, wide-baseline lines 200-204,
    249-251): given the reference's inputs and the reference's OWN state, EVERY one of the 17 frames is within the north-star
    bound (1e-4 depth rel-L1, full resolution) of the reference's depth, and the engine's z-buffer decision on the reference's
    previous depth has 0 flipped pixels.  (Round 3 could assert engine-vs-golden on 4-5 of the long run's frames only: the
    fixtures held sub-sampled depth and the full-resolution state came from a CPU stand-in that itself left the golden run.)"""
    dev = hip_device
    fast = mode == "graphs_folded_cached"
    z3 = np.load(os.path.join(golden_dir, "fusionnet_e2e.npz"))
    zl = np.load(os.path.join(golden_dir, "fusionnet_long.npz"))
    state = np.load(os.path.join(golden_dir, "fusionnet_state.npz"))
    lines = syn.keyframe_index_lines(2)
    runs = [("3 golden frames", "f", list(syn.E2E_FRAMES), lambda n: z3[f"f{n}_depth_estimation_full"]),
            ("long reference run", "s", [None if i is None else lines[i] for i in syn.LONG_SCHEDULE], lambda n: zl[f"s{n}_depth_estimation"])]
    checked = 0
    for name, tag, frames, golden_estimate in runs:
        mods, engine = build(dev, fusion=True, fold_bn=fast, cache_features=fast, use_graphs=fast)
        for n, vs_reference, flipped in run_teacher_forced(engine, dev, tag, frames, state, golden_estimate):
            print("%s, %s step %2d: engine depth rel-L1 vs the reference (full resolution, reference's state installed) %.3e, flipped estimate pixels %d"
                  % (mode, name, n, vs_reference, flipped))
            assert vs_reference <= REL_L1_TARGET, (name, n, vs_reference)
            assert flipped == 0, (name, n, flipped)
            checked += 1
    assert checked == 17


def test_fusionnet_long_reference_run(hip_device, golden_dir, fixture_host_algebra):
    """Free-running engine as benchmarked (BN folded, feature cache, hipGraph replay) over the REFERENCE's 14-keyframe run
    (tests/golden/fusionnet_long.npz).  Per frame: depth rel-L1 vs the reference and the number of low-resolution estimate pixels
    that differ from the reference's; tight bound while the run is on the reference's inputs, sanity bound after the first
    flipped pixel until the next restart (see the tolerance note above)."""
    import json
    from dvmvs.hip import ops
    z = np.load(os.path.join(golden_dir, "fusionnet_long.npz"))
    dev = hip_device
    mods, engine = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    lines = syn.keyframe_index_lines(2)
    fullK = syn.full_K()
    fullK_dev, halfK_dev = fullK.to(dev), syn.scaled_K(fullK, 2.0).to(dev)
    rows, previous, clean = [], None, True
    for n, item in enumerate(syn.LONG_SCHEDULE):
        if item is None:
            engine.reset()
            previous, clean = None, True
            continue
        r, ms = lines[item]
        flipped = 0
        if previous is not None:
            _, low = hipcall.depth_reproject(ops, syn.pose(r), previous[0], previous[1], fullK_dev, halfK_dev, 16)
            flipped = flipped_pixels(low.cpu().numpy(), z[f"s{n}_depth_estimation"])
        clean = clean and flipped == 0
        depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                            frame_id=r, measurement_ids=list(ms))
        previous = (syn.pose(r), depth.clone().view(1, 1, 256, 320))
        rows.append({"step": n, "index_line": item, "engine_vs_reference": rel_l1(depth[0, ::4, ::4].cpu().numpy().astype(np.float64),
                                                                                  z[f"s{n}_depth_sub4"].astype(np.float64)),
                     "flipped_estimate_pixels_vs_reference": flipped, "on_reference_inputs": clean})
        print("long run step %2d (index line %3d): engine depth rel-L1 vs reference %.3e, %d flipped estimate pixels%s"
              % (n, item, rows[-1]["engine_vs_reference"], flipped, "" if clean else "  [after a flipped pixel]"))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "engine_long_run.json"), "w") as f:
            json.dump(rows, f, indent=1)
    for row in rows:
        assert row["engine_vs_reference"] <= (ENGINE_VS_REFERENCE if row["on_reference_inputs"] else AFTER_A_FLIPPED_PIXEL), row
    # how many frames run on the reference's inputs before the first near-tie of the z-buffer falls the other way depends on the
    # engine's fp32 summation orders (4-5 frames with MIOpen's Winograd kernels in the decoder, 3 with the direct convolutions: lines
    # 2 and 118 flip one pixel each); what must hold is the first frame of both segments and the step after the first one -- every
    # frame's parity from the reference's own state is asserted above (test_fusionnet_frames_from_the_reference_state)
    assert sum(r["on_reference_inputs"] for r in rows) >= 3


def test_exact_pose_algebra_engine(hip_device, golden_dir):
    """The opt-in "exact" mode (fp64 pose algebra on the device, inside the captured frame, no host involvement): runs through graphs,
    and lands where round 2's engine did -- within the reference's own fp32-vs-float64 distance of the reference (frames 0 and 1 of the
    golden run), i.e. as close to float64 as the reference is, not bit-close to the reference."""
    z = np.load(os.path.join(golden_dir, "fusionnet_e2e.npz"))
    dev = hip_device
    mods, engine = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True, pose_algebra="exact")
    assert engine.pose_algebra == "exact"
    fullK = syn.full_K()
    for sweep in range(2):       # second sweep: replayed graphs
        engine.reset()
        for n, (r, ms) in enumerate(syn.E2E_FRAMES[:2]):
            depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                                frame_id=r, measurement_ids=list(ms))
            d = depth[0, ::4, ::4].cpu().numpy().astype(np.float64)
            ref32, ref64 = z[f"f{n}_depth_sub4"].astype(np.float64), z[f"f{n}_depth64_sub4"]
            print(f"exact mode sweep {sweep} frame {n}: vs reference {rel_l1(d, ref32):.3e}, vs float64 {rel_l1(d, ref64):.3e} "
                  f"(reference vs float64 {rel_l1(ref32, ref64):.3e})")
            assert rel_l1(d, ref64) <= 1.5 * rel_l1(ref32, ref64) + 2e-5
            assert rel_l1(d, ref32) <= 3e-4


def test_pairnet_frame_matches_the_reference(hip_device, golden_dir, fixture_host_algebra):
    z = np.load(os.path.join(golden_dir, "pairnet_e2e.npz"))
    dev = hip_device
    mods, engine = build(dev, fusion=False, fold_bn=True, cache_features=False, use_graphs=False)
    depth = engine.step(syn.e2e_image(12).to(dev), syn.pose(12), [syn.e2e_image(9).to(dev)], [syn.pose(9)], syn.full_K())
    err = rel_l1(depth[0, ::4, ::4].cpu().numpy(), z["depth_sub4"])
    print(f"pairnet depth rel-L1 vs reference {err:.3e}")
    assert err <= ENGINE_VS_REFERENCE


def test_engine_matches_cpu_oracle_pipeline_stage_by_stage(hip_device):
    """Same weights, same inputs, free-running: HIP engine vs oracle/fusionnet_cpu.py, including the state reset rule.  Tight while
    both runs feed their ConvLSTMs the same discrete depth estimate, sanity-bounded after a flipped pixel (tolerance note above)."""
    from dvmvs.hip import ops
    dev = hip_device
    mods, engine = build(dev, fusion=True, fold_bn=False, cache_features=True, use_graphs=False)
    cpu = reference_pipeline(mods)
    fullK = syn.full_K()
    fullK_dev, halfK_dev = fullK.to(dev), syn.scaled_K(fullK, 2.0).to(dev)
    frames = list(syn.E2E_FRAMES) + [None, (13, (12, 10)), (16, (13, 12))]   # None = "TRACKING LOST"
    previous, clean, tight = None, True, 0
    for item in frames:
        if item is None:
            engine.reset()
            cpu.reset()
            previous, clean = None, True
            continue
        r, ms = item
        rec = {}
        cpu.step(syn.e2e_image(r), syn.pose(r), [syn.e2e_image(i) for i in ms], [syn.pose(i) for i in ms], fullK,
                 record=lambda **kw: rec.update(kw))
        if previous is not None:
            _, low = hipcall.depth_reproject(ops, syn.pose(r), previous[0], previous[1], fullK_dev, halfK_dev, 16)
            clean = clean and flipped_pixels(low.cpu().numpy(), rec["depth_estimation"].numpy()) == 0
        depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms],
                            [syn.pose(i) for i in ms], fullK, frame_id=r, measurement_ids=list(ms))
        previous = (syn.pose(r), depth.clone().view(1, 1, 256, 320))
        err = rel_l1(depth.cpu().numpy(), rec["depth"].numpy())
        print(f"engine vs CPU pipeline, frame {item}: depth rel-L1 {err:.3e}" + ("" if clean else "  [after a flipped estimate pixel]"))
        assert err <= (ENGINE_VS_REFERENCE if clean else AFTER_A_FLIPPED_PIXEL), (item, err)
        if clean:
            assert (engine._static["h"].cpu() - rec["h"]).abs().mean().item() <= 2e-3 * rec["h"].abs().mean().item()
        tight += clean
    assert tight >= 2      # the first frame of the run and the first one after the tracking loss are always on equal inputs


def test_lockstep_sequences_equal_single_sequence_runs(hip_device):
    """S independent sequences on one engine (batch S): the HIP hot-path ops are bit-identical per sequence to their batch-1
    runs (same code path per batch item), and each sequence's depth equals what a one-sequence engine computes for it -- up to
    MIOpen choosing a different float32 convolution algorithm for batch S than for batch 1 (bounded like every other
    two-float32-evaluations comparison here).  Includes a per-sequence tracking loss."""
    from dvmvs.hip import ops
    dev = hip_device
    S = 3
    fullK = syn.full_K()
    frames = [(9, (6, 0)), (10, (9, 6)), (11, (9, 10)), (12, (11, 9))]
    image = lambda s, i: syn.smooth_noise((1, 3, 256, 320), seed=7000 + 100 * s + i).to(dev)

    # op level, bit for bit: cost volume, re-projection, hidden warp, gates at batch S vs batch 1
    feats = [syn.smooth_noise((S, 32, 128, 160), seed=90 + i).to(dev) for i in range(3)]
    poses = [torch.cat([syn.pose(9 + 3 * s + i) for s in range(S)]) for i in range(3)]
    halfK = syn.scaled_K(syn.full_K(), 2.0).repeat(S, 1, 1)
    cv = hipcall.cost_volume(ops, feats[0], feats[1:], poses[0], poses[1:], halfK, 0.25, 20.0, 64, True, 0)
    for s in range(S):
        one = hipcall.cost_volume(ops, feats[0][s:s + 1], [f[s:s + 1] for f in feats[1:]], poses[0][s:s + 1], [p[s:s + 1] for p in poses[1:]],
                                  halfK[s:s + 1], 0.25, 20.0, 64, True, 0)
        assert torch.equal(one[0], cv[s])
    cc, c0 = torch.randn(S, 2048, 8, 10, device=dev), torch.randn(S, 512, 8, 10, device=dev)
    h_b, c_b = ops.lstm_gates(cc, c0)
    for s in range(S):
        h_1, c_1 = ops.lstm_gates(cc[s:s + 1], c0[s:s + 1])
        assert torch.equal(h_1[0], h_b[s]) and torch.equal(c_1[0], c_b[s])

    # engine level
    mods, batched = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True, sequences=S)
    singles = [build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=False)[1] for _ in range(S)]
    fresh = [True] * S                   # sequences whose next frame has no previous frame
    for n, (r, ms) in enumerate(frames):
        if n == 2:                       # sequence 1 loses tracking before its third frame
            batched.reset(sequence=1)
            singles[1].reset()
            fresh[1] = True
        ref = torch.cat([image(s, r) for s in range(S)])
        meas = [torch.cat([image(s, i) for s in range(S)]) for i in ms]
        pose = torch.cat([syn.pose(r + 20 * s) for s in range(S)])
        mposes = [torch.cat([syn.pose(i + 20 * s) for s in range(S)]) for i in ms]
        # teacher forcing: every one-sequence engine starts the frame from the batched run's state of its sequence, so that a
        # per-sequence state bug in the batched engine (wrong previous pose after reset(sequence), state leaking between
        # sequences) cannot hide behind the loose bound a free-running comparison needs after a flipped z-buffer pixel
        if n > 0:
            h, c, prev_depth, prev_pose = batched.state()
            for s in range(S):
                if not fresh[s]:
                    singles[s].load_state(h[s:s + 1], c[s:s + 1], prev_depth[s:s + 1], prev_pose[s:s + 1])
        depth = batched.step(ref, pose, meas, mposes, fullK.repeat(S, 1, 1), frame_id=r, measurement_ids=list(ms)).clone()
        for s in range(S):
            d1 = singles[s].step(ref[s:s + 1], pose[s:s + 1], [m[s:s + 1] for m in meas], [p[s:s + 1] for p in mposes], fullK,
                                 frame_id=r, measurement_ids=list(ms)).clone()
            err = rel_l1(depth[s].cpu().numpy(), d1[0].cpu().numpy())
            print(f"lockstep frame {n} sequence {s}: depth rel-L1 vs a one-sequence engine started from the same state {err:.3e}")
            assert err <= ENGINE_VS_REFERENCE, (n, s, err)
            fresh[s] = False


def test_lookahead_is_bit_identical(hip_device):
    """DepthEngine.step(next_reference_image=..., [next_reference_pose=..., next_measurement_poses=..., next_measurement_ids=...]): the
    next keyframe's MnasNet + FPN features -- and, with its poses, its plane sweep and encoder -- are computed during the current step
    on a second stream, concurrently with the current frame's ConvLSTM and decoder (tools/frame_stage_probe.py).  Same kernels on the
    same inputs: every depth map and the recurrent state must equal, bit for bit, what an engine without look-ahead produces --
    through eager frames and replayed graphs, both buffer sets, both sweep configurations, a tracking loss, and an announced next
    frame that does not come."""
    dev = hip_device
    mods, shallow = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    _, deep = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    _, plain = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    deep.max_lookahead = 2      # (the engine's default stops at the feature extraction: DepthEngine.__init__)
    fullK = syn.full_K()
    lines = syn.keyframe_index_lines(2)
    # (index line or None = tracking loss, announce the next frame?: True, False, or "wrong" = another frame than the one that comes)
    schedule = [(0, True), (1, True), (2, True), (3, True), (4, False), (5, True), (None, None), (117, True), (118, True), (119, "wrong"),
                (200, True), (201, True), (202, True), (203, True), (204, False)]
    frames = [(None if i is None else lines[i], flag) for i, flag in schedule]
    used = {"shallow": 0, "deep": 0}
    held = {}      # the device image of a frame: the SAME tensor when it is announced and when it comes (what a runner hands over)
    image = lambda i: held.setdefault(i, syn.e2e_image(i).to(dev))
    for n, (item, announce) in enumerate(frames):
        if item is None:
            for e in (shallow, deep, plain):
                e.reset()
            continue
        r, ms = item
        args = (image(r), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK)
        upcoming = next((f[0] for f in frames[n + 1:] if f[0] is not None), None)
        kw1, kw2 = {}, {}
        if announce and upcoming is not None:
            nr, nms = upcoming
            nxt = nr if announce is True else nr + 1
            kw1 = dict(next_reference_image=image(nxt), next_frame_id=nxt)
            kw2 = dict(kw1, next_reference_pose=syn.pose(nxt), next_measurement_poses=[syn.pose(i) for i in nms], next_measurement_ids=list(nms))
        ready = {name: (e._prefetched["level"] if e._prefetched and e._prefetched["frame_id"] == r else 0) for name, e in (("shallow", shallow), ("deep", deep))}
        a = shallow.step(*args, frame_id=r, measurement_ids=list(ms), **kw1).clone()
        b = deep.step(*args, frame_id=r, measurement_ids=list(ms), **kw2).clone()
        c = plain.step(*args, frame_id=r, measurement_ids=list(ms)).clone()
        for name in used:
            used[name] += ready[name] > 0
        print(f"step {n} (reference frame {r}): prepared by the previous step: shallow {ready['shallow']}, deep {ready['deep']}; "
              f"identical to the engine without look-ahead: {torch.equal(a, c)}, {torch.equal(b, c)}")
        assert torch.equal(a, c) and torch.equal(b, c), n
        for e in (shallow, deep):
            assert torch.equal(e._static["h"], plain._static["h"]) and torch.equal(e._static["c"], plain._static["c"]), n
    assert used["shallow"] >= 8 and used["deep"] >= 8          # the prepared stages were actually used ...
    assert any(k[4] == 1 and k[5] == 1 for k in shallow._graphs)      # ... through replayed graphs of the steady-state patterns
    assert any(k[4] == 2 and k[5] == 2 for k in deep._graphs)
    assert deep.sweep_variant_counts == plain.sweep_variant_counts


@pytest.mark.parametrize("seed", [11, 12])
def test_lookahead_random_schedules_are_bit_identical(hip_device, seed):
    """Property test: random schedules of keyframes with correct / wrong / missing announcements, an announced frame id
    that then comes with ANOTHER image, tracking losses and changes of the measurement count, through engines at look-ahead levels 1 and 2
    (level 2 is where a newly needed graph is captured while this frame's sweep + encoder are already done: ADVICE r4) -- every depth map
    and the recurrent state equal, bit for bit, those of an engine without look-ahead."""
    dev = hip_device
    rng = np.random.default_rng(seed)
    mods, shallow = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True, max_lookahead=1)
    _, deep = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True, max_lookahead=2)
    _, plain = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True, max_lookahead=0)
    fullK = syn.full_K()
    lines = syn.keyframe_index_lines(2)
    held = {}
    image = lambda i: held.setdefault(i, syn.e2e_image(i).to(dev))
    start = int(rng.integers(0, 200))
    schedule = []      # (index line or None, number of measurement frames)
    for j in range(22):
        if j in (7, 15) and rng.random() < 0.7:
            schedule.append((None, 0))
        schedule.append((start + j, 1 if rng.random() < 0.25 else 2))
    accepted = {"shallow": 0, "deep": 0}
    for n, (li, n_meas) in enumerate(schedule):
        if li is None:
            for e in (shallow, deep, plain):
                e.reset()
            continue
        r, ms = lines[li]
        ms = list(ms)[:n_meas]
        upcoming = next(((lines[l], k) for l, k in schedule[n + 1:] if l is not None), None)
        kw1, kw2 = {}, {}
        what = rng.choice(["right", "right", "right", "none", "wrong_frame", "wrong_image"]) if upcoming is not None else "none"
        if what != "none":
            (nr, nms), nk = upcoming
            nms = list(nms)[:nk]
            if what == "wrong_frame":
                nr = nr + 1
            # "wrong_image": the right frame id, but the image that will come is another tensor (here: a fresh copy of another frame's image)
            announced = image(nr) if what != "wrong_image" else syn.e2e_image(nr + 2).to(dev)
            kw1 = dict(next_reference_image=announced, next_frame_id=nr)
            kw2 = dict(kw1, next_reference_pose=syn.pose(nr), next_measurement_poses=[syn.pose(i) for i in nms], next_measurement_ids=nms)
        args = (image(r), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK)
        for name, e in (("shallow", shallow), ("deep", deep)):
            p = e._prefetched
            accepted[name] += bool(p and p["frame_id"] == r and p["image"].data_ptr() == args[0].data_ptr())
        a = shallow.step(*args, frame_id=r, measurement_ids=ms, **kw1).clone()
        b = deep.step(*args, frame_id=r, measurement_ids=ms, **kw2).clone()
        c = plain.step(*args, frame_id=r, measurement_ids=ms).clone()
        assert torch.equal(a, c), (seed, n, li, what, "level 1")
        assert torch.equal(b, c), (seed, n, li, what, "level 2")
        for e in (shallow, deep):
            assert torch.equal(e._static["h"], plain._static["h"]) and torch.equal(e._static["c"], plain._static["c"]), (seed, n)
    print(f"seed {seed}: prepared stages accepted on {accepted} of {sum(l is not None for l, _ in schedule)} frames; "
          f"graphs: level 1 {len(shallow._graphs)}, level 2 {len(deep._graphs)}; warm-up {deep.graph_memory_report()}")
    assert accepted["shallow"] >= 4 and accepted["deep"] >= 4


def test_planning_a_frame_ahead_is_bit_identical(hip_device):
    """DepthEngine.plan_ahead: the announced next frame's parameter block (pose algebra, sweep configuration, work list) is evaluated
    on a second host thread during this frame's graph launch.  Same functions on the same inputs: depth equals, bit for bit, an engine
    that evaluates every block inside its own step -- through a tracking loss, a wrong announcement and a missing one."""
    dev = hip_device
    mods, ahead = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    _, inline = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    inline.plan_frames_ahead = False
    assert ahead.plan_frames_ahead
    fullK = syn.full_K()
    lines = syn.keyframe_index_lines(2)
    schedule = [(0, True), (1, True), (2, True), (3, True), (4, False), (5, True), (None, None), (117, True), (118, True), (119, "wrong"),
                (200, True), (201, True), (202, True)]
    frames = [(None if i is None else lines[i], flag) for i, flag in schedule]
    for n, (item, announce) in enumerate(frames):
        if item is None:
            ahead.reset()
            inline.reset()
            continue
        r, ms = item
        args = (syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK)
        upcoming = next((f[0] for f in frames[n + 1:] if f[0] is not None), None)
        kw = {}
        if announce and upcoming is not None:
            nxt, nms = upcoming if announce is True else lines[7]
            kw = dict(next_reference_image=syn.e2e_image(nxt).to(dev), next_frame_id=nxt, next_reference_pose=syn.pose(nxt),
                      next_measurement_poses=[syn.pose(i) for i in nms], next_measurement_ids=list(nms))
        a = ahead.step(*args, frame_id=r, measurement_ids=list(ms), **kw).clone()
        b = inline.step(*args, frame_id=r, measurement_ids=list(ms), **kw).clone()
        assert torch.equal(a, b), n
    print(f"parameter blocks taken from the planning thread: {ahead.planned_frames_used} of {sum(f[0] is not None for f in frames)} frames")
    assert ahead.planned_frames_used >= 6 and inline.planned_frames_used == 0


def test_refresh_weights_repacks_every_kernel_copy(hip_device):
    """The MFMA convolution kernels read re-packed copies of the weights (direct, bottleneck, 1x1 -- round 6 --, ConvLSTM), made at first use and
    baked into the captured graphs.  After new parameters are loaded into the engine's modules, ``refresh_weights`` re-packs them in place: the
    engine must then give, bit for bit, what an engine that saw the new parameters from the start gives."""
    from dvmvs.engine import FusedConv2d
    dev = hip_device
    _, used = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    _, fresh = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    fullK = syn.full_K()
    frames = list(syn.E2E_FRAMES) + [(12, (11, 9)), (13, (12, 10))]

    def run(engine):
        engine.reset()
        out = []
        for r, ms in frames:
            out.append(engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                                   frame_id=r, measurement_ids=list(ms)).clone())
        return out

    def perturb(engine):
        n = 0
        for m in (engine.fe, engine.fs, engine.enc, engine.lstm, engine.dec):
            for sub in m.modules():
                if isinstance(sub, FusedConv2d):
                    sub.weight.data.mul_(1.0 + 0.01 * ((n % 5) - 2))
                    n += 1
        engine.lstm.lstm_cell.conv.weight.data.mul_(0.99)
        return n

    before = run(used)                   # packs every kernel copy and captures the graphs with the ORIGINAL weights
    assert perturb(used) == perturb(fresh) > 50
    used.refresh_weights()
    used.clear_feature_cache()
    after, expected = run(used), run(fresh)
    kinds = {"direct": 0, "bottleneck": 0, "pointwise": 0}
    for m in (used.fe, used.fs, used.enc, used.dec):
        for sub in m.modules():
            if isinstance(sub, FusedConv2d):
                kinds["direct"] += bool(sub._direct_packed)
                kinds["bottleneck"] += sub._bottleneck_packed is not None
                kinds["pointwise"] += sub._pointwise_packed is not None
    print(f"layers with packed kernel copies: {kinds}, ConvLSTM: {used._lstm_packed is not None}")
    assert kinds["pointwise"] >= 30 and kinds["direct"] >= 15 and kinds["bottleneck"] >= 8 and used._lstm_packed is not None
    for n, (a, b, c) in enumerate(zip(after, expected, before)):
        assert torch.equal(a, b), n
        assert not torch.equal(a, c), n      # (the perturbation reaches the depth)


@pytest.mark.parametrize("fork_after", [-1, 0, 2, 4])
def test_every_fork_point_of_the_lookahead_is_bit_identical(hip_device, monkeypatch, fork_after):
    """DVMVS_FORK_AFTER (round 6): where the frame graph forks the next frame's feature extraction off -- at the start, behind the sweep, behind
    encoder level k - 1 (default 1, covered by the tests above).  Same kernels on the same inputs: the depth and the recurrent state equal the
    engine without look-ahead bit for bit, through eager frames and replayed graphs on both buffer sets."""
    import dvmvs.engine as engine_module
    monkeypatch.setattr(engine_module, "_FORK_AFTER", fork_after)
    dev = hip_device
    _, ahead = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    _, plain = build(dev, fusion=True, fold_bn=True, cache_features=True, use_graphs=True)
    fullK = syn.full_K()
    lines = syn.keyframe_index_lines(2)[:9]
    held = {}
    image = lambda i: held.setdefault(i, syn.e2e_image(i).to(dev))
    used = 0
    for n, (r, ms) in enumerate(lines):
        args = (image(r), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK)
        kw = {}
        if n + 1 < len(lines):
            kw = dict(next_reference_image=image(lines[n + 1][0]), next_frame_id=lines[n + 1][0])
        used += bool(ahead._prefetched and ahead._prefetched["frame_id"] == r)
        a = ahead.step(*args, frame_id=r, measurement_ids=list(ms), **kw).clone()
        b = plain.step(*args, frame_id=r, measurement_ids=list(ms)).clone()
        assert torch.equal(a, b), (fork_after, n)
        assert torch.equal(ahead._static["h"], plain._static["h"]) and torch.equal(ahead._static["c"], plain._static["c"]), (fork_after, n)
    assert used >= 6 and any(k[4] == 1 and k[5] == 1 for k in ahead._graphs)
