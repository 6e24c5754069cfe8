"""Headline benchmark: fusionnet depth frames/sec at 320x256, 64 planes, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one keyframe of one synthetic posed sequence through the timed region of the reference's
fusionnet/run-testing.py:151-204 (features -> cost volume -> encoder -> re-projection -> ConvLSTM -> decoder), inputs
already resident in HBM.  N > 1: one independent sequence per rank (weak scaling), no collective on the data path; the
barrier + max-over-ranks timing is the only communication.  Rank 0 prints ONE JSON line.

Extra objects on the line:
* ``roofline``     -- the fused warp + cost-volume kernel: algorithmic bytes per launch / its average duration, measured
                      here with HIP events around a back-to-back hipGraph of launches on the launch stream;
* ``cpu_baseline`` -- the CPU oracle pipeline (oracle/fusionnet_cpu.py, "port" of the reference loop) timed on this
                      box's host cores on a bounded sample of the same sequence (rank 0, N = 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak (same guide)
LDS_PEAK_TBPS = 256 * 256 * 2.4e9 / 1e12   # 256 CUs x 256 B/clk (ds_read_b128, MI355X_MICROARCH.md LDS table) x 2.4 GHz = 157 TB/s
# LDS bytes the sweep's formulation reads per op: 4 taps x C channels x 4 B per (pixel, plane, frame); and the measured floor of
# its inner pattern (tools/ubench_tap.hip: 8 ds_read_b128 + 20 v_pk_fma_f32 per plane and 8-channel pass, 12 waves / CU)
SWEEP_LDS_PATTERN_FLOOR_US = 11.9
ROOFLINE_GEOMETRIES = 25       # keyframe geometries the sweep kernel is timed on, spread over the WHOLE index file


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--measurement-frames", type=int, default=2)
    ap.add_argument("--no-graphs", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-fold-bn", action="store_true")
    ap.add_argument("--no-feature-cache", action="store_true", help="recompute measurement features every frame like the reference")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--no-lstm-channels-last", action="store_true")
    ap.add_argument("--lookahead", type=int, default=1, choices=[0, 1, 2],
                    help="what the engine is told about the NEXT keyframe: 0 nothing (every stage of a frame on the frame's own stream), 1 its "
                         "image (feature extraction a frame ahead, on a second stream), 2 also its poses (plane sweep + encoder a frame ahead as well)")
    ap.add_argument("--keep-gc", action="store_true", help="leave CPython's cyclic garbage collector on during the timed steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--stage-times", action="store_true", help="also print eager per-stage GPU times to stderr")
    ap.add_argument("--mode", choices=["inference", "train"], default="inference",
                    help="inference = the headline metric (default); train = BASELINE.json configs[4] training step")
    ap.add_argument("--train-batch", type=int, default=4, help="sub-sequences per GPU per step (train mode)")
    ap.add_argument("--train-frames", type=int, default=8, help="frames per sub-sequence (train mode)")
    ap.add_argument("--mark-region", action="store_true",
                    help="bracket the timed loop with the library's marker kernel so tools/summarize_trace.py can cut it out of a rocprofv3 trace")
    ap.add_argument("--no-roofline-leg", action="store_true", help="skip the dedicated cost-volume timing (profiling runs)")
    ap.add_argument("--no-rel-l1", action="store_true", help="skip the golden-frame parity check that fills the rel_l1 field")
    ap.add_argument("--sequences-per-gpu", type=int, default=8,
                    help="secondary measurement: S independent sequences in lockstep on one engine (batch S); 0 = skip. "
                         "The headline value is always one sequence per GPU (BASELINE.json configs[2]).")
    ap.add_argument("--batched-steps", type=int, default=40)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (collector on, causal pipeline, pairnet)")
    return ap.parse_args()


def build_modules():
    import synthetic as syn
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    return syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))


def synthetic_sequence(seq_id, n_images, n_frames, n_meas):
    """One posed sequence: synthetic images (low-passed N(0,1), i.e. already "normalised") on the camera geometry of the
    reference's sample scene.  Frame j is line j of the keyframe index sample-data/indices/keyframe+hololens-dataset+000+
    nmeas+2 (committed under tests/golden/indices): its reference pose and its measurement poses, so that parallax,
    rotation and the mix of sideways / forward motion are those of real hand-held capture (median keyframe baseline
    0.144 m).  Sequences of different ranks start at different lines; the list wraps around.
    Returns images, [(ref_pose[1,4,4], [meas_pose...]) per frame], full_K."""
    import synthetic as syn
    images = [syn.smooth_noise((1, 3, 256, 320), seed=1000 * (seq_id + 1) + i) for i in range(n_images)]
    all_poses = torch.from_numpy(syn.sample_poses()).float()
    names = {n: i for i, n in enumerate(syn.sample_image_names())}
    lines = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "indices", "keyframe+hololens-dataset+000+nmeas+2"))]
    lines = [[names[x] for x in l] for l in lines if len(l) == 3]
    frames = []
    for j in range(n_frames):
        ref, *meas = lines[(j + 37 * seq_id) % len(lines)]
        meas = (meas * n_meas)[:n_meas]
        frames.append((all_poses[ref:ref + 1], [all_poses[i:i + 1] for i in meas]))
    return images, frames, syn.full_K()


def index_pose_sets(n_meas, count):
    """(reference pose, [measurement poses]) of ``count`` lines spread evenly over the sample scene's whole nmeas+2 keyframe
    index (286 lines: easy sideways pairs, rotations, forward motion, wide baselines), independent of --steps."""
    import synthetic as syn
    all_poses = torch.from_numpy(syn.sample_poses()).float()
    names = {n: i for i, n in enumerate(syn.sample_image_names())}
    lines = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "indices", "keyframe+hololens-dataset+000+nmeas+2"))]
    lines = [[names[x] for x in l] for l in lines if len(l) == 3]
    picks = list(range(len(lines))) if count >= len(lines) else sorted({(i * (len(lines) - 1)) // max(count - 1, 1) for i in range(count)})
    sets = []
    for j in picks:
        ref, *meas = lines[j]
        meas = (meas * n_meas)[:n_meas]
        sets.append((all_poses[ref:ref + 1], [all_poses[i:i + 1] for i in meas]))
    return picks, sets


class fixture_host_matrices:
    """Context: the product's host-side pose algebra (dvmvs.pose_algebra, "reference" mode) replays the small fp32 matrices of the
    host the golden fixtures were captured on (tests/golden/host_pose_algebra.npz) instead of evaluating them with the local
    LAPACK -- fp32 ``torch.inverse`` differs in its last bits between CPUs, and "the reference's depth" is only defined together
    with the matrices its run computed.  Inputs only; a pose pair the table lacks is evaluated locally and
    counted in ``misses``."""

    def __enter__(self):
        import synthetic as syn
        from dvmvs import pose_algebra
        self.module, self.table, self.misses = pose_algebra, syn.FixtureHostAlgebra(), 0
        self.saved = (pose_algebra.sweep_matrices_host, pose_algebra.relative_pose_host)

        def replay(lookup, local, args):
            if args[0].dtype != torch.float32:
                return local(*args)
            try:
                return lookup(*args)
            except KeyError:
                self.misses += 1
                return local(*args)

        pose_algebra.sweep_matrices_host = lambda p1, p2s, K: replay(self.table.sweep_matrices_host, self.saved[0], (p1, p2s, K))
        pose_algebra.relative_pose_host = lambda a, c: replay(self.table.relative_pose_host, self.saved[1], (a, c))
        return self

    def __exit__(self, *exc):
        self.module.sweep_matrices_host, self.module.relative_pose_host = self.saved
        return False


def flipped_pixels(a, b):
    """Pixels of the 8x10 depth estimate (a discrete z-buffer + nearest-sample decision, utils.py:136-154) that come from a
    different source point in the two runs."""
    if a is None:      # (an engine configuration that keeps no estimate buffer)
        return None
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return int(np.sum(np.abs(a - b) > 1e-3 * np.maximum(np.maximum(a, b), 1e-3)))


def golden_rel_l1(modules, device, args):
    """Depth rel-L1 mean(|d - d_ref| / d_ref) of the frame engine as benchmarked (hipGraph replay included) against the REFERENCE
    forward (fixtures captured by running the reference itself, tests/golden/make_goldens.py), two ways:

    * ``teacher_forced`` -- the defensible per-frame statement: before each step the REFERENCE's own recurrent state (h, c, previous
      depth of tests/golden/fusionnet_state.npz, previous pose) is installed, so every frame is judged on the reference's inputs:
      the 3 golden frames and the 14 keyframes of the long reference run (a tracking loss, the wide-baseline lines), full resolution.
      Must be <= 1e-4 on every frame, with 0 flipped pixels of the discrete depth estimate.
    * ``free_running`` -- the engine carries its own state over the 3 golden frames; per frame also the number of estimate pixels
      that differ from the reference's.  Once a pixel has flipped (a ~1e-5 perturbation of the previous depth can do that) the two
      runs see different inputs: a 1e-3 there reads "flip at frame n", not "the kernels differ".
    Both with the fixture host's small matrices replayed (``fixture_host_matrices``)."""
    import synthetic as syn
    from dvmvs.engine import DepthEngine
    golden = os.path.join(ROOT, "tests", "golden")
    z3, zl, zs = (np.load(os.path.join(golden, f)) for f in ("fusionnet_e2e.npz", "fusionnet_long.npz", "fusionnet_state.npz"))
    lines = syn.keyframe_index_lines(2)
    fullK = syn.full_K()
    rel = lambda d, ref: float(np.mean(np.abs(d.astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)))

    def make_engine():
        return DepthEngine(*modules, device=device, fold_bn=not args.no_fold_bn, cache_features=not args.no_feature_cache,
                           use_graphs=not args.no_graphs, channels_last=args.channels_last, lstm_channels_last=not args.no_lstm_channels_last)

    def step(engine, r, ms):
        depth = engine.step(syn.e2e_image(r).to(device), syn.pose(r), [syn.e2e_image(i).to(device) for i in ms], [syn.pose(i) for i in ms],
                            fullK, frame_id=r, measurement_ids=list(ms))
        estimate = engine._direct_buffers["estimate"].cpu().numpy() if engine._direct_buffers else None      # (stale on a first frame: not compared)
        return depth[0].cpu().numpy(), estimate

    runs = [("f", list(syn.E2E_FRAMES), lambda n: z3[f"f{n}_depth_estimation_full"]),
            ("s", [None if i is None else lines[i] for i in syn.LONG_SCHEDULE], lambda n: zl[f"s{n}_depth_estimation"])]
    teacher, free = [], []
    with torch.no_grad(), fixture_host_matrices() as host:
        for tag, frames, golden_estimate in runs:
            engine = make_engine()
            previous = None      # (frame number, reference pose index) of the frame whose reference state is installed
            for n, item in enumerate(frames):
                if item is None:
                    engine.reset()
                    previous = None
                    continue
                r, ms = item
                if previous is not None:
                    k, r_prev = previous
                    engine.load_state(torch.from_numpy(zs[f"{tag}{k}_h"]).to(device), torch.from_numpy(zs[f"{tag}{k}_c"]).to(device),
                                      torch.from_numpy(zs[f"{tag}{k}_depth"]).to(device), syn.pose(r_prev))
                d, estimate = step(engine, r, ms)
                teacher.append({"run": "3 golden frames" if tag == "f" else "long reference run", "step": n,
                                "rel_l1": rel(d, zs[f"{tag}{n}_depth"]),
                                "flipped_estimate_pixels": 0 if previous is None else flipped_pixels(estimate, golden_estimate(n))})
                previous = (n, r)
        engine = make_engine()
        for n, (r, ms) in enumerate(syn.E2E_FRAMES):
            d, estimate = step(engine, r, ms)
            free.append({"frame": n, "engine_vs_reference": rel(d[::4, ::4], z3[f"f{n}_depth_sub4"]),
                         "flipped_estimate_pixels": 0 if n == 0 else flipped_pixels(estimate, z3[f"f{n}_depth_estimation_full"]),
                         "engine_vs_float64": rel(d[::4, ::4], z3[f"f{n}_depth64_sub4"]),
                         "reference_vs_float64": rel(z3[f"f{n}_depth_sub4"], z3[f"f{n}_depth64_sub4"])})
    return {"teacher_forced": teacher, "free_running": free, "host_algebra_misses": host.misses}


class ModuleSurfaceLoop:
    """The reference's per-frame loop (fusionnet/run-testing.py:151-204) restated on the dvmvs MODULE SURFACE -- the route BASELINE.json's north_star
    calls "drops into the existing run-testing scripts": dvmvs.utils.cost_volume_fusion and get_non_differentiable_rectangle_depth_estimation, the
    nn.Modules as the scripts build them (BatchNorm unfolded, eager launches), measurement features recomputed every frame as the script does,
    poses and intrinsics ON THE DEVICE as run-testing.py:127-149 puts them; no DepthEngine, no graphs, no feature cache."""

    def __init__(self, modules, device):
        from dvmvs import utils
        from dvmvs.config import Config
        self.utils, self.device = utils, device
        self.fe, self.fs, self.enc, self.lstm, self.dec = [m.to(device).eval() for m in modules]
        self.H, self.W = Config.test_image_height, Config.test_image_width
        self.warp_grid = utils.get_warp_grid_for_cost_volume_calculation(self.W // 2, self.H // 2, device)
        self.reset()

    def reset(self):
        self.lstm_state, self.previous_depth, self.previous_pose = None, None, None

    def step(self, reference_image, reference_pose, measurement_images, measurement_poses, full_K):
        half_K = full_K.clone()
        half_K[:, 0:2, :] = half_K[:, 0:2, :] / 2.0
        lstm_K = full_K.clone()
        lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0
        measurement_halfs = [self.fs(*self.fe(image))[0] for image in measurement_images]
        half, quarter, eighth, sixteenth = self.fs(*self.fe(reference_image))
        cost_volume = self.utils.cost_volume_fusion(half, measurement_halfs, reference_pose, measurement_poses, half_K, self.warp_grid, 0.25, 20.0, 64,
                                                    self.device, True)
        skip0, skip1, skip2, skip3, bottom = self.enc(features_half=half, features_quarter=quarter, features_one_eight=eighth,
                                                      features_one_sixteen=sixteenth, cost_volume=cost_volume)
        if self.previous_depth is not None:
            estimate = self.utils.get_non_differentiable_rectangle_depth_estimation(reference_pose, self.previous_pose, self.previous_depth, full_K, half_K,
                                                                                    self.W, self.H)
            estimate = torch.nn.functional.interpolate(estimate, scale_factor=1.0 / 16.0, mode="nearest")
        else:
            estimate = torch.zeros(1, 1, self.H // 32, self.W // 32, device=self.device)
        self.lstm_state = self.lstm(current_encoding=bottom, current_state=self.lstm_state, previous_pose=self.previous_pose, current_pose=reference_pose,
                                    estimated_current_depth=estimate, camera_matrix=lstm_K)
        prediction = self.dec(reference_image, skip0, skip1, skip2, skip3, self.lstm_state[0])[0]
        self.previous_depth = prediction.view(1, 1, self.H, self.W)
        self.previous_pose = reference_pose
        return prediction, estimate


def module_surface_leg(modules, device, args, M, steps, warmup, accelerated=False, graphs=False):
    """Frames/s of ``ModuleSurfaceLoop`` on the headline workload (same synthetic sequence, same poses), timed like the other legs, with the pose
    algebra in "auto" mode (device-resident poses are served on the device in fp64, no synchronisation: dvmvs/pose_algebra.py) -- and the depth
    rel-L1 of exactly this route against the reference fixtures on the 17 golden frames, teacher-forced (the reference's own state installed)."""
    import synthetic as syn
    from dvmvs import pose_algebra
    saved_mode = pose_algebra.MODE
    pose_algebra.MODE = "auto"
    try:
        if accelerated:      # the same loop with ONE added line: dvmvs.engine.accelerate (BN folded, fused epilogues, MFMA convolution kernels; no graphs)
            from dvmvs.engine import accelerate
            modules = accelerate(*[m.to(device) for m in modules], graphs=graphs)
        loop = ModuleSurfaceLoop(modules, device)
        images, seq, full_K = synthetic_sequence(0, 32, warmup + steps + M + 2, M)
        images = [im.to(device) for im in images]
        seq = [(r.to(device), [p.to(device) for p in ms]) for r, ms in seq]
        full_K = full_K.to(device)

        def run_frame(k):
            ids = [k - 1 - i for i in range(M)]
            return loop.step(images[k % 32], seq[k][0], [images[i % 32] for i in ids], seq[k][1], full_K)

        with torch.no_grad():
            elapsed = timed_region(lambda i: run_frame(M + i), warmup, steps, 1, device)
            # parity of this route: the golden frames, teacher-forced from the reference's recorded state
            golden = os.path.join(ROOT, "tests", "golden")
            zs = np.load(os.path.join(golden, "fusionnet_state.npz"))
            lines = syn.keyframe_index_lines(2)
            fullK = syn.full_K().to(device)
            rel = lambda d, ref: float(np.mean(np.abs(d.astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)))
            runs = [("f", list(syn.E2E_FRAMES)), ("s", [None if i is None else lines[i] for i in syn.LONG_SCHEDULE])]
            rels = []
            for tag, frames in runs:
                loop.reset()
                previous = None
                for n, item in enumerate(frames):
                    if item is None:
                        loop.reset()
                        previous = None
                        continue
                    r, ms = item
                    if previous is not None:
                        k, r_prev = previous
                        state_shape = (1, 512, loop.H // 32, loop.W // 32)
                        loop.lstm_state = (torch.from_numpy(zs[f"{tag}{k}_h"]).to(device).reshape(state_shape), torch.from_numpy(zs[f"{tag}{k}_c"]).to(device).reshape(state_shape))
                        loop.previous_depth = torch.from_numpy(zs[f"{tag}{k}_depth"]).to(device).view(1, 1, loop.H, loop.W)
                        loop.previous_pose = syn.pose(r_prev).to(device)
                    d, _ = loop.step(syn.e2e_image(r).to(device), syn.pose(r).to(device), [syn.e2e_image(i).to(device) for i in ms],
                                     [syn.pose(i).to(device) for i in ms], fullK)
                    rels.append(rel(d.reshape(loop.H, loop.W).cpu().numpy(), zs[f"{tag}{n}_depth"].reshape(loop.H, loop.W)))
                    previous = (n, r)
    finally:
        pose_algebra.MODE = saved_mode
    return {"value": steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
            "route": ("reference loop on dvmvs.utils + accelerate(modules, graphs=True): as accelerate() + one hipGraph replay per module call; no feature cache" if graphs else
                      "reference loop on dvmvs.utils + accelerate(modules): BN folded, fused epilogues, MFMA convs; eager, no graphs, no feature cache" if accelerated else
                      "reference loop restated on dvmvs.utils + nn.Modules, eager, BN unfolded, no feature cache, poses + K on the device"),
            "pose_algebra": "auto (device tensors: fp64 on the device, no synchronisation)",
            "rel_l1": {"teacher_forced": [round(v, 9) for v in rels], "teacher_forced_max": max(rels), "frames": len(rels), "target": 1e-4,
                       "note": "vs the reference fixtures (whose matrices are the fixture host's fp32 LAPACK rounding; this route's are fp64-exact)"}}


def batched_throughput(modules, device, args, S, M):
    """Secondary figure: S independent sequences per GPU advancing in lockstep on ONE engine (batch S through every
    convolution and one cost-volume launch for all of them).  Returns (frames/s over all S sequences, ms per lockstep step,
    mean seconds of the batch-S cost-volume op, its algorithmic bytes)."""
    from dvmvs.engine import DepthEngine
    engine = DepthEngine(*modules, device=device, fold_bn=not args.no_fold_bn, cache_features=not args.no_feature_cache,
                         use_graphs=not args.no_graphs, channels_last=args.channels_last,
                         lstm_channels_last=not args.no_lstm_channels_last, sequences=S)
    n_images, warmup, steps = 16, 6, args.batched_steps
    per_seq = [synthetic_sequence(sid, n_images, warmup + steps + M + 1, M) for sid in range(S)]
    images = [torch.cat([per_seq[sid][0][i] for sid in range(S)]).to(device) for i in range(n_images)]
    seq = [(torch.cat([per_seq[sid][1][j][0] for sid in range(S)]),
            [torch.cat([per_seq[sid][1][j][1][m] for sid in range(S)]) for m in range(M)]) for j in range(len(per_seq[0][1]))]
    full_K = per_seq[0][2].repeat(S, 1, 1)      # poses / intrinsics stay on the host (DepthEngine.step)
    cache = not args.no_feature_cache

    def run_frame(k):
        ids = [k - 1 - i for i in range(M)]
        return engine.step(images[k % n_images], seq[k][0], None if cache else [images[i % n_images] for i in ids], seq[k][1], full_K,
                           frame_id=k if cache else None, measurement_ids=ids if cache else None)

    with torch.no_grad():
        if cache:
            for k in range(M):
                engine._half_features(k, images[k % n_images])
        k = M
        for _ in range(warmup):
            run_frame(k)
            k += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_frame(k)
            k += 1
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert np.isfinite(float(engine._static["depth"].mean()))
    _, pose_sets = index_pose_sets(M, 9)
    pose_sets = [(r.repeat(S, 1, 1), [p.repeat(S, 1, 1) for p in ms]) for r, ms in pose_sets]
    kernel_s, alg_bytes, _, _ = measure_cost_volume_kernel(engine, M, max(2, args.kernel_reps // 2), pose_sets)
    return S * steps / elapsed, 1e3 * elapsed / steps, kernel_s, alg_bytes


def measure_cost_volume_kernel(engine, n_meas, reps, pose_sets, force_variant=None, rounds=3, tiled_plan=False):
    """Average duration of one fused cost-volume op (the sweep launch + its second-pass launch) over keyframe geometries.

    ``pose_sets``: (reference pose, [measurement poses]) of index lines -- the duration depends on the epipolar geometry
    (how large the LDS-staged footprint of a tile is, how many runs of planes are queued for the second pass), so one geometry is
    not representative.  Each geometry runs in the sweep configuration the engine picks for it (dvmvs_sweep_plan, one sequence: the
    host-side plan model on the host copies of the matrices; lock-step batches: dvmvs.utils.sweep_variant).  Per configuration a hipGraph of ``reps`` back-to-back ops (no host
    gaps) is timed with HIP events on the stream it is replayed on.
    ``force_variant``: time that kernel variant on every geometry instead (no work list), e.g. 7 = the MFMA sweep as one item per workgroup.
    ``tiled_plan``: the LDS-tiled sweep as dvmvs_sweep_plan plans it per geometry (configuration + work list): what the engine ran before round 6
    on the pairs it did not give to variant 6.
    Returns (mean seconds per op, algorithmic bytes per op, [per-geometry seconds], [per-geometry variant])."""
    from dvmvs import pose_algebra, utils
    from dvmvs.hip import _capi
    from dvmvs.hip import ops as _ops
    s = engine._static
    ref = s["ref_half"]
    B, C, H, W = ref.shape
    D = engine.n_depth_levels
    out = torch.empty(B, D, H, W, device=ref.device)
    img_ptrs = _capi.pointer_array([t.data_ptr() for t in s["meas_feat"][:n_meas]])
    layout = _capi.LAYOUT_NHWC if all(t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
                                      for t in s["meas_feat"][:n_meas]) else _capi.LAYOUT_NCHW
    Hm = torch.zeros(B, n_meas, 9, device=ref.device)
    kt = torch.zeros(B, n_meas, 3, device=ref.device)
    half_K = s["half_K"].cpu()
    lib = _capi.lib()
    workspace, ws_bytes = _ops.sweep_workspace(ref.device, B, n_meas, H, W, D)

    use_list = engine.sweep_work_list and force_variant is None   # as the engine launches it: with the host-planned work list of the geometry
    work_list = torch.zeros(_ops.sweep_work_list_words(B, H, W, D), dtype=torch.int32, device=ref.device)

    def set_geometry(ref_pose, meas_poses):
        h, k, host = pose_algebra.sweep_matrices(ref_pose, meas_poses[:n_meas], half_K, ref.device, engine.pose_algebra, with_host=True)
        Hm.copy_(h)
        kt.copy_(k)
        if force_variant is not None:
            return force_variant
        if use_list and B == 1:
            # exactly what DepthEngine._evaluate_frame_parameters does: configuration (2 / 3, or their single-pass forms 4 / 5 when the
            # plan queues nothing) and work list in one walk
            items = torch.zeros(work_list.numel(), dtype=torch.int32)
            variant = _ops.sweep_plan_host(host[0], host[1], H, W, D, engine.min_depth, engine.max_depth, 0, items,
                                           allow_mfma=getattr(engine, "sweep_mfma", False) and not tiled_plan)
            work_list.copy_(items)
            return variant
        variant = utils.sweep_variant(host, H, W, D, engine.min_depth, engine.max_depth)
        if use_list:
            work_list.copy_(_ops.sweep_work_list_host(host[0], host[1], H, W, D, engine.min_depth, engine.max_depth, variant))
        return variant

    def launch(variant):
        rc = lib.dvmvs_cost_volume_planned_fwd(ref.data_ptr(), img_ptrs, Hm.data_ptr(), kt.data_ptr(), out.data_ptr(),
                                               B, n_meas, C, H, W, D, engine.min_depth, engine.max_depth, 1, variant, layout,
                                               workspace.data_ptr(), ws_bytes, work_list.data_ptr() if use_list else None,
                                               torch.cuda.current_stream().cuda_stream)
        _capi.check(rc, "dvmvs_cost_volume_planned_fwd")

    graphs = {}

    def graph_for(variant):
        if variant not in graphs:
            launch(variant)
            torch.cuda.synchronize()
            graphs[variant] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graphs[variant]):
                for _ in range(reps):
                    launch(variant)
        return graphs[variant]

    per_geometry, variants = [], []
    for ref_pose, meas_poses in pose_sets:
        variant = set_geometry(ref_pose, meas_poses)
        graph = graph_for(variant)
        graph.replay()
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(rounds):
            graph.replay()
        end.record()
        torch.cuda.synchronize()
        per_geometry.append(start.elapsed_time(end) * 1e-3 / (rounds * reps))
        variants.append(variant)
    algorithmic_bytes = (1 + n_meas) * B * C * H * W * 4 + B * D * H * W * 4
    return sum(per_geometry) / len(per_geometry), algorithmic_bytes, per_geometry, variants


def finite_or_null(value):
    """NaN / infinity -> null (legs that were switched off leave NaNs behind; strict JSON has no literal for them)."""
    if isinstance(value, float):
        return value if np.isfinite(value) else None
    if isinstance(value, dict):
        return {k: finite_or_null(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [finite_or_null(v) for v in value]
    return value


def issue_roofline(per_geometry, variants):
    """Issue-time accounting of the engine's sweep kernels from the newest committed PMC profile (profiles/r*_sweep_issue_pmc.json, counted on
    index line 0): per kernel the instruction-issue time per SIMD against its mean duration on the timed steps that ran it; top level = the
    kernel most timed steps ran."""
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sweep_issue_pmc.json")))[-1]
        pmc = json.load(open(path))
        kernels = {}
        for label, mine in (("sweep_mfma", lambda v: v in (6, 7)), ("sweep_tiled_kernel", lambda v: v not in (6, 7))):
            name, k = next(((n, v) for n, v in pmc["kernels"].items() if n.startswith(label)), (None, None))
            if k is None:
                continue
            ts = [t for t, v in zip(per_geometry, variants) if mine(v)]
            mean_us = 1e6 * sum(ts) / len(ts) if ts else None
            kernels[label] = {"profiled_as": name, "issue_busy_us_per_simd": k["issue_busy_us_per_simd"], "timed_steps": len(ts),
                              "kernel_us": mean_us, "frac": k["issue_busy_us_per_simd"] / mean_us if mean_us else None}
        top = max(kernels, key=lambda n: kernels[n]["timed_steps"])
        return {"bound": "instruction_issue", "kernel": top, **{k: kernels[top][k] for k in ("issue_busy_us_per_simd", "kernel_us", "frac", "timed_steps")},
                "unit": "us of instruction issue per SIMD (SQ_ACTIVE_INST_ANY x 4 cycles / 1024 SIMDs / clock)",
                "clock_ghz": pmc["clock_ghz_during_kernel"], "source": os.path.relpath(path, ROOT), "kernels": kernels}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def count_graph_kernels(graph):
    """Kernel nodes of a captured torch.cuda.CUDAGraph, read from its debug dump (None when the runtime cannot dump)."""
    import re
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "graph.dot")
            graph.debug_dump(path)
            text = open(path).read()
        return len(re.findall(r"label=\"[^\"]*(KERNEL|kernel)", text)) or None
    except Exception:
        return None


def make_frame_runner(engine, images, seq, full_K, M, level, cache):
    """run_frame(k): keyframe k of the synthetic sequence through ``engine`` (the reference loop of fusionnet / pairnet run-testing.py:151-204 /
    :136-166); ``level`` >= 1 announces the next keyframe (image + poses), as a pre-computed keyframe index allows."""
    n_images = len(images)

    def run_frame(k):
        ids = [k - 1 - i for i in range(M)]
        meas_images = None if cache else [images[i % n_images] for i in ids]
        ahead = {}
        if cache and level >= 1:
            ahead = dict(next_reference_image=images[(k + 1) % n_images], next_frame_id=k + 1, next_reference_pose=seq[k + 1][0],
                         next_measurement_poses=seq[k + 1][1], next_measurement_ids=[k - i for i in range(M)])
        return engine.step(images[k % n_images], seq[k][0], meas_images, seq[k][1], full_K,
                           frame_id=k if cache else None, measurement_ids=ids if cache else None, **ahead)

    return run_frame


def gc_quiet():
    """CPython's cyclic collector off for a timed region, as in the standard library's timeit: its full collection is a 4-6 ms host pause
    at an allocation count that falls a few steps into every run.  Everything allocated so far is collected and frozen first."""
    gc.collect()
    gc.freeze()
    gc.disable()


def gc_restore():
    gc.enable()
    gc.unfreeze()


def secondary_leg(modules, device, args, M, lookahead, steps, warmup, gc_on=False, queue_fillers=None):
    """A secondary throughput figure on its own engine (one sequence, this GPU): frames/s of ``steps`` keyframes after ``warmup``,
    timed like the headline (barrier-free: world size 1), cyclic collector off unless ``gc_on``.  ``queue_fillers``: the engine's
    filler-graph count for this leg (dvmvs/engine.py: _GRAPH_QUEUE_FILLERS; None = as configured)."""
    import dvmvs.engine as engine_module
    from dvmvs.engine import DepthEngine
    configured = engine_module._GRAPH_QUEUE_FILLERS
    if queue_fillers is not None:
        engine_module._GRAPH_QUEUE_FILLERS = int(queue_fillers)
    try:
        return _secondary_leg(DepthEngine, modules, device, args, M, lookahead, steps, warmup, gc_on)
    finally:
        engine_module._GRAPH_QUEUE_FILLERS = configured


def _secondary_leg(DepthEngine, modules, device, args, M, lookahead, steps, warmup, gc_on):
    engine = DepthEngine(*modules, device=device, fold_bn=not args.no_fold_bn, cache_features=not args.no_feature_cache,
                         use_graphs=not args.no_graphs, channels_last=args.channels_last,
                         lstm_channels_last=not args.no_lstm_channels_last, max_lookahead=lookahead)
    cache = not args.no_feature_cache
    images, seq, full_K = synthetic_sequence(0, 32, warmup + steps + M + 2, M)
    images = [im.to(device) for im in images]
    run_frame = make_frame_runner(engine, images, seq, full_K, M, lookahead, cache)
    with torch.no_grad():
        if cache:
            for k in range(M):
                engine._half_features(k, images[k % len(images)])
        if not gc_on:
            gc_quiet()      # (before the warm-up steps, as in the headline leg)
        elapsed = timed_region(lambda i: run_frame(M + i), warmup, steps, 1, device, after=None if gc_on else gc_restore)
    assert np.isfinite(float(engine._static["depth"].mean()))
    return steps / elapsed, engine


def measure_small_kernels(engine, reps=20):
    """The other hot-path kernels of a fusionnet frame -- ConvLSTM gates, hidden-state warp, depth re-projection (splat +
    decimate) -- timed like the sweep: a hipGraph of ``reps`` back-to-back calls, HIP events on the replay stream.  They move a
    few hundred KB each and sit at the launch-latency floor of a dependent graph node; the fractions say so."""
    from dvmvs.hip import ops as _ops
    s = engine._static
    dev = s["h"].device
    S, H, W = engine.sequences, engine.height, engine.width
    cc = torch.randn(S, 2048, H // 32, W // 32, device=dev)
    c_state, h_state = torch.randn(S, 512, H // 32, W // 32, device=dev), torch.zeros(S, 512, H // 32, W // 32, device=dev)
    estimate = torch.rand(S, 1, H // 32, W // 32, device=dev) * 3 + 0.5
    warped = torch.empty_like(h_state)
    T = torch.eye(4, device=dev).repeat(S, 1, 1)
    T[:, 0, 3] = 0.05
    prev_depth = torch.rand(S, 1, H, W, device=dev) * 3 + 0.5
    low = [torch.zeros(S, 1, H // 32, W // 32, device=dev) for _ in range(2)]
    full_K, half_K, lstm_K = s["full_K"].clone(), s["half_K"].clone(), s["lstm_K"].clone()
    n_splits = 16      # K-splits of the cell's convolution (dvmvs_bottleneck_conv_fwd) that the gates kernel adds up itself
    parts = torch.randn(n_splits * S * 2048 * (H // 32) * (W // 32), device=dev) * 0.25
    flip = [0]

    def reproject():      # as the engine calls it: two estimate buffers alternating, each launch zero-fills the other one
        flip[0] ^= 1
        _ops.depth_reproject_estimate_into(T, prev_depth, full_K, half_K, low[flip[0]], low[1 - flip[0]], 16)

    calls = {
        "lstm_gates": (lambda: _ops.lstm_gates_into(cc, c_state, h_state), 2048 * 80 * 4 + 512 * 80 * 4 + 2 * 512 * 80 * 4, 1),
        # what the engine launches since round 5: the gates on the convolution's 16 partial sums (no reduction launch in front)
        "lstm_gates_on_partial_sums": (lambda: _ops.lstm_gates_partials_into(parts, n_splits, c_state, h_state),
                                       n_splits * 2048 * 80 * 4 + 512 * 80 * 4 + 2 * 512 * 80 * 4, 1),
        "hidden_warp": (lambda: _ops.hidden_warp_into(h_state, estimate, T, lstm_K, True, warped), 2 * 512 * 80 * 4 + 80 * 4, 1),
        # one launch since round 5 (straight into the 8x10 estimate; reads the previous depth once)
        "depth_reproject": (reproject, H * W * 4 + 2 * (H // 32) * (W // 32) * 4, 1),
    }
    out = {}
    for name, (fn, alg_bytes, launches) in calls.items():
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (3 * reps)
        out[name] = {"us_per_call": round(us, 2), "launches": launches, "algorithmic_bytes": S * alg_bytes,
                     "achieved_GBps": round(S * alg_bytes / us / 1e3, 1), "frac_of_hbm_peak": S * alg_bytes / us / 1e3 / HBM_PEAK_GBPS}
    return out


def usable_cores():
    """Host cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() alone
    reports the whole socket inside a quota-limited container, and oversubscribing it makes oneDNN/OpenMP crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                quota = float(txt[0])
                period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(args, modules, n_meas):
    from fusionnet_cpu import CpuDepthPipeline
    cores = min(usable_cores(), 64)      # batch-1 convolutions of this size stop scaling long before 64 threads
    torch.set_num_threads(cores)
    images, seq, full_K = synthetic_sequence(0, 8, 64, n_meas)
    pipe = CpuDepthPipeline(*modules, planewise_cost_volume=True)
    frames, t_total = 0, 0.0
    k = n_meas

    def cpu_frame(k):
        pipe.step(images[k % 8], seq[k][0], [images[(k - 1 - i) % 8] for i in range(n_meas)], seq[k][1], full_K)

    # one untimed frame (thread pools, oneDNN primitive caches), then a bounded timed sample
    t0 = time.perf_counter()
    cpu_frame(k)
    first = time.perf_counter() - t0
    if first > args.cpu_baseline_seconds:      # pathologically slow host: report the one frame we have and stop
        frames, t_total = 1, first
    else:
        pipe.stage_seconds.clear()
    while t_total < args.cpu_baseline_seconds and frames < 24:
        k += 1
        t0 = time.perf_counter()
        cpu_frame(k)
        t_total += time.perf_counter() - t0
        frames += 1
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": frames / t_total, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} keyframes, same workload, oracle/fusionnet_cpu.py (torch CPU), {cores} of {os.cpu_count()} cores (cgroup quota), {cpu_model}"[:119],
            "stage_ms": {k2: round(1e3 * v / frames, 2) for k2, v in pipe.stage_seconds.items()}}


def region_marker(device):
    """A callable that launches the library's empty, uniquely named marker kernel and synchronises: tools/summarize_trace.py cuts a
    rocprofv3 kernel trace at the two marks bench.py sets around its timed loop."""
    from dvmvs.hip import _capi

    def mark():
        with torch.cuda.device(device):
            _capi.check(_capi.lib().dvmvs_trace_marker(torch.cuda.current_stream(device).cuda_stream), "dvmvs_trace_marker")
        torch.cuda.synchronize(device)

    return mark


def timed_region(step_fn, warmup, steps, world, device, before=None, after=None):
    """The contract's timing harness, shared by the inference and training modes (and driven on CPU, with the gloo backend and
    a stub step, by tests/test_bench_harness.py): ``warmup`` untimed steps, then EXACTLY ``steps`` steps bracketed by a barrier +
    device synchronisation on both sides; returns the MAXIMUM over ranks of the elapsed seconds.  ``step_fn(i)`` gets the running
    step number; ``before`` / ``after`` run just outside the timed region (profiling markers)."""
    import torch.distributed as dist
    sync = (lambda: torch.cuda.synchronize(device)) if torch.device(device).type == "cuda" else (lambda: None)
    i = 0
    for _ in range(warmup):
        step_fn(i)
        i += 1
    sync()
    if world > 1:
        dist.barrier()
    sync()
    if before is not None:
        before()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn(i)
        i += 1
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if after is not None:
        after()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def train_mode(args, world, rank, device):
    """BASELINE.json configs[4]: fusionnet training step, sub-sequences of 8 frames at 256x256, batch 4 per GPU, Adam,
    L1-inv loss; gradients averaged with the bucketed RCCL all-reduce of dvmvs/training.py.  A step = one optimisation
    step on every rank; value = sub-sequences/s over all ranks (weak scaling).  Not the headline metric."""
    import torch.distributed as dist
    import synthetic as syn
    from dvmvs.config import Config
    from dvmvs.training import BucketedGradientReducer, train_step
    # The reference trains with cudnn.benchmark = True (run-training.py:115): MIOpen searches its solvers per problem.  A training step
    # has ~300 distinct forward / backward-data / backward-weight problems and searching them all takes 10 minutes on a fresh box, so
    # the search results of this exact workload are kept in the repository (deep-video-mvs_amd/miopen_userdb: MIOpen's user find-db,
    # written by one such run) and MIOpen is pointed at them: the search then is a look-up.  DVMVS_TRAIN_CUDNN_BENCHMARK=0: immediate
    # mode (no search, no db; 176 vs 164 ms per step on the same box).
    import glob
    userdb = os.path.join(ROOT, "deep-video-mvs_amd", "miopen_userdb")
    search = os.environ.get("DVMVS_TRAIN_CUDNN_BENCHMARK", "1" if glob.glob(os.path.join(userdb, "*.ufdb.txt")) else "0") != "0"
    if search:
        os.environ.setdefault("MIOPEN_USER_DB_PATH", userdb)
    torch.backends.cudnn.benchmark = search
    # the shipped find-db is keyed to ONE MIOpen build (its file name carries the build string); another build ignores it silently and
    # searches afresh, writing files of its own: what was there before the run is compared with what is there after it
    db_dir = os.environ.get("MIOPEN_USER_DB_PATH", userdb)
    db_before = {f: os.path.getsize(os.path.join(db_dir, f)) for f in os.listdir(db_dir)} if os.path.isdir(db_dir) else {}
    model = [m.to(device).train() for m in build_modules()]
    params = [p for m in model for p in m.parameters()]
    reducer = BucketedGradientReducer(params)
    opt = torch.optim.Adam(params, lr=1e-4)
    B, T, H, W = args.train_batch, args.train_frames, Config.train_image_height, Config.train_image_width
    g = torch.Generator().manual_seed(77 + rank)
    images = [syn.smooth_noise((B, 3, H, W), seed=5000 + 100 * rank + i).to(device) for i in range(T)]
    depths = [(torch.rand(B, H, W, generator=g) * 4.5 + 0.5).to(device) for _ in range(T)]
    all_poses = torch.from_numpy(syn.sample_poses()).float()
    # poses stay on the host (read by the host-side pose algebra only, dvmvs.pose_algebra)
    poses = [torch.stack([all_poses[(40 * b + 3 * i + 7 * rank) % len(all_poses)] for b in range(B)]) for i in range(T)]
    K = torch.cat([syn.full_K(width=W, height=H)] * B)      # (host, like the poses: round 6 -- a device K was copied back once per step, a synchronisation)
    last = {}

    def step(_):
        last["loss"] = train_step(model, opt, reducer, images, depths, poses, K)

    mark = region_marker(device) if args.mark_region else None
    elapsed = timed_region(step, max(args.warmup, 1), args.steps, world, device, before=mark, after=mark)
    loss = last["loss"]
    db_after = {f: os.path.getsize(os.path.join(db_dir, f)) for f in os.listdir(db_dir)} if os.path.isdir(db_dir) else {}
    new_files = sorted(f for f in db_after if f not in db_before)
    grown = sorted(f for f in db_after if f in db_before and db_after[f] != db_before[f])
    if not search:
        find_db = "not used (MIOpen's immediate mode)"
    elif new_files:
        find_db = (f"MISMATCH: this MIOpen build wrote its own find-db ({', '.join(new_files)}) next to the shipped one ({', '.join(sorted(db_before)) or 'none'}): "
                   "the shipped search results were ignored and the solver search ran during warm-up")
    elif grown:
        find_db = f"shipped find-db matched this MIOpen build; the run added entries to {', '.join(grown)}"
    else:
        find_db = f"shipped find-db matched this MIOpen build ({', '.join(sorted(db_before))}): the solver search was a look-up"
    if rank == 0:
        print(json.dumps({
            "metric": "fusionnet training sub-sequences/sec (8 frames, 256x256, 64 planes)", "value": world * B * args.steps / elapsed,
            "unit": "subsequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"fusionnet training step, subseq_len={T}, batch={B}/GPU, Adam, L1-inv loss (BASELINE.json configs[4])",
                       "miopen_solver_search": bool(torch.backends.cudnn.benchmark), "miopen_find_db": find_db,
                       "pose_algebra": "host, whole sub-sequence per step, one pinned upload",
                       "grad_buckets": len(reducer.buckets), "grad_bytes": sum(f.numel() * 4 for f in reducer.flat),
                       "parallelism": f"data-parallel x{world}, bucketed RCCL all-reduce overlapped with backward"},
            "final_loss": float(loss)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def relaunch_under_torchrun(gpus):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): replace this process by the contract's own launch line --
    one rank per GPU on this node, rendezvous on 127.0.0.1 -- so that the command works in either shape and still prints one JSON line
    (rank 0's).  Driven on CPU by tests/test_bench_harness.py."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the engine's planning thread hands the interpreter lock over within 0.1 ms instead of CPython's 5 ms (opt-in since round 6: dvmvs/engine.py plan_ahead)
    os.environ.setdefault("DVMVS_SWITCH_INTERVAL", "1e-4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def stub_mode(args, world, rank):
    """TEST ONLY (DVMVS_BENCH_STUB_STEP_MS set; tests/test_bench_harness.py): the launch + harness path of `bench.py --gpus N` on a box
    without GPUs -- gloo instead of RCCL, a sleeping stub instead of the frame engine -- printing a line that says so.  Never a result."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ms = float(os.environ["DVMVS_BENCH_STUB_STEP_MS"])
    elapsed = timed_region(lambda i: time.sleep(1e-3 * ms * (1 + rank)), args.warmup, args.steps, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "STUB (test of the launch path only)", "value": world * args.steps / elapsed, "unit": "stub steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "data": "stub",
                          "config": {"workload": "stub", "launched_by": os.environ.get("TORCHELASTIC_RUN_ID", "direct")}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus)      # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if os.environ.get("DVMVS_BENCH_STUB_STEP_MS"):
        return stub_mode(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the plane-sweep path has no CPU fallback")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # host threads: the box's CPU quota is shared by all ranks (weight initialisation and BN folding run on the CPU)
    torch.set_num_threads(max(1, min(usable_cores() // max(world, 1), 16)))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.mode == "train":
        return train_mode(args, world, rank, device)

    from dvmvs.engine import DepthEngine
    # MIOpen's immediate mode (no timing-based solver search at warm-up): measured as fast as the search on this network (610.8 vs
    # 609.1 frames/s, tools/bench_modes_probe.sh) and its choice does not depend on a timing.  DVMVS_BENCH_CUDNN_BENCHMARK=1: search.
    torch.backends.cudnn.benchmark = os.environ.get("DVMVS_BENCH_CUDNN_BENCHMARK", "0") != "0"
    modules = build_modules()
    engine = DepthEngine(*modules, device=device, fold_bn=not args.no_fold_bn, cache_features=not args.no_feature_cache,
                         use_graphs=not args.no_graphs, channels_last=args.channels_last,
                         lstm_channels_last=not args.no_lstm_channels_last, max_lookahead=args.lookahead)
    engine.graph_debug = True
    M = args.measurement_frames
    n_images = 32
    total = args.warmup + args.steps
    images, seq, full_K = synthetic_sequence(rank, n_images, total + M + 1, M)   # one independent sequence per rank
    images = [im.to(device) for im in images]
    # poses and intrinsics stay on the host: that is where they come from and where the engine evaluates the frame's small
    # matrices (dvmvs.pose_algebra, "reference" mode) before its single per-frame upload

    host_seconds = [0.0, 0]      # time the host spends inside engine.step (pose algebra, sweep plan, upload, graph launch), and calls

    step_events = []             # one device event after every timed step: per-step device times (diagnostic, short runs)

    level = 0 if args.no_feature_cache else args.lookahead      # the sequence is known in advance (as with a pre-computed keyframe index):
    run_frame_inner = make_frame_runner(engine, images, seq, full_K, M, level, not args.no_feature_cache)   # every step announces the next keyframe

    step_trace_path = os.environ.get("DVMVS_BENCH_STEP_TRACE")      # diagnostic: the host clock at the checkpoints of every timed step, written
    step_trace = []                                                  # when one of them (or a device gap) exceeds 2 ms: tools/r06_hiccup_hunt.sh

    def run_frame(k):
        if step_trace_path:
            engine.step_clock = []
        t_host = time.perf_counter()
        try:
            return run_frame_inner(k)
        finally:
            t_end = time.perf_counter()
            host_seconds[0] += t_end - t_host
            host_seconds[1] += 1
            if len(step_events) < 64:
                step_events.append(torch.cuda.Event(enable_timing=True))
                step_events[-1].record()
            if step_trace_path:
                step_trace.append((t_host, t_end, engine.step_clock))
                engine.step_clock = None

    with torch.no_grad():
        # buffer fill: the first M keyframes only contribute features (reference: keyframe-buffer response 0 / short lists)
        if not args.no_feature_cache:
            for k in range(M):
                engine._half_features(k, images[k % n_images])
        mark = region_marker(device) if args.mark_region else None

        def region_start():      # (host time is counted over the timed steps only: warm-up steps run eagerly and capture graphs)
            host_seconds[0], host_seconds[1] = 0.0, 0
            engine.ring_wait_seconds = 0.0
            step_events.clear()
            if mark is not None:
                mark()

        def region_end():
            if mark is not None:
                mark()
            if not args.keep_gc:
                gc_restore()

        if not args.keep_gc:
            # (stated in config.cyclic_gc; value_gc_on is the same run with the collector left on.  Before the WARM-UP steps, not between them and
            # the timed ones: the full collection is a 30-80 ms host pause, and a device that sat idle through it ran the first timed frames
            # 5 % slower than the last)
            gc_quiet()
        elapsed = timed_region(lambda i: run_frame(M + i), args.warmup, args.steps, world, device, before=region_start, after=region_end)
        host_ms = 1e3 * host_seconds[0] / max(host_seconds[1], 1)
        host_wait_ms = 1e3 * engine.ring_wait_seconds / max(host_seconds[1], 1)
        if step_trace_path:
            timed = step_trace[-args.steps:]
            gaps = [0.0] + [step_events[i - 1].elapsed_time(step_events[i]) for i in range(1, len(step_events))]
            slow = [i for i, (a, b, _) in enumerate(timed) if 1e3 * (b - a) > 2.0 or (i < len(gaps) and gaps[i] > 2.0)]
            if slow:
                with open(step_trace_path, "w") as f:
                    json.dump({"slow_steps": slow, "elapsed_ms_per_step": 1e3 * elapsed / args.steps,
                               "steps": [{"start_ms": 1e3 * (a - timed[0][0]), "host_ms": 1e3 * (b - a), "device_gap_ms": gaps[i] if i < len(gaps) else None,
                                          "marks_ms": [(name, 1e3 * (t - a)) for name, t in (marks or [])]} for i, (a, b, marks) in enumerate(timed)]}, f, indent=1)
    depth_mean = float(engine._static["depth"].mean())
    assert np.isfinite(depth_mean), "non-finite depth"

    result = None
    if rank == 0:
        picks, whole_index, all_pairs, mfma_variant = [], None, None, None
        if args.no_roofline_leg:
            kernel_s, alg_bytes, per_geometry, variants = float("nan"), (1 + M) * 32 * 128 * 160 * 4 + 64 * 128 * 160 * 4, [], []
        else:
            # the geometries of the TIMED steps (what a rocprofv3 kernel trace of this command's timed region averages over) ...
            timed = [seq[M + args.warmup + i] for i in range(args.steps)]
            kernel_s, alg_bytes, per_geometry, variants = measure_cost_volume_kernel(engine, M, args.kernel_reps, timed)
            # ... and, secondary, 25 lines spread over the WHOLE keyframe index (the timed steps of a short run are its first lines)
            picks, pose_sets = index_pose_sets(M, ROOFLINE_GEOMETRIES)
            w_s, _, w_per, w_var = measure_cost_volume_kernel(engine, M, args.kernel_reps, pose_sets)
            whole_index = {"kernel_us": w_s * 1e6, "frac": alg_bytes / w_s / 1e9 / HBM_PEAK_GBPS, "index_lines": picks,
                           "kernel_us_per_geometry": [round(t * 1e6, 2) for t in w_per], "sweep_variant_per_geometry": w_var,
                           "worst_us": max(w_per) * 1e6}
            # ... and EVERY keyframe pair of the index (285 lines; fewer repetitions per pair): the figure that does not flatter
            all_lines, all_sets = index_pose_sets(M, 10 ** 6)
            reps_all = args.kernel_reps      # (round 6: as many back-to-back ops per graph as on the timed lines -- with 3 per graph the gap between two graph
                                             # replays, ~5 - 10 us on the device, was 1.5 - 2 us of every op's figure)
            a_s, _, a_per, a_var = measure_cost_volume_kernel(engine, M, reps_all, all_sets, rounds=2)
            all_pairs = {"pairs": len(a_per), "kernel_us": a_s * 1e6, "frac": alg_bytes / a_s / 1e9 / HBM_PEAK_GBPS, "worst_us": max(a_per) * 1e6,
                         "worst_index_line": all_lines[int(np.argmax(a_per))], "p90_us": float(np.percentile(a_per, 90)) * 1e6,
                         "sweep_variants": {str(v): a_var.count(v) for v in sorted(set(a_var))}}
            # Comparison legs, same harness, same geometries: (i) the LDS-tiled sweep as dvmvs_sweep_plan plans it (what the engine ran before round 6
            # on the pairs it did not give to variant 6), (ii) variant 7 = the MFMA sweep as one work item per workgroup (round 5's launch shape; the
            # same arithmetic, bit-identical volumes): what the persistent form buys
            t_s, _, t_per, t_var = measure_cost_volume_kernel(engine, M, args.kernel_reps, timed, tiled_plan=True)
            ta_s, _, ta_per, _ = measure_cost_volume_kernel(engine, M, reps_all, all_sets, rounds=2, tiled_plan=True)
            g_s, _, g_per, _ = measure_cost_volume_kernel(engine, M, args.kernel_reps, timed, force_variant=7)
            ga_s, _, ga_per, _ = measure_cost_volume_kernel(engine, M, reps_all, all_sets, force_variant=7, rounds=2)
            mfma_variant = {"tiled_plan": {"kernel": "sweep_tiled_kernel [+ sweep_spill_kernel] as dvmvs_sweep_plan plans it", "kernel_us_timed_steps": t_s * 1e6,
                                           "kernel_us_all_pairs": ta_s * 1e6, "worst_us_all_pairs": max(ta_per) * 1e6, "p90_us_all_pairs": float(np.percentile(ta_per, 90)) * 1e6,
                                           "pairs_where_faster_than_engine_choice": int(sum(1 for x, y in zip(ta_per, a_per) if x < y)),
                                           "variants_timed_steps": {str(v): t_var.count(v) for v in sorted(set(t_var))}},
                            "one_item_per_workgroup": {"kernel": "sweep_mfma_kernel (variant 7)", "kernel_us_timed_steps": g_s * 1e6, "kernel_us_all_pairs": ga_s * 1e6,
                                                       "worst_us_all_pairs": max(ga_per) * 1e6, "p90_us_all_pairs": float(np.percentile(ga_per, 90)) * 1e6,
                                                       "pairs_where_faster_than_engine_choice": int(sum(1 for x, y in zip(ga_per, a_per) if x < y))}}
        achieved = alg_bytes / kernel_s / 1e9
        # HBM bytes per op from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: only when the committed measurement was taken on
        # exactly the kernel sources that are being benchmarked, otherwise null (profiles/README.md says how to re-collect)
        traffic, traffic_source = None, None
        try:
            import glob
            import hashlib
            digest = hashlib.sha256(b"".join(open(os.path.join(ROOT, "deep-video-mvs_amd", "csrc", f), "rb").read()
                                             for f in ("sweep_tiled.hip", "sweep_mfma.hip", "cost_volume.hip", "plane_sweep.h", "sweep_sample.h"))).hexdigest()
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cost_volume_pmc.json")), reverse=True):
                pmc = json.load(open(path))
                if pmc.get("shape") == [1, M, 32, 128, 160, 64] and pmc.get("kernel_sources_sha256") == digest:
                    traffic, traffic_source = pmc["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
                    break
        except (OSError, ValueError, KeyError):
            pass
        # useful arithmetic of the op: per (pixel, plane, frame) 4 taps x C channels of FMA + 4 weight FMAs, 2 flop each
        useful_flop = 128 * 160 * 64 * M * (4 * 32 + 4) * 2
        valu_tflops = useful_flop / kernel_s / 1e12
        lds_bytes = 128 * 160 * 64 * M * 4 * 32 * 4
        # (the LDS accounting below is the TILED formulation's -- the MFMA sweep reads 4 dot products per sample, not 4 x 32 channels: the steps that ran it are left out)
        tiled_times = [t for t, v in zip(per_geometry, variants) if v not in (6, 7)]
        tiled_s = sum(tiled_times) / len(tiled_times) if tiled_times else float("nan")
        other_kernels = None if args.no_roofline_leg else measure_small_kernels(engine)
        launches_per_frame = {}
        try:     # kernel nodes of the captured frame graph (what one replay launches), where the runtime can dump a graph
            counts = {f"n_meas={k[0]},has_previous={k[1]},sweep_variant={k[2]}": count_graph_kernels(g) for k, g in engine._graphs.items()}
            launches_per_frame.update({k: v for k, v in counts.items() if v is not None})
        except Exception:
            pass
        try:     # ... and the count a rocprofv3 kernel trace of THIS command's timed region recorded at the benchmarked look-ahead level
            import glob      # (profiles/, per round: the newest round that has one)
            level = 0 if args.no_feature_cache else args.lookahead
            region = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_bench_timed_region_lookahead{level}.csv")))[-1]
            last = open(region).read().strip().splitlines()[-1].split(",")
            launches_per_frame.update(profiled=float(last[5]) / float(last[7]), profiled_source=os.path.relpath(region, ROOT), profiled_lookahead=level)
        except Exception:
            pass
        launches_per_frame = launches_per_frame or None
        rel = golden_rel_l1(modules, device, args) if not args.no_rel_l1 else None
        result = {
            "metric": "depth frames/sec/GPU @ 320x256x64 planes (fusionnet); rel-L1 vs ref",
            "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # host time per step inside engine.step (asynchronous to the GPU: it matters only where it exceeds ms_per_step)
            "host_ms_per_step": host_ms,
            # ... of which the step WAITED for the device at the one-slot staging ring (the host is allowed one frame ahead), and the rest: the work
            # (parameter block, copies, hipGraphLaunch of the frame graph).  A host faster than the device shows as wait, not as a smaller total.
            "host_wait_for_device_ms_per_step": host_wait_ms, "host_work_ms_per_step": host_ms - host_wait_ms,
            "device_ms_between_step_ends": [round(step_events[i - 1].elapsed_time(step_events[i]), 3) for i in range(1, len(step_events))],
            # (strings of the line stay below 120 characters: the driver's record truncates longer ones)
            "config": {"workload": f"fusionnet inference, 1 synthetic sequence/GPU, sample-scene keyframe poses, 320x256x64 planes, M={M}, batch 1 (configs[2])",
                       "sequences_per_gpu": 1, "hip_graphs": not args.no_graphs, "bn_folded": not args.no_fold_bn,
                       "feature_cache": not args.no_feature_cache, "full_resolution_only": True,
                       "miopen_solver_search": bool(torch.backends.cudnn.benchmark),
                       "lookahead": {0: "none", 1: "next keyframe's feature extraction on a second stream",
                                     2: "next keyframe's features + sweep + encoder on a second stream (bit-identical results)"}[0 if args.no_feature_cache else args.lookahead],
                       "conv_epilogues_inside_miopen": (lambda rep: f"{sum(1 for r in rep if r[2])} of {len(rep)} dense problems "
                                                        "(bit-identical AND faster at warm-up)")(engine.conv_plan_report()),
                       "weights": "seeded (tests/synthetic.py) + published FPN checkpoint",
                       "cyclic_gc": "left on" if args.keep_gc else "off for warm-up + timed steps as in timeit (collect + freeze before); value_gc_on = collector left on",
                       "parallelism": f"sequence-sharded x{world}, no data-path collective"},
            "roofline": {"kernel": "sweep_mfma_persistent_kernel (variant 6: every pair since round 6, dvmvs_sweep_plan6); 1 launch = all planes x M frames",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "sample": "frac/achieved/kernel_us: mean over the timed steps' keyframe pairs; all_pairs: all 285 index lines; whole_index: 25 spread",
                         "all_pairs": all_pairs,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel_us": kernel_s * 1e6, "algorithmic_bytes": alg_bytes,
                         # the channels-last copy of a keyframe's features that variant 6 reads: round 5 a launch per frame outside this accounting
                         # (5 - 16 us in the frame), since round 6 written by the FPN smoothing layer's own epilogue (dvmvs_direct_conv_dual_fwd)
                         "kernel_us_incl_layout": kernel_s * 1e6 + 0.0, "layout_launches_per_frame": 0 if getattr(engine, "sweep_mfma", False) and engine.direct_convs else 1,
                         "geometries": f"{len(per_geometry)} timed steps = index lines {(M + args.warmup + 37 * rank) % 286}.. (nmeas+2 index), each as the engine launches it",
                         "kernel_us_per_geometry": [round(t * 1e6, 2) for t in per_geometry],
                         "sweep_variant_per_geometry": variants,
                         "sweep_variants": {"2 (default: 3 x 48 KB boxes, 256 threads; two passes)": variants.count(2),
                                            "3 (wide-baseline: 2 x 72 KB boxes, 512 threads; two passes)": variants.count(3),
                                            "4 (default, single pass: the host's plan queues nothing)": variants.count(4),
                                            "5 (wide-baseline, single pass)": variants.count(5),
                                            "6 (correlate-then-interpolate on the fp32 matrix cores, persistent form; channels-last maps)": variants.count(6)},
                         "engine_frames_per_sweep_variant": {str(k): v for k, v in sorted(engine.sweep_variant_counts.items())},
                         "whole_index": whole_index},
            # HBM is not what binds this op (13 MB of algorithmic traffic against 0.69 GFLOP of tap arithmetic and 1.3 GB of LDS
            # reads): the same duration against the fp32 vector peak, counting only the useful tap FMAs
            "roofline_valu": {"bound": "valu_fp32", "achieved": valu_tflops, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": valu_tflops / FP32_VALU_PEAK_TFLOPS, "useful_flop": useful_flop},
            # the pipe this formulation is actually bound by: 4 taps x 32 channels x 4 B from LDS per (pixel, plane, frame).  "peak" is
            # the ds_read_b128 rate of the chip; "pattern_floor_us" the measured floor of the kernel's own inner pattern (bank conflicts
            # of the tap addresses and the packed FMAs between the reads included)
            "roofline_lds": None if not tiled_times else {
                "bound": "lds", "kernel": f"sweep_tiled_kernel ({len(tiled_times)} of the {len(per_geometry)} timed steps ran it)", "achieved": lds_bytes / tiled_s / 1e12,
                "peak": LDS_PEAK_TBPS, "unit": "TB/s", "frac": lds_bytes / tiled_s / 1e12 / LDS_PEAK_TBPS, "lds_bytes": lds_bytes, "kernel_us": tiled_s * 1e6,
                "pattern_floor_us": SWEEP_LDS_PATTERN_FLOOR_US, "frac_of_pattern_floor": SWEEP_LDS_PATTERN_FLOOR_US / (tiled_s * 1e6)},
            # what the launch's duration is made of (round 5): instruction ISSUE.  Summed over a SIMD's waves, the cycles with an instruction
            # issuing / executing (SQ_ACTIVE_INST_ANY of the committed PMC profile, index line 0) add up to the waves' lifetime in both sweep
            # kernels: neither HBM nor LDS bandwidth nor the matrix pipe but the instruction count bounds them
            "roofline_issue": issue_roofline(per_geometry, variants) if per_geometry else None,
            "roofline_comparison_kernels": mfma_variant,
            "roofline_other": other_kernels,
            # what the engine's warm-up costs (outside the timed steps): eager first frames, graph capture, first launches of the graphs
            # captured ahead; device memory reserved beyond live tensors (the captured graphs' private pools are part of it)
            "warmup_report": engine.graph_memory_report(),      # ("warmup" is the contract's integer above)
            "launches_per_frame": launches_per_frame,
            "rel_l1": None if rel is None else {
                "what": "depth rel-L1 mean(|d - d_ref| / d_ref) of this engine configuration (graph replay included) vs the REFERENCE forward "
                        "(fixtures captured from the reference itself); target 1e-4.  teacher_forced: the reference's own (h, c, previous depth, "
                        "previous pose) installed before every step -- 3 golden frames + the 14-keyframe long run, full resolution; free_running: "
                        "the engine's own state over the 3 golden frames, with the number of pixels of the discrete 8x10 depth estimate that "
                        "differ from the reference's (after a flipped pixel the runs see different inputs).  The fixture host's fp32 pose "
                        "matrices are replayed (tests/golden/host_pose_algebra.npz); the kernels alone are pinned by tests/test_hybrid_parity.py",
                "target": 1e-4,
                "teacher_forced": [round(r["rel_l1"], 9) for r in rel["teacher_forced"]],
                "teacher_forced_max": max(r["rel_l1"] for r in rel["teacher_forced"]),
                "teacher_forced_frames": len(rel["teacher_forced"]),
                "teacher_forced_flipped_estimate_pixels": [r["flipped_estimate_pixels"] for r in rel["teacher_forced"]],
                "teacher_forced_steps": [f"{r['run']} step {r['step']}" for r in rel["teacher_forced"]],
                "free_running": [r["engine_vs_reference"] for r in rel["free_running"]],
                "free_running_flipped_estimate_pixels": [r["flipped_estimate_pixels"] for r in rel["free_running"]],
                "engine_vs_float64": [r["engine_vs_float64"] for r in rel["free_running"]],
                "reference_vs_float64": [r["reference_vs_float64"] for r in rel["free_running"]],
                "host_algebra_misses": rel["host_algebra_misses"]},
        }
        if world == 1 and args.sequences_per_gpu > 1:
            try:
                S = args.sequences_per_gpu
                fps_b, ms_b, kernel_b, bytes_b = batched_throughput(modules, device, args, S, M)
                result["value_batched"] = fps_b
                result["batched"] = {"sequences_per_gpu": S, "frames_per_s": fps_b, "ms_per_lockstep_step": ms_b, "steps": args.batched_steps,
                                     "note": "secondary: S independent sequences in lockstep on one engine (batch S); not the headline"}
                result["roofline"]["frac_batched"] = bytes_b / kernel_b / 1e9 / HBM_PEAK_GBPS
                result["roofline"]["kernel_us_batched"] = kernel_b * 1e6
                result["roofline_valu"]["frac_batched"] = (S * useful_flop / kernel_b / 1e12) / FP32_VALU_PEAK_TFLOPS
            except Exception as e:   # the secondary figure must never cost the headline line
                result["value_batched"] = None
                result["batched"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_secondary:
            # secondary figures, each on its own engine, never at the cost of the headline line: (i) the collector left on, (ii) the causal
            # pipeline (no look-ahead: what an online caller without the next frame gets), (iii) BASELINE.json configs[1]: pairnet inference
            # (M = 1, no ConvLSTM; /root/reference/dvmvs/pairnet/run-testing.py:136-166)
            steps2, warm2 = min(args.steps, 60), max(args.warmup, 5)
            level = 0 if args.no_feature_cache else args.lookahead
            for key, fn in (("value_gc_on", lambda: secondary_leg(modules, device, args, M, level, steps2, warm2, gc_on=True)[0]),
                            ("value_causal", lambda: secondary_leg(modules, device, args, M, 0, steps2, warm2)[0])):
                try:
                    result[key] = fn()
                except Exception as e:
                    result[key] = None
                    result[key + "_error"] = f"{type(e).__name__}: {e}"
            # The headline depends on a property of the runtime that is observed, not documented (the hardware queue
            # hipGraphInstantiate hands a frame graph's second branch; the engine steers it with never-launched filler graphs).  The same
            # measurement with and without them, back to back on this box: a runtime that behaves differently shows up here.
            try:
                import dvmvs.engine as engine_module
                fps_on = secondary_leg(modules, device, args, M, level, steps2, warm2, queue_fillers=engine_module._GRAPH_QUEUE_FILLERS)[0]
                fps_off = secondary_leg(modules, device, args, M, level, steps2, warm2, queue_fillers=0)[0]
                result["graph_queue_fillers"] = {"configured": engine_module._GRAPH_QUEUE_FILLERS, "ms_per_step_on": 1e3 / fps_on, "ms_per_step_off": 1e3 / fps_off,
                                                 "steps": steps2, "note": "secondary engines, same box, back to back; on = as the headline"}
            except Exception as e:
                result["graph_queue_fillers"] = {"error": f"{type(e).__name__}: {e}"}
            try:      # the drop-in route north_star names, timed: no engine, no graphs, poses on the device
                surface = module_surface_leg(build_modules(), device, args, M, steps2, warm2)
                result["module_surface"] = surface
                result["value_module_surface"] = surface["value"]
                fast = module_surface_leg(build_modules(), device, args, M, steps2, warm2, accelerated=True)
                result["module_surface_accelerated"] = fast
                result["value_module_surface_accelerated"] = fast["value"]
                graphed = module_surface_leg(build_modules(), device, args, M, steps2, max(warm2, 6), accelerated=True, graphs=True)
                result["module_surface_accelerated_graphs"] = graphed
                result["value_module_surface_accelerated_graphs"] = graphed["value"]
            except Exception as e:
                result["module_surface"] = {"error": f"{type(e).__name__}: {e}"}
                result["value_module_surface"] = None
            try:
                import synthetic as syn
                from dvmvs.pairnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker
                pair_modules = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, CostVolumeDecoder))
                pair_modules.insert(3, None)
                fps_p, pair_engine = secondary_leg(pair_modules, device, args, 1, level, steps2, warm2)
                result["pairnet"] = {"value": fps_p, "unit": "frames/s", "ms_per_step": 1e3 / fps_p, "steps": steps2, "warmup": warm2,
                                     "config": {"workload": "pairnet inference, one synthetic-image sequence on the sample scene's keyframe poses, 320x256, "
                                                            "64 planes, M=1 measurement frame, batch 1 (BASELINE.json configs[1])",
                                                "lookahead": level, "hip_graphs": not args.no_graphs, "feature_cache": not args.no_feature_cache}}
            except Exception as e:
                result["pairnet"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            cpu_mods = build_modules()
            result["cpu_baseline"] = cpu_baseline(args, cpu_mods, M)
        else:
            result["cpu_baseline"] = None
        if args.stage_times:
            print(f"[bench] final depth mean {depth_mean:.4f}", file=sys.stderr)
        print(json.dumps(finite_or_null(result)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
