"""Deterministic inputs shared by the golden generator (build container) and the tests (build container + GPU box).

Everything here is computed from closed-form expressions in float64 and rounded once to float32, or read from the
small data fixtures under ``tests/golden/`` (``hololens_000_poses.txt`` / ``hololens_000_K.txt`` are the pose and
intrinsics files of the reference's sample scene -- data, not code).  No RNG state is relied upon except where a
function takes an explicit seed and uses a local ``torch.Generator``.
"""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MIN_DEPTH, MAX_DEPTH, N_PLANES = 0.25, 20.0, 64


def sample_poses():
    """[373,4,4] float64 camera-to-world poses of sample-data/hololens-dataset/000 (parsed as run-testing.py:76 does)."""
    return np.fromfile(os.path.join(GOLDEN_DIR, "hololens_000_poses.txt"), dtype=float, sep="\n ").reshape((-1, 4, 4))


def pose(i):
    return torch.from_numpy(sample_poses()[i]).float().unsqueeze(0)


def full_K(width=320, height=256, raw_width=540, raw_height=360):
    """Intrinsics of the sample scene rescaled to the network input size without cropping
    (what PreprocessImage(..., perform_crop=False) produces; dataset_loader.py:314-323)."""
    K = np.loadtxt(os.path.join(GOLDEN_DIR, "hololens_000_K.txt")).astype(np.float32)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    sx, sy = float(width) / float(raw_width), float(height) / float(raw_height)
    out = np.array([[fx * sx, 0, cx * sx], [0, fy * sy, cy * sy], [0, 0, 1]])
    return torch.from_numpy(out).float().unsqueeze(0)


def scaled_K(K, factor):
    """K with its first two rows divided by ``factor`` (half_K = /2, lstm_K = /32; run-testing.py:143-149)."""
    out = K.clone()
    out[:, 0:2, :] = out[:, 0:2, :] / factor
    return out


def analytic_features(s, channels=32, height=128, width=160):
    """feat(s)[0,c,y,x] = sin(0.1(c+1+s)x/8 + 0.07(c+2)y/8 + 0.3s), float64 -> float32."""
    c = np.arange(channels, dtype=np.float64).reshape(-1, 1, 1)
    y = np.arange(height, dtype=np.float64).reshape(1, -1, 1)
    x = np.arange(width, dtype=np.float64).reshape(1, 1, -1)
    f = np.sin(0.1 * (c + 1 + s) * x / 8 + 0.07 * (c + 2) * y / 8 + 0.3 * s)
    return torch.from_numpy(f.astype(np.float32)).unsqueeze(0)


def analytic_depth(height=256, width=320):
    """SURVEY KAT-REPROJ: prev[0,0,y,x] = 1.5 + 0.5 sin(x/40) + 0.3 cos(y/30)."""
    y = np.arange(height, dtype=np.float64).reshape(-1, 1)
    x = np.arange(width, dtype=np.float64).reshape(1, -1)
    return torch.from_numpy((1.5 + 0.5 * np.sin(x / 40) + 0.3 * np.cos(y / 30)).astype(np.float32)).reshape(1, 1, height, width)


def analytic_lstm_inputs(hidden=512, height=8, width=10):
    """SURVEY KAT-LSTM inputs: conv weight, x, h0, c0 (closed form)."""
    o = np.arange(4 * hidden, dtype=np.float64).reshape(-1, 1, 1, 1)
    i = np.arange(2 * hidden, dtype=np.float64).reshape(1, -1, 1, 1)
    ky = np.arange(3, dtype=np.float64).reshape(1, 1, -1, 1)
    kx = np.arange(3, dtype=np.float64).reshape(1, 1, 1, -1)
    weight = np.sin(0.37 * o + 0.11 * i + 1.3 * ky + 0.7 * kx) / 96
    c = np.arange(hidden, dtype=np.float64).reshape(-1, 1, 1)
    y = np.arange(height, dtype=np.float64).reshape(1, -1, 1)
    x = np.arange(width, dtype=np.float64).reshape(1, 1, -1)
    xin = np.sin(0.05 * c + 0.9 * y + 0.4 * x)
    h0 = np.cos(0.03 * c + 0.5 * y - 0.6 * x)
    c0 = np.sin(0.02 * c - 0.2 * y + 0.8 * x)
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))
    return f32(weight), f32(xin).unsqueeze(0), f32(h0).unsqueeze(0), f32(c0).unsqueeze(0)


def smooth_noise(shape, seed, passes=3):
    """Seeded N(0,1) field low-pass filtered with a 5x5 box ``passes`` times and re-normalised to unit variance.

    Used for synthetic "already normalised" images and feature maps that are not degenerate for correlation.
    """
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    lead = x.shape[:-2]
    x = x.reshape(-1, 1, x.shape[-2], x.shape[-1])
    k = torch.ones(1, 1, 5, 5) / 25.0
    for _ in range(passes):
        x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), k)
    x = x.reshape(*lead, x.shape[-2], x.shape[-1])
    return (x - x.mean()) / x.std()


def sample_indices(numel, count=4096, seed=12345):
    """Fixed pseudo-random flat indices used to pin large tensors by a subset of their entries."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(count, numel),), generator=g)


def tensor_pins(t, count=4096):
    """Summary of a tensor that a test on another machine can compare against: sums + sampled entries."""
    flat = t.detach().reshape(-1).double()
    idx = sample_indices(flat.numel(), count)
    return {"shape": np.array(t.shape), "sum": flat.sum().item(), "abs_sum": flat.abs().sum().item(),
            "samples": t.detach().reshape(-1)[idx].float().numpy().copy()}


def synthetic_trajectory(n_frames, seed=1000):
    """Smooth camera-to-world trajectory: ~0.12 m baseline and ~3 degrees of rotation between consecutive keyframes
    (the sample scene's median keyframe baseline is 0.144 m).  Returns [n,4,4] float64."""
    rng = np.random.RandomState(seed)
    poses = np.zeros((n_frames, 4, 4))
    phase = rng.uniform(0, 2 * np.pi, size=3)
    for i in range(n_frames):
        t = np.array([0.12 * i, 0.03 * np.sin(0.35 * i + phase[0]), 0.04 * np.sin(0.2 * i + phase[1])])
        yaw = np.deg2rad(3.0) * i * 0.5 + 0.05 * np.sin(0.3 * i + phase[2])
        pitch = 0.03 * np.sin(0.25 * i + phase[0])
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        poses[i, :3, :3] = Ry @ Rx
        poses[i, :3, 3] = t
        poses[i, 3, 3] = 1.0
    return poses


def deterministic_init(module, seed=0):
    """Fills every parameter / buffer of ``module`` from a generator keyed on (seed, tensor name), so that two
    modules with the same state-dict keys and shapes (the reference's and this repo's) get identical weights
    without shipping a checkpoint.  Conv weights ~ N(0, 2/fan_in); BN weight ~ 1 +- 0.1, bias ~ +-0.1,
    running_mean ~ +-0.1, running_var ~ 1 +- 0.2; conv bias ~ +-0.05."""
    import zlib
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue  # num_batches_tracked
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
            if t.dim() == 4:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            elif name.endswith("running_var"):
                t.copy_(1.0 + 0.2 * (torch.rand(t.shape, generator=g) - 0.5) * 2)
            elif name.endswith("running_mean"):
                t.copy_(0.1 * (torch.rand(t.shape, generator=g) - 0.5) * 2)
            elif name.endswith("weight"):
                t.copy_(1.0 + 0.1 * (torch.rand(t.shape, generator=g) - 0.5) * 2)
            elif name.endswith("bias"):
                scale = 0.1 if t.numel() > 1 else 0.05
                t.copy_(scale * (torch.rand(t.shape, generator=g) - 0.5) * 2)
            else:
                raise ValueError(f"deterministic_init: do not know how to fill {name} {tuple(t.shape)}")
    return module


def load_fpn_checkpoint():
    """State dict of the reference's published FPN weights (the only checkpoint that survives in the reference
    snapshot: dvmvs/fusionnet/weights/1_feature_pyramid), kept as a data fixture."""
    return torch.load(os.path.join(GOLDEN_DIR, "fpn_checkpoint.pth"), map_location="cpu", weights_only=True)


E2E_MODULE_NAMES = ("feature_extractor", "feature_shrinker", "encoder", "lstm_fusion", "decoder")


def calibrate_batchnorm(run_forward, modules):
    """Sets every BatchNorm's running statistics to the batch statistics of ONE forward pass (``run_forward()``),
    so that seeded random weights give O(1) activations like a trained network.  Used only by make_goldens.py; the
    resulting statistics are stored in tests/golden/e2e_bn_stats.npz and re-loaded by ``apply_bn_stats``."""
    bns = [m for mod in modules for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [(m.momentum, m.training) for m in bns]
    for m in bns:
        m.momentum = 1.0
        m.train()
    with torch.no_grad():
        run_forward()
    for m, (mom, training) in zip(bns, saved):
        m.momentum = mom
        m.train(training)


def collect_bn_stats(named_modules):
    out = {}
    for mod_name, mod in named_modules:
        for k, v in mod.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out[f"{mod_name}/{k}"] = v.detach().cpu().numpy().copy()
    return out


def apply_bn_stats(named_modules, path=None):
    stats = np.load(path or os.path.join(GOLDEN_DIR, "e2e_bn_stats.npz"))
    with torch.no_grad():
        for mod_name, mod in named_modules:
            sd = mod.state_dict()
            for k in sd:
                if k.endswith("running_mean") or k.endswith("running_var"):
                    sd[k].copy_(torch.from_numpy(stats[f"{mod_name}/{k}"]))


def build_e2e_modules(constructors, with_bn_stats=True):
    """constructors: the five (fusionnet) or four (pairnet: no lstm) module classes in checkpoint order.
    Returns eval-mode modules with the deterministic weights the e2e goldens were generated with."""
    names = E2E_MODULE_NAMES if len(constructors) == 5 else tuple(n for n in E2E_MODULE_NAMES if n != "lstm_fusion")
    seeds = {n: i for i, n in enumerate(E2E_MODULE_NAMES)}
    mods = []
    for name, ctor in zip(names, constructors):
        mod = deterministic_init(ctor(), seed=seeds[name])
        if name == "feature_shrinker":
            mod.load_state_dict(load_fpn_checkpoint())
        mods.append(mod.eval())
    if with_bn_stats:
        apply_bn_stats(list(zip(names, mods)))
    return mods


# first lines of the nmeas+2 keyframe index of the sample scene ("00012.png 00009.png 00003.png", ...) as POSE indices:
# the scene's image files start at 00003.png, and the scripts map a file name to its position in the sorted list
E2E_FRAMES = ((9, (6, 0)), (10, (9, 6)), (11, (9, 10)))


# Long reference-run sequence (tests/golden/fusionnet_long.npz): lines of the sample scene's nmeas+2 keyframe index, None =
# "TRACKING LOST" (fusionnet/run-testing.py:97-101).  Openings, a tracking loss, an easy stretch, the wide-baseline lines
# 200-204 whose footprints spill, and 249-251.
LONG_SCHEDULE = (0, 1, 2, None, 117, 118, 119, 200, 201, 202, 203, 204, 249, 250, 251)


def keyframe_index_lines(n_meas=2):
    """[(reference pose index, (measurement pose indices))] of the committed keyframe index, "TRACKING LOST" lines dropped."""
    names = {n: i for i, n in enumerate(sample_image_names())}
    out = []
    with open(os.path.join(GOLDEN_DIR, "indices", f"keyframe+hololens-dataset+000+nmeas+{n_meas}")) as f:
        for line in f:
            parts = line.split()
            if len(parts) == n_meas + 1 and all(p in names for p in parts):
                out.append((names[parts[0]], tuple(names[p] for p in parts[1:])))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# host pose algebra of the fixture host
# ----------------------------------------------------------------------------------------------------------------------
# The reference evaluates its few small matrices with fp32 LAPACK (torch.inverse -> MKL getrf/getrs).  MKL picks its kernels by
# CPU, so the last bits of an fp32 4x4 inverse differ between hosts (observed: the Intel build container the fixtures were
# captured on vs the AMD host of the GPU box; the resulting relative translations differ by ~1e-7 m, i.e. as much as either is
# off the real-number result).  A fixture captured from the reference on one host therefore pins the DEPTH only together with
# the matrices that host computed.  tests/golden/host_pose_algebra.npz holds them (make_goldens.pose_algebra_goldens: the
# reference's expressions, evaluated in the same process as the reference run), keyed by the exact input bits, for every pose
# pair the fixture-comparing tests use; the `fixture_host_algebra` pytest fixture replays them in place of the local LAPACK.
def golden_algebra_pairs():
    """(sweep pairs [(ref, meas, K tag)], relative-pose pairs [(a, c)]) whose matrices the fixture host recorded."""
    sweeps, rels = set(), set()
    for n_meas in (1, 2, 3):
        lines = keyframe_index_lines(n_meas)
        for r, ms in lines:
            sweeps.update((r, m, "half") for m in ms)
        for (r0, _), (r1, _) in zip(lines, lines[1:]):
            rels.update([(r1, r0), (r0, r1)])
    refs = [keyframe_index_lines(2)[i][0] for i in LONG_SCHEDULE if i is not None]      # consecutive frames of the long run
    for r0, r1 in zip(refs, refs[1:]):
        rels.update([(r1, r0), (r0, r1)])
    for r in range(0, 40):
        for m in range(max(0, r - 12), r + 3):
            sweeps.add((r, m, "half"))
            rels.update([(r, m), (m, r)])
    for r, ms in ((141, (135, 130, 120)), (202, (196, 188, 180)), (170, (168, 167, 160)), (166, (165, 164)), (278, (277, 276))):
        sweeps.update((r, m, "half") for m in ms)
        rels.update((r, m) for m in ms)
        rels.update((m, r) for m in ms)
    for r, ms in ((12, (9, 3)), (23, (22, 21, 20)), (141, (135,))):
        sweeps.update((r, m, "small") for m in ms)
    return sorted(sweeps), sorted(rels)


def golden_algebra_K(tag):
    half = scaled_K(full_K(), 2.0)
    return half if tag == "half" else scaled_K(half, 4.0)


class FixtureHostAlgebra:
    """Replays the fixture host's fp32 pose algebra (host_pose_algebra.npz) for inputs whose bits it has seen; anything else
    raises KeyError (add the pair to golden_algebra_pairs and regenerate)."""

    def __init__(self, path=None):
        import numpy as np
        z = np.load(path or os.path.join(GOLDEN_DIR, "host_pose_algebra.npz"))
        self.sweep = {z["sweep_in"][i].tobytes(): z["sweep_out"][i] for i in range(len(z["sweep_in"]))}
        self.rel = {z["rel_in"][i].tobytes(): z["rel_out"][i] for i in range(len(z["rel_in"]))}

    @staticmethod
    def _row(*tensors):
        import numpy as np
        return np.concatenate([t.detach().cpu().contiguous().numpy().astype(np.float32).reshape(-1) for t in tensors]).tobytes()

    def relative_pose_host(self, a, c):
        out = [self.rel[self._row(a[b], c[b])] for b in range(a.shape[0])]
        return torch.from_numpy(__import__("numpy").stack(out)).reshape(-1, 4, 4).clone()

    def sweep_matrices_host(self, pose1, pose2s, K):
        import numpy as np
        B, M = pose1.shape[0], len(pose2s)
        rows = np.stack([np.stack([self.sweep[self._row(pose1[b], pose2s[m][b], K[b])] for m in range(M)]) for b in range(B)])   # [B,M,12]
        rows = torch.from_numpy(rows)
        return rows[:, :, :9].contiguous().clone(), rows[:, :, 9:].contiguous().clone()

    def plane_sweep_setup(self, pose1, pose2, K):    # oracle/dvmvs_oracle.py's signature: (KRKinv [B,3,3], Kt [B,3,1])
        Hm, kt = self.sweep_matrices_host(pose1, [pose2], K)
        return Hm[:, 0].reshape(-1, 3, 3), kt[:, 0].reshape(-1, 3, 1)


def sample_image_names():
    """Sorted image file names of the sample scene (row i of poses.txt belongs to the i-th name)."""
    with open(os.path.join(GOLDEN_DIR, "hololens_000_image_names.txt")) as f:
        return [line.strip() for line in f if line.strip()]


def sample_image_name(pose_index):
    return sample_image_names()[pose_index]


def e2e_image(index):
    return smooth_noise((1, 3, 256, 320), seed=2000 + index)


def error_metric_inputs():
    """Deterministic (ground truth, prediction) pair for the evaluation-metric golden: invalid pixels (< 0.5 m) included."""
    gt = analytic_depth().numpy()[0, 0].astype(np.float64)
    gt[:20, :30] = 0.0
    pred = gt * (1.0 + 0.1 * np.sin(np.arange(320) / 17.0))[None, :] + 0.05
    pred[:20, :30] = 1.0
    return gt, pred


def loss_inputs():
    """Deterministic (ground truth [B,H,W], [predictions at 1/4, 1/2, full resolution]) for the training-loss golden:
    invalid (zero) ground-truth pixels, errors on both sides of the smooth-L1 knee."""
    B, H, W = 2, 32, 40
    y = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    x = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1)
    gt = 1.5 + 0.8 * torch.sin(x / 7.0 + b) + 0.5 * torch.cos(y / 5.0)
    gt[:, :6, :9] = 0.0
    gt[1, 20:24, 30:] = 0.0
    gt = gt.float()
    preds = []
    for s in (4, 2, 1):
        h, w = H // s, W // s
        yy = torch.arange(h, dtype=torch.float64).view(1, h, 1) * s
        xx = torch.arange(w, dtype=torch.float64).view(1, 1, w) * s
        p = 1.5 + 0.8 * torch.sin(xx / 7.0 + b + 0.05 * s) + 0.5 * torch.cos(yy / 5.0) + 0.3 * torch.sin(xx * yy / 90.0) * s
        preds.append(p.clamp(min=0.3).float())
    return gt, preds


def tsdf_inputs():
    """Deterministic two-frame RGB-D input for the TSDF goldens: a tilted wall at 0.9-1.3 m with an invalid (zero depth)
    corner, 64 x 48 images, poses = two keyframes of the sample scene moved to the origin, volume 20 x 16 x 24 voxels of 4 cm."""
    H, W = 48, 64
    K = np.array([[61.3, 0.0, 31.37], [0.0, 59.1, 23.61], [0.0, 0.0, 1.0]])   # (irrational-ish: no systematic projection ties)
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    frames = []
    base = np.linalg.inv(sample_poses()[9])
    for n, idx in enumerate((9, 12)):
        depth = (0.9 + 0.006 * x + 0.003 * y + 0.05 * n).astype(np.float32)
        depth[:6, :8] = 0.0
        color = np.stack([(40 + 3 * x + n * 17) % 256, (200 - 2 * y + n * 5) % 256, (x + 2 * y + 90) % 256], axis=-1).astype(np.uint8)
        frames.append((color, depth, K.copy(), (base @ sample_poses()[idx])))
    bounds = np.array([[-0.42, 0.38], [-0.34, 0.30], [0.62, 1.58]])
    return frames, bounds, 0.04
