"""CPU: the search-window argument of the measurement-gradient gather kernel (csrc/cost_volume_bwd.hip,
cost_volume_bwd_meas_gather_kernel), restated in numpy and checked against float64 autograd through the oracle.

The kernel's correctness rests on one geometric claim: for a measurement pixel q and a sweep plane, every reference pixel whose
bilinear 2x2 footprint contains q lies in the bounding box of the inverse-homography images of the corners of (q - 1.01, q + 1.01)^2
-- or the kernel scans the whole image (footprint on the plane's vanishing line, singular plane matrix).  Here the same window rule
(same margins, same trust test) selects the candidates, the candidates' forward positions give taps and weights as in the forward
pass, and the accumulated gradient must equal autograd's for ordinary pairs, a behind-camera pair, wide baselines and rotations that
put the vanishing line inside the image.  Also counted: how many candidates the window visits (the kernel's work per pixel and plane).
"""
import math

import numpy as np
import pytest
import torch

import dvmvs_oracle as orc
import synthetic as syn
from dvmvs import pose_algebra


def gather_gradient(ref, grad_cost, Hm, kt, D, min_depth, max_depth):
    """d cost / d measurement features [C,H,W] for one measurement frame (M = 1), by the kernel's algorithm.
    ``ref`` [C,H,W] float64, ``grad_cost`` [D,H,W] float64, ``Hm`` [9], ``kt`` [3] float32 (dvmvs.pose_algebra)."""
    C, H, W = ref.shape
    out = np.zeros((C, H, W))
    inv_depth = 1.0 / max_depth + np.arange(D) * (1.0 / min_depth - 1.0 / max_depth) / (D - 1)
    scale = 1.0 / C
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    Hf = Hm.astype(np.float32)
    whole_image = candidates = 0
    for d in range(D):
        ktd = (kt.astype(np.float32) / np.float32(1.0 / inv_depth[d])).astype(np.float32)
        A = Hm.reshape(3, 3).astype(np.float64).copy()
        A[:, 2] += ktd.astype(np.float64)
        with np.errstate(all="ignore"):
            inv = np.linalg.inv(A) if abs(np.linalg.det(A)) > 0 else np.full((3, 3), np.nan)
        inv[:, 0] *= W / (W - 1)
        inv[:, 1] *= H / (H - 1)
        iv = inv.astype(np.float32)
        with np.errstate(all="ignore"):      # forward positions, the forward kernel's expressions (csrc/plane_sweep.h, sweep_position)
            X = Hf[2] + Hf[1] * ys + Hf[0] * xs + ktd[0]
            Y = Hf[5] + Hf[4] * ys + Hf[3] * xs + ktd[1]
            Z = Hf[8] + Hf[7] * ys + Hf[6] * xs + ktd[2] + np.float32(1e-8)
            ix = ((X / Z - W * 0.5) / (W * 0.5) + 1) * 0.5 * (W - 1)
            iy = ((Y / Z - H * 0.5) / (H * 0.5) + 1) * 0.5 * (H - 1)
        for qy in range(H):
            for qx in range(W):
                px, py, pw = [], [], []
                for k in range(4):
                    tx, ty = qx + (1.01 if k & 1 else -1.01), qy + (1.01 if k & 2 else -1.01)
                    w = iv[2, 0] * tx + iv[2, 1] * ty + iv[2, 2]
                    with np.errstate(all="ignore"):
                        px.append((iv[0, 0] * tx + iv[0, 1] * ty + iv[0, 2]) / w)
                        py.append((iv[1, 0] * tx + iv[1, 1] * ty + iv[1, 2]) / w)
                    pw.append(w)
                lo_w, hi_w = min(pw), max(pw)
                one_sign = (lo_w > 0 and lo_w > 1e-3 * hi_w) or (hi_w < 0 and hi_w < 1e-3 * lo_w)
                bounded = min(px) > -1e7 and max(px) < 1e7 and min(py) > -1e7 and max(py) < 1e7
                x0, x1, y0, y1 = 0, W - 1, 0, H - 1
                if one_sign and bounded:
                    x0, x1 = max(0, math.ceil(min(px) - 0.05)), min(W - 1, math.floor(max(px) + 0.05))
                    y0, y1 = max(0, math.ceil(min(py) - 0.05)), min(H - 1, math.floor(max(py) + 0.05))
                else:
                    whole_image += 1
                if x1 < x0 or y1 < y0:
                    continue
                sx, sy = ix[y0:y1 + 1, x0:x1 + 1], iy[y0:y1 + 1, x0:x1 + 1]
                candidates += sx.size
                ok = (sx > -2) & (sx < W + 1) & (sy > -2) & (sy < H + 1)
                fx, fy = np.floor(np.where(ok, sx, 0)), np.floor(np.where(ok, sy, 0))
                dx, dy = qx - fx, qy - fy
                ok &= (dx >= 0) & (dx <= 1) & (dy >= 0) & (dy <= 1)
                if not ok.any():
                    continue
                wx = np.where(dx > 0, sx - fx, fx + 1 - sx)
                wy = np.where(dy > 0, sy - fy, fy + 1 - sy)
                coef = np.where(ok, wx * wy * grad_cost[d, y0:y1 + 1, x0:x1 + 1] * scale, 0.0)
                out[:, qy, qx] += (ref[:, y0:y1 + 1, x0:x1 + 1] * coef[None]).sum(axis=(1, 2))
    return out, whole_image, candidates / (H * W * D)


def rot_y(deg):
    t = np.deg2rad(deg)
    R = torch.eye(4)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = float(np.cos(t)), float(np.sin(t)), float(-np.sin(t)), float(np.cos(t))
    return R[None]


CASES = [
    # name, reference pose, measurement pose, focal scale, (C, H, W, D), expects whole-image scans
    ("easy sideways pair", lambda: (syn.pose(10), syn.pose(9)), 1.0, (4, 20, 28, 10), False),
    ("behind-camera corner", lambda: (syn.pose(141), syn.pose(135)), 1.0, (4, 20, 28, 10), False),
    ("wide baseline", lambda: (syn.pose(202), syn.pose(188)), 1.0, (3, 24, 32, 8), False),
    ("forward motion", lambda: (syn.pose(170), syn.pose(160)), 1.0, (3, 24, 32, 8), False),
    ("identical poses", lambda: (syn.pose(33), syn.pose(33)), 1.0, (3, 16, 20, 6), False),
    ("vanishing line in the image, 50 degrees", lambda: (syn.pose(10), syn.pose(10) @ rot_y(50.0)), 0.35, (3, 24, 32, 8), True),
    ("vanishing line in the image, 75 degrees", lambda: (syn.pose(10), syn.pose(10) @ rot_y(75.0)), 0.2, (3, 20, 28, 6), True),
]


@pytest.mark.parametrize("name,poses,focal_scale,shape,scans", CASES, ids=[c[0] for c in CASES])
def test_window_rule_finds_every_contributor(name, poses, focal_scale, shape, scans):
    C, H, W, D = shape
    g = torch.Generator().manual_seed(len(name))
    a, b = torch.randn(1, C, H, W, generator=g), torch.randn(1, C, H, W, generator=g)
    go = torch.randn(1, D, H, W, generator=g)
    p1, p2 = poses()
    K = syn.scaled_K(syn.full_K(), 320.0 / W).clone()
    K[:, 0, 0] *= focal_scale
    K[:, 1, 1] *= focal_scale
    ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    orc.cost_volume_fusion(ac, [bc], p1, [p2], K, 0.25, 20.0, D, True).backward(go)
    expected = bc.grad[0].numpy().astype(np.float64)
    Hm, kt = pose_algebra.sweep_matrices_host(p1, [p2], K)
    got, whole_image, per_pixel_plane = gather_gradient(a[0].numpy().astype(np.float64), go[0].numpy().astype(np.float64),
                                                        Hm[0, 0].numpy(), kt[0, 0].numpy(), D, 0.25, 20.0)
    # a missed contributor is a whole tap (O(|g| |f|) ~ 1), float32-vs-float64 positions only move weights by ~1e-5
    assert np.abs(got - expected).max() <= 2e-4 * max(1.0, np.abs(expected).max()), name
    assert (whole_image > 0) == scans, (name, whole_image)
    if not scans:
        assert per_pixel_plane <= 12.0        # a handful of candidates per (pixel, plane): the kernel's work is bounded


def random_pose(rng, max_angle_deg, max_shift):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    angle = np.deg2rad(rng.uniform(-max_angle_deg, max_angle_deg))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * Kx @ Kx
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-max_shift, max_shift, size=3)
    return torch.from_numpy(T)[None]


def test_window_rule_on_random_geometries():
    """30 seeded random relative poses (rotations up to 45 degrees about any axis, shifts up to 0.6 m, focal lengths from normal to
    very wide): whatever the window rule decides -- a handful of candidates or a whole-image scan -- no contributor is missed."""
    rng = np.random.default_rng(2024)
    C, H, W, D = 2, 12, 16, 5
    scanned = 0
    for trial in range(30):
        g = torch.Generator().manual_seed(trial)
        a, b = torch.randn(1, C, H, W, generator=g), torch.randn(1, C, H, W, generator=g)
        go = torch.randn(1, D, H, W, generator=g)
        p1 = syn.pose(int(rng.integers(0, 300)))
        p2 = p1 @ random_pose(rng, 45.0, 0.6)
        K = syn.scaled_K(syn.full_K(), 320.0 / W).clone()
        fs = float(rng.choice([1.0, 0.5, 0.25]))
        K[:, 0, 0] *= fs
        K[:, 1, 1] *= fs
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        orc.cost_volume_fusion(ac, [bc], p1, [p2], K, 0.25, 20.0, D, True).backward(go)
        expected = bc.grad[0].numpy().astype(np.float64)
        Hm, kt = pose_algebra.sweep_matrices_host(p1, [p2], K)
        got, whole_image, _ = gather_gradient(a[0].numpy().astype(np.float64), go[0].numpy().astype(np.float64), Hm[0, 0].numpy(),
                                              kt[0, 0].numpy(), D, 0.25, 20.0)
        scanned += whole_image > 0
        assert np.abs(got - expected).max() <= 2e-4 * max(1.0, np.abs(expected).max()), (trial, fs)
    assert scanned < 30          # the scan is the exception, not the rule
