"""Known-answer tests for the kornia==0.3.2 helpers that the reference's hidden-state warp and depth re-projection call
(/root/reference/dvmvs/utils.py:122-136, :241-256).  kornia is not vendored in the reference and not installable here, so
oracle/dvmvs_oracle.py restates the four functions; the reference-generated goldens cannot pin them (the golden generator
feeds the imported reference those same restatements).  These tests pin them INDEPENDENTLY of the oracle: every expected
value below is worked out by hand (or by the few-line scalar evaluator ``by_the_book`` that transcribes the library's
documented formulas point by point), and both the CPU oracle and -- on the GPU -- the HIP kernels are held to it.

kornia 0.3.2, as used by the reference:
* ``depth_to_3d(depth[B,1,H,W], K, normalize_points=False)``: pixel grid x in [0, W-1], y in [0, H-1] (not normalised);
  P = depth * [(x - cx) / fx, (y - cy) / fy, 1].
* ``transform_points(T, P)``: [P, 1] is multiplied by T, then ``convert_points_from_homogeneous`` divides by w.
* ``convert_points_from_homogeneous(p, eps=1e-8)``: scale = 1 / z WHERE |z| > eps, ELSE 1;  returns scale * p[..., :-1].
  (So a point with z == 0, e.g. after the reference's relu(z), is NOT sent to infinity: it keeps its x, y.)
* ``project_points(P, K)``: (x', y') = convert_points_from_homogeneous(P);  u = fx x' + cx,  v = fy y' + cy.
* ``normalize_pixel_coordinates(p, H, W, eps=1e-8)``: p * 2 / max(size - 1, eps) - 1   (size = W for x, H for y).
* the reference then calls grid_sample(bilinear, zeros, align_corners=True): pixel = (g + 1) / 2 * (size - 1).
"""
import math

import numpy as np
import pytest
import torch

import dvmvs_oracle as orc

EPS = 1e-8


def from_h(p):
    z = p[-1]
    scale = 1.0 / z if abs(z) > EPS else 1.0
    return [scale * v for v in p[:-1]]


def by_the_book_warp(src, depth, T, K):
    """Scalar transcription of warp_frame_depth (utils.py:205-258) + the caller's mask (convlstm.py:32-41), float64."""
    C, H, W = src.shape
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    out = np.zeros((C, H, W))
    for y in range(H):
        for x in range(W):
            d = depth[y][x]
            P = [d * (x - cx) / fx, d * (y - cy) / fy, d]                          # depth_to_3d
            Q = from_h([sum(T[r][c] * v for c, v in enumerate(P + [1.0])) for r in range(4)])   # transform_points
            Q[2] = max(Q[2], 0.0)                                                  # torch.relu(z), utils.py:244-247
            xn, yn = from_h(Q)                                                     # project_points
            u, v = fx * xn + cx, fy * yn + cy
            gx = u * 2.0 / max(W - 1, EPS) - 1.0                                   # normalize_pixel_coordinates
            gy = v * 2.0 / max(H - 1, EPS) - 1.0
            ix, iy = (gx + 1.0) / 2.0 * (W - 1), (gy + 1.0) / 2.0 * (H - 1)        # grid_sample, align_corners=True
            if not (math.isfinite(ix) and math.isfinite(iy)):
                continue
            x0, y0 = math.floor(ix), math.floor(iy)
            for (xx, yy, w) in ((x0, y0, (x0 + 1 - ix) * (y0 + 1 - iy)), (x0 + 1, y0, (ix - x0) * (y0 + 1 - iy)),
                                (x0, y0 + 1, (x0 + 1 - ix) * (iy - y0)), (x0 + 1, y0 + 1, (ix - x0) * (iy - y0))):
                if 0 <= xx < W and 0 <= yy < H:
                    out[:, y, x] += w * src[:, yy, xx]
            if d <= 0.01:
                out[:, y, x] = 0.0
    return out


def by_the_book_splat(depth, T, full_K, half_K):
    """Scalar transcription of get_non_differentiable_rectangle_depth_estimation (utils.py:110-154): farthest z wins."""
    H, W = depth.shape
    hh, hw = H // 2, W // 2
    out = np.zeros((hh, hw))
    for y in range(H):
        for x in range(W):
            d = float(depth[y][x])
            P = [d * (x - full_K[0][2]) / full_K[0][0], d * (y - full_K[1][2]) / full_K[1][1], d]
            Q = from_h([sum(T[r][c] * v for c, v in enumerate(P + [1.0])) for r in range(4)])
            z = max(Q[2], 0.0)
            xn, yn = from_h(Q)                        # un-clamped z in the projection (:134-136)
            u, v = half_K[0][0] * xn + half_K[0][2], half_K[1][1] * yn + half_K[1][2]
            if not (math.isfinite(u) and math.isfinite(v)):
                continue
            j, i = round(u), round(v)                 # Python's round is half-to-even, like torch.round
            if 0 <= j < hw and 0 <= i < hh:
                out[i][j] = max(out[i][j], z)
    return out


def K_of(fx, fy, cx, cy):
    return [[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]


def translation(tx, ty, tz):
    return [[1.0, 0.0, 0.0, tx], [0.0, 1.0, 0.0, ty], [0.0, 0.0, 1.0, tz], [0.0, 0.0, 0.0, 1.0]]


def t32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def warp_cases():
    rng = np.random.RandomState(5)
    src = rng.randn(3, 8, 10)
    K = K_of(8.0, 6.0, 4.5, 3.5)
    cases = {}
    # 1. identity transform: every pixel samples itself
    cases["identity"] = (src, np.full((8, 10), 2.0), translation(0, 0, 0), K, src.copy())
    # 2. the principal point is not a pixel (4.5, 3.5); a pure x translation with fx * tx / d == 1 shifts by exactly one pixel
    shifted = np.zeros_like(src)
    shifted[:, :, :-1] = src[:, :, 1:]
    cases["one_pixel_shift"] = (src, np.full((8, 10), 2.0), translation(0.25, 0, 0), K, shifted)
    # 3. z == 0 after the transform (tz = -d): scale 1 instead of a division -> u = fx X + cx = d (x - cx) + cx, v likewise.
    #    d = 2: u = 2x - 4.5, v = 2y - 3.5: bilinear between four pixels at half-integer positions, zeros outside
    cases["z_exactly_zero"] = (src, np.full((8, 10), 2.0), translation(0, 0, -2.0), K, None)
    # 4. behind the camera (tz = -2d): relu sets z = 0, same rule as 3 (|0| <= eps -> scale 1)
    cases["behind_camera"] = (src, np.full((8, 10), 2.0), translation(0, 0, -4.0), K, None)
    # 5. invalid depth (<= 0.01) is zeroed by the caller's mask even where the sample position is fine
    d = np.full((8, 10), 2.0)
    d[2, 3] = 0.01
    d[5, 7] = 0.0
    exp = src.copy()
    exp[:, 2, 3] = 0.0
    exp[:, 5, 7] = 0.0
    cases["mask"] = (src, d, translation(0, 0, 0), K, exp)
    return cases


def check_hand_values():
    """Two values of case 3 worked out on paper, so that the scalar evaluator itself is pinned."""
    src, depth, T, K, _ = warp_cases()["z_exactly_zero"]
    out = by_the_book_warp(src, depth, T, K)
    # pixel (x=4, y=3): u = 2*4 - 4.5 = 3.5, v = 2*3 - 3.5 = 2.5 -> mean of src[:, 2:4, 3:5]
    np.testing.assert_allclose(out[:, 3, 4], src[:, 2:4, 3:5].mean(axis=(1, 2)), atol=1e-12)
    # pixel (x=0, y=0): u = -4.5 -> all four taps outside -> 0
    assert np.all(out[:, 0, 0] == 0.0)
    return out


def test_scalar_evaluator_reproduces_hand_computed_values():
    check_hand_values()
    src, depth, T, K, exp = warp_cases()["one_pixel_shift"]
    np.testing.assert_allclose(by_the_book_warp(src, depth, T, K), exp, atol=1e-12)


@pytest.mark.parametrize("name", sorted(warp_cases().keys()))
def test_oracle_hidden_warp_known_answers(name):
    src, depth, T, K, exp = warp_cases()[name]
    exp = by_the_book_warp(src, depth, T, K) if exp is None else exp
    got = orc.warp_hidden_state(t32(src)[None], t32(depth)[None, None], t32(T)[None], t32(K)[None], zero_invalid=True)
    np.testing.assert_allclose(got[0].numpy(), exp, atol=2e-6)


def test_oracle_hidden_warp_degenerate_sizes():
    """W == 1 / H == 1: normalize_pixel_coordinates divides by max(size - 1, 1e-8) and align_corners multiplies by
    (size - 1) == 0, so every finite coordinate lands on pixel 0 of that axis."""
    rng = np.random.RandomState(6)
    for (H, W) in ((1, 6), (5, 1)):
        src = rng.randn(2, H, W)
        depth = np.full((H, W), 1.5)
        T = translation(0.3, -0.2, 0.1)
        K = K_of(3.0, 2.0, 0.5 * (W - 1), 0.5 * (H - 1))
        exp = by_the_book_warp(src, depth, T, K)
        got = orc.warp_hidden_state(t32(src)[None], t32(depth)[None, None], t32(T)[None], t32(K)[None], zero_invalid=True)
        np.testing.assert_allclose(got[0].numpy(), exp, atol=2e-6)


def splat_cases():
    full_K = K_of(8.0, 6.0, 5.0, 4.0)
    half_K = K_of(4.0, 3.0, 2.5, 2.0)
    cases = {}
    # identity pose, depth 1 + (x odd) + 2 (y odd): full-res pixel (x, y) lands on (x / 2, y / 2); halves round to even, the
    # farthest point wins.  Hand value: target column 2 receives x = 3 (1.5 -> 2), 4 (2.0), 5 (2.5 -> 2): odd x carry +1
    xs, ys = np.meshgrid(np.arange(12), np.arange(8))
    cases["parity"] = (1.0 + (xs % 2) + 2.0 * (ys % 2), translation(0, 0, 0), full_K, half_K)
    # a translation towards the camera by more than some depths: those points end up behind it (z < 0 -> stored relu(z) = 0,
    # projected with the negative z), the others move outwards
    cases["partly_behind"] = (0.5 + 0.25 * xs + 0.0 * ys, translation(0.1, 0.0, -1.5), full_K, half_K)
    # z exactly 0 for one column (depth 1.5, tz = -1.5): projection falls back to scale 1
    cases["z_zero_column"] = (np.where(xs == 4, 1.5, 3.0) + 0.0 * ys, translation(0, 0, -1.5), full_K, half_K)
    return cases


def test_splat_hand_value():
    depth, T, fK, hK = splat_cases()["parity"]
    out = by_the_book_splat(depth, T, fK, hK)
    # target (row 0, column 2): sources y in {0} (0.0 -> 0; y = 1 -> 0.5 -> 0 as well, +2), x in {3, 4, 5} -> max = 1 + 1 + 2
    assert out[0][2] == 4.0
    # target (row 1, column 1): only y = 2 (1.0) and x = 2 (1.0) land there (1.5 and 0.5 round to the even neighbours) -> 1
    assert out[1][1] == 1.0


@pytest.mark.parametrize("name", sorted(splat_cases().keys()))
def test_oracle_depth_reprojection_known_answers(name):
    depth, T, fK, hK = splat_cases()[name]
    exp = by_the_book_splat(depth, T, fK, hK)
    H, W = depth.shape
    # reference_pose^-1 * measurement_pose == T  with reference_pose = identity
    got = orc.reproject_depth(torch.eye(4)[None], t32(T)[None], t32(depth)[None, None], t32(fK)[None], t32(hK)[None], W, H)
    np.testing.assert_allclose(got[0, 0].numpy(), exp, atol=1e-6)


# ---- the same answers from the HIP kernels (through the C ABI) ---------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(warp_cases().keys()))
def test_hip_hidden_warp_known_answers(hip_device, name):
    from dvmvs.hip import ops
    src, depth, T, K, exp = warp_cases()[name]
    exp = by_the_book_warp(src, depth, T, K) if exp is None else exp
    dev = hip_device
    got = ops.hidden_warp(t32(src)[None].to(dev), t32(depth)[None, None].to(dev), t32(T)[None].to(dev), t32(K)[None].to(dev), True)
    np.testing.assert_allclose(got[0].cpu().numpy(), exp, atol=2e-6)


@pytest.mark.gpu
def test_hip_hidden_warp_degenerate_sizes(hip_device):
    from dvmvs.hip import ops
    rng = np.random.RandomState(6)
    dev = hip_device
    for (H, W) in ((1, 6), (5, 1)):
        src = rng.randn(2, H, W)
        depth = np.full((H, W), 1.5)
        T = translation(0.3, -0.2, 0.1)
        K = K_of(3.0, 2.0, 0.5 * (W - 1), 0.5 * (H - 1))
        exp = by_the_book_warp(src, depth, T, K)
        got = ops.hidden_warp(t32(src)[None].to(dev), t32(depth)[None, None].to(dev), t32(T)[None].to(dev), t32(K)[None].to(dev), True)
        np.testing.assert_allclose(got[0].cpu().numpy(), exp, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(splat_cases().keys()))
def test_hip_depth_reprojection_known_answers(hip_device, name):
    from dvmvs.hip import ops
    depth, T, fK, hK = splat_cases()[name]
    exp = by_the_book_splat(depth, T, fK, hK)
    dev = hip_device
    # reference pose = identity, measurement pose = T: the splat transform inverse(reference) @ measurement is T itself
    got = ops.depth_reproject(t32(T)[None].to(dev), t32(depth)[None, None].to(dev), t32(fK)[None].to(dev),
                              t32(hK)[None].to(dev))
    np.testing.assert_allclose(got[0, 0].cpu().numpy(), exp, atol=1e-6)
