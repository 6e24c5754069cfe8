"""CPU oracle for the plane-sweep depth hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch on the CPU, the arithmetic of the reference hot path
(ardaduz/deep-video-mvs).  It exists so that the HIP kernels in ``deep-video-mvs_amd/csrc`` can be
checked against something that is (a) independent of them and (b) itself pinned to the reference.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product package (``deep-video-mvs_amd/dvmvs``) never does: its ops raise when the HIP library is absent.

Pinning status
--------------
* cost volume / fusion / LSTM gates: PINNED.  ``tests/golden/make_goldens.py`` imports the reference
  (``/root/reference/dvmvs/utils.py:45-107``, ``dvmvs/convlstm.py:43-59``) in the build container and the
  fixtures it wrote under ``tests/golden/`` are compared against this file in ``tests/test_oracle_golden.py``.
* hidden-state warp and depth re-projection: PARITY UNPINNED AT THE KORNIA BOUNDARY.  The reference calls
  kornia==0.3.2 (``depth_to_3d``, ``transform_points``, ``project_points``, ``normalize_pixel_coordinates``;
  ``dvmvs/utils.py:122-136,241-256``), which is neither vendored under /root/reference nor installed.  The four
  functions are restated below from the library's published behaviour (pin-hole algebra; the one non-obvious
  rule is "divide by z only where |z| > 1e-8, else scale 1").  Everything around them (sort / first-occurrence
  splat, ReLU on z, masking, grid_sample) is reference code and is pinned by the same goldens.

All functions take/return CPU tensors.  ``dtype`` lets the tests evaluate the same algebra in float64 to
separate "different rounding" from "different algorithm".
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor

HOMOGENEOUS_EPS = 1e-8  # kornia.convert_points_from_homogeneous default eps


# ----------------------------------------------------------------------------------------------------------------------
# shared building blocks
# ----------------------------------------------------------------------------------------------------------------------
def pixel_grid(width: int, height: int, dtype=torch.float32) -> Tensor:
    """Homogeneous pixel grid, row-major (index = y*W + x), shape [3, H*W], rows (x, y, 1).

    Follows /root/reference/dvmvs/utils.py:34-42 (get_warp_grid_for_cost_volume_calculation).
    """
    ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dtype)], dim=0)


def bilinear_zeros_gather(src: Tensor, ix: Tensor, iy: Tensor) -> Tensor:
    """Bilinear sample of ``src`` [B,C,H,W] at *pixel* coordinates ``ix, iy`` [B,...] with zeros padding.

    This is what ``grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True)`` does once the
    normalised grid has been mapped back to pixels (used by /root/reference/dvmvs/utils.py:75-79 and :258).
    A tap contributes only if it lies inside [0,W-1]x[0,H-1]; non-finite coordinates contribute nothing.
    Output shape [B, C, *ix.shape[1:]].
    """
    B, C, H, W = src.shape
    out_shape = ix.shape[1:]
    ix = ix.reshape(B, -1)
    iy = iy.reshape(B, -1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    flat = src.reshape(B, C, H * W)
    out = torch.zeros(B, C, ix.shape[1], dtype=src.dtype)
    for xs, ys, w in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        inside = (xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1) & torch.isfinite(ix) & torch.isfinite(iy)
        xi = torch.where(inside, xs, torch.zeros_like(xs)).long()
        yi = torch.where(inside, ys, torch.zeros_like(ys)).long()
        lin = (yi * W + xi).unsqueeze(1).expand(B, C, -1)
        vals = torch.gather(flat, 2, lin)
        wz = torch.where(inside, w, torch.zeros_like(w)).unsqueeze(1)
        out = out + vals * wz
    return out.reshape(B, C, *out_shape)


def unnormalize_align_corners(g: Tensor, size: int) -> Tensor:
    """grid_sample's [-1,1] -> pixel mapping for align_corners=True: ((g + 1) / 2) * (size - 1)."""
    return ((g + 1.0) / 2.0) * (size - 1)


def plane_depths(min_depth: float, max_depth: float, n_depth_levels: int) -> List[float]:
    """Depth of every sweep plane, far -> near, uniform in inverse depth (python doubles).

    /root/reference/dvmvs/utils.py:59-66.
    """
    base = 1.0 / max_depth
    step = (1.0 / min_depth - 1.0 / max_depth) / (n_depth_levels - 1)
    return [1 / (base + i * step) for i in range(n_depth_levels)]


# ----------------------------------------------------------------------------------------------------------------------
# a2 / a3: plane-sweep cost volume
# ----------------------------------------------------------------------------------------------------------------------
# Diagnostic switch (tests/test_hybrid_parity.py): None = the reference's arithmetic, i.e. the small pose / intrinsics
# algebra (inverse(pose2) @ pose1, K R K^-1, K t, inverse(prev_pose) @ pose) in the tensors' own dtype, float32 in practice.
# Set to torch.float64 (``with exact_pose_algebra():``) those few matrices are evaluated in float64 and rounded once -- what
# the HIP kernels do on the device.  In float32 the inverse of a camera-to-world pose with translations of several metres
# carries ~5e-7 m of cancellation error in the relative translation, which the sweep turns into up to ~3e-4 px at the
# 0.25 m plane: by far the largest difference between "the reference" and "the exact result", and the only one the
# kernels do not reproduce.  The switch lets a test separate that term from everything else.
POSE_ALGEBRA_DTYPE = None


class exact_pose_algebra:
    def __enter__(self):
        global POSE_ALGEBRA_DTYPE
        self.saved, POSE_ALGEBRA_DTYPE = POSE_ALGEBRA_DTYPE, torch.float64

    def __exit__(self, *exc):
        global POSE_ALGEBRA_DTYPE
        POSE_ALGEBRA_DTYPE = self.saved


def relative_pose(a: Tensor, c: Tensor) -> Tensor:
    """inverse(a) @ c  (utils.py:51, :121; convlstm.py:30), in POSE_ALGEBRA_DTYPE when that is set."""
    if POSE_ALGEBRA_DTYPE is None:
        return torch.linalg.inv(a) @ c
    return (torch.linalg.inv(a.to(POSE_ALGEBRA_DTYPE)) @ c.to(POSE_ALGEBRA_DTYPE)).to(c.dtype)


def plane_sweep_setup(pose1: Tensor, pose2: Tensor, K: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-batch homography part ``K R K^-1`` [B,3,3] and translation part ``K t`` [B,3,1].

    /root/reference/dvmvs/utils.py:51-56: extrinsic2 = inv(pose2) @ pose1.
    """
    dt = POSE_ALGEBRA_DTYPE or pose1.dtype
    E = torch.linalg.inv(pose2.to(dt)) @ pose1.to(dt)
    R = E[:, 0:3, 0:3]
    t = E[:, 0:3, 3:4]
    Kd = K.to(dt)
    return (Kd @ R @ torch.linalg.inv(Kd)).to(pose1.dtype), (Kd @ t).to(pose1.dtype)


def cost_volume(image1: Tensor, image2: Tensor, pose1: Tensor, pose2: Tensor, K: Tensor,
                min_depth: float, max_depth: float, n_depth_levels: int, dot_product: bool = True) -> Tensor:
    """Plane-sweep cost volume [B,D,H,W] of one measurement frame against the reference frame.

    Restates /root/reference/dvmvs/utils.py:45-86 (calculate_cost_volume_by_warping) with all planes
    evaluated by one vectorised gather instead of a per-plane grid_sample loop.  Quirks kept on purpose:
    ``+1e-8`` in the perspective divide and no test on the sign of Z (:70); normalisation by W/2, H/2 (:72-73)
    combined with align_corners=True, i.e. an effective sample position u*(W-1)/W; dot mode divides by C (:82),
    SAD mode does not (:84).
    """
    B, C, H, W = image1.shape
    dt = image1.dtype
    KRKinv, Kt = plane_sweep_setup(pose1, pose2, K)
    base = KRKinv @ pixel_grid(W, H, dt).unsqueeze(0)                       # [B,3,HW]
    depths = torch.tensor(plane_depths(min_depth, max_depth, n_depth_levels), dtype=dt)
    # warping = K_R_Kinv_UV + Kt / this_depth, for every plane at once: [B,D,3,HW]
    warp = base.unsqueeze(1) + (Kt.unsqueeze(1) / depths.view(1, -1, 1, 1))
    denom = warp[:, :, 2] + 1e-8
    u = warp[:, :, 0] / denom
    v = warp[:, :, 1] / denom
    gx = (u - W / 2.0) / (W / 2.0)
    gy = (v - H / 2.0) / (H / 2.0)
    ix = unnormalize_align_corners(gx, W)                                   # [B,D,HW]
    iy = unnormalize_align_corners(gy, H)
    out = torch.empty(B, n_depth_levels, H, W, dtype=dt)
    ref = image1.reshape(B, C, 1, H * W)
    # chunk the planes to bound memory (C*D*HW floats otherwise)
    chunk = max(1, min(n_depth_levels, (1 << 24) // max(1, C * H * W)))
    for d0 in range(0, n_depth_levels, chunk):
        d1 = min(n_depth_levels, d0 + chunk)
        warped = bilinear_zeros_gather(image2, ix[:, d0:d1], iy[:, d0:d1])  # [B,C,d,HW]
        if dot_product:
            out[:, d0:d1] = ((ref * warped).sum(dim=1) / C).reshape(B, d1 - d0, H, W)
        else:
            out[:, d0:d1] = (ref - warped).abs().sum(dim=1).reshape(B, d1 - d0, H, W)
    return out


def cost_volume_planewise(image1: Tensor, image2: Tensor, pose1: Tensor, pose2: Tensor, K: Tensor,
                          min_depth: float, max_depth: float, n_depth_levels: int, dot_product: bool = True) -> Tensor:
    """Same result as ``cost_volume`` computed plane by plane with torch's own ``grid_sample`` -- the efficient way to run
    this op on a CPU (one fused bilinear kernel per plane instead of four index gathers), used for the ``cpu_baseline`` leg
    of bench.py so that the CPU number is not handicapped by the oracle's vectorised-gather formulation.
    Loop structure as in /root/reference/dvmvs/utils.py:65-84; tests assert it equals ``cost_volume``."""
    B, C, H, W = image1.shape
    dt = image1.dtype
    KRKinv, Kt = plane_sweep_setup(pose1, pose2, K)
    base = KRKinv @ pixel_grid(W, H, dt).unsqueeze(0)
    out = torch.empty(B, n_depth_levels, H, W, dtype=dt)
    for i, depth in enumerate(plane_depths(min_depth, max_depth, n_depth_levels)):
        warp = (base + Kt / depth).transpose(1, 2)
        uv = warp[:, :, 0:2] / (warp[:, :, 2:3] + 1e-8)
        grid = torch.stack([(uv[..., 0] - W / 2.0) / (W / 2.0), (uv[..., 1] - H / 2.0) / (H / 2.0)], dim=-1).view(B, H, W, 2)
        warped = torch.nn.functional.grid_sample(image2, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        out[:, i] = (image1 * warped).sum(dim=1) / C if dot_product else (image1 - warped).abs().sum(dim=1)
    return out


def cost_volume_fusion(image1: Tensor, image2s: Sequence[Tensor], pose1: Tensor, pose2s: Sequence[Tensor], K: Tensor,
                       min_depth: float, max_depth: float, n_depth_levels: int, dot_product: bool = True,
                       planewise: bool = False) -> Tensor:
    """Mean of ``cost_volume`` over the measurement frames (sum, then one division).

    /root/reference/dvmvs/utils.py:89-107.
    """
    B, C, H, W = image1.shape
    fused = torch.zeros(B, n_depth_levels, H, W, dtype=image1.dtype)
    one = cost_volume_planewise if planewise else cost_volume
    for image2, pose2 in zip(image2s, pose2s):
        fused = fused + one(image1, image2, pose1, pose2, K, min_depth, max_depth, n_depth_levels, dot_product)
    return fused / len(pose2s)


# ----------------------------------------------------------------------------------------------------------------------
# kornia==0.3.2 pin-hole helpers (restated; see module docstring: unpinned)
# ----------------------------------------------------------------------------------------------------------------------
def from_homogeneous(p: Tensor, eps: float = HOMOGENEOUS_EPS) -> Tensor:
    z = p[..., -1:]
    scale = torch.where(z.abs() > eps, 1.0 / z, torch.ones_like(z))
    return scale * p[..., :-1]


def depth_to_points(depth: Tensor, K: Tensor) -> Tensor:
    """kornia.depth_to_3d(depth[B,1,H,W], K[B,3,3], normalize_points=False), returned as [B,H,W,3]."""
    B, _, H, W = depth.shape
    dt = depth.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    fx = K[:, 0, 0].view(B, 1, 1)
    fy = K[:, 1, 1].view(B, 1, 1)
    cx = K[:, 0, 2].view(B, 1, 1)
    cy = K[:, 1, 2].view(B, 1, 1)
    x = (xs.unsqueeze(0) - cx) / fx
    y = (ys.unsqueeze(0) - cy) / fy
    rays = torch.stack([x, y, torch.ones_like(x)], dim=-1)                 # [B,H,W,3]
    return rays * depth.permute(0, 2, 3, 1)


def rigid_transform(T: Tensor, pts: Tensor) -> Tensor:
    """kornia.transform_points(T[:, None], pts[B,H,W,3]): homogeneous multiply, then de-homogenise."""
    B = pts.shape[0]
    ph = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1)           # [B,H,W,4]
    out = torch.einsum("bij,bhwj->bhwi", T, ph)
    return from_homogeneous(out)


def project(pts: Tensor, K: Tensor) -> Tensor:
    """kornia.project_points(pts[B,...,3], K[B,3,3]) -> [B,...,2] pixel coordinates (u, v)."""
    B = pts.shape[0]
    xy = from_homogeneous(pts)
    shape = [B] + [1] * (pts.dim() - 2)
    u = xy[..., 0] * K[:, 0, 0].view(shape) + K[:, 0, 2].view(shape)
    v = xy[..., 1] * K[:, 1, 1].view(shape) + K[:, 1, 2].view(shape)
    return torch.stack([u, v], dim=-1)


# ----------------------------------------------------------------------------------------------------------------------
# a4: forward splat of the previous depth map into the current half-resolution view
# ----------------------------------------------------------------------------------------------------------------------
def reproject_depth(reference_pose: Tensor, measurement_pose: Tensor, previous_depth: Tensor,
                    full_K: Tensor, half_K: Tensor, original_width: int, original_height: int) -> Tensor:
    """Z-buffered forward splat, farthest z wins, untouched pixels 0.  Returns [B,1,H/2,W/2].

    Restates /root/reference/dvmvs/utils.py:110-154 (get_non_differentiable_rectangle_depth_estimation).
    The reference sorts by relu(z) descending and keeps the first occurrence per target pixel
    (``np.unique(return_index=True)``); that is "max of relu(z) over the points landing on the pixel", which is
    what is computed here with an order-independent scatter-max.  Note the projection uses the *un-clamped* z
    (:134-136) while the stored value is relu(z) (:129).
    """
    B = reference_pose.shape[0]
    hw, hh = int(original_width / 2), int(original_height / 2)
    trans = relative_pose(reference_pose, measurement_pose)
    pts = rigid_transform(trans, depth_to_points(previous_depth, full_K)).reshape(B, -1, 3)
    z = torch.relu(pts[..., 2])
    proj = torch.round(project(pts, half_K))
    j, i = proj[..., 0], proj[..., 1]
    ok = torch.isfinite(j) & torch.isfinite(i) & (j >= 0) & (i >= 0) & (j < hw) & (i < hh)
    out = torch.zeros(B, hh * hw, dtype=previous_depth.dtype)
    lin = torch.where(ok, i * hw + j, torch.zeros_like(j)).long()
    zz = torch.where(ok, z, torch.zeros_like(z))
    out = out.scatter_reduce(1, lin, zz, reduce="amax", include_self=True)
    return out.reshape(B, 1, hh, hw)


def nearest_downsample(x: Tensor, factor: int) -> Tensor:
    """F.interpolate(x, scale_factor=1/factor, mode='nearest') for integer factors: rows/cols 0, f, 2f, ...

    Call site /root/reference/dvmvs/fusionnet/run-testing.py:187-189.
    """
    return x[..., ::factor, ::factor].contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# a5: depth-conditioned warp of the hidden state
# ----------------------------------------------------------------------------------------------------------------------
def warp_hidden_state(image_src: Tensor, depth_dst: Tensor, src_trans_dst: Tensor, camera_matrix: Tensor,
                      zero_invalid: bool = False) -> Tensor:
    """Inverse warp of ``image_src`` [B,C,H,W] into the destination view given the destination depth.

    Restates /root/reference/dvmvs/utils.py:205-258 (warp_frame_depth, normalize_points=False, bilinear).
    With ``zero_invalid`` it also applies the caller's mask ``h[depth <= 0.01] = 0``
    (/root/reference/dvmvs/convlstm.py:32-41).
    """
    B, C, H, W = image_src.shape
    pts = rigid_transform(src_trans_dst, depth_to_points(depth_dst, camera_matrix))
    pts = torch.cat([pts[..., :2], torch.relu(pts[..., 2:3])], dim=-1)
    uv = project(pts, camera_matrix)
    # kornia.normalize_pixel_coordinates: factor = 2 / (size - 1).clamp(eps);  p * factor - 1
    fx = 2.0 / max(W - 1, HOMOGENEOUS_EPS)
    fy = 2.0 / max(H - 1, HOMOGENEOUS_EPS)
    gx = uv[..., 0] * fx - 1
    gy = uv[..., 1] * fy - 1
    out = bilinear_zeros_gather(image_src, unnormalize_align_corners(gx, W), unnormalize_align_corners(gy, H))
    if zero_invalid:
        # the reference zeroes through ``.data`` (convlstm.py:41): the forward value is masked, the gradient is not
        keep = (depth_dst > 0.01).to(out.dtype)
        out = out + (out * keep - out).detach()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# a6: ConvLSTM gates
# ----------------------------------------------------------------------------------------------------------------------
def spatial_layer_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """torch.layer_norm(x, [H, W]) without affine: per (b, channel), biased variance."""
    mean = x.mean(dim=(-2, -1), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(-2, -1), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def celu(x: Tensor) -> Tensor:
    return torch.where(x > 0, x, torch.expm1(torch.clamp(x, max=0)))


def lstm_gates(combined_conv: Tensor, c_cur: Tensor) -> Tuple[Tensor, Tensor]:
    """Gate fusion given the 4*hidden channel conv output; split order i, f, o, g.

    Restates /root/reference/dvmvs/convlstm.py:45-59 with activation_function=torch.celu
    (/root/reference/dvmvs/fusionnet/model.py:316-319).  Returns (h_next, c_next); c_next is the normalised one.
    """
    hid = c_cur.shape[1]
    cc_i, cc_f, cc_o, cc_g = torch.split(combined_conv, hid, dim=1)
    i = torch.sigmoid(cc_i)
    f = torch.sigmoid(cc_f)
    o = torch.sigmoid(cc_o)
    g = celu(spatial_layer_norm(cc_g))
    c_next = spatial_layer_norm(f * c_cur + i * g)
    h_next = o * celu(c_next)
    return h_next, c_next


def convlstm_cell(conv_weight: Tensor, x: Tensor, h_cur: Tensor, c_cur: Tensor, previous_pose: Optional[Tensor],
                  current_pose: Tensor, estimated_current_depth: Tensor, camera_matrix: Tensor) -> Tuple[Tensor, Tensor]:
    """Whole cell forward (/root/reference/dvmvs/convlstm.py:26-59): warp+mask, 3x3 conv (no bias), gates."""
    if previous_pose is not None:
        T = relative_pose(previous_pose, current_pose)
        h_cur = warp_hidden_state(h_cur, estimated_current_depth, T, camera_matrix, zero_invalid=True)
    pad = (conv_weight.shape[2] // 2, conv_weight.shape[3] // 2)
    cc = torch.nn.functional.conv2d(torch.cat([x, h_cur], dim=1), conv_weight, bias=None, padding=pad)
    return lstm_gates(cc, c_cur)
