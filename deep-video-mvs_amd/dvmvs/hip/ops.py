"""``torch.library`` custom ops over the C ABI of ``libdvmvs_hip.so``.

Each op is a thin shim: check device/dtype, make inputs contiguous, allocate the output with torch, pass raw device
pointers and the current HIP stream to the ``extern "C"`` launcher.  No op synchronises or touches the host, so a
whole frame can be captured into a hipGraph (``torch.cuda.graph``).  Tensors that are not on the GPU are rejected:
the CPU restatement lives in ``oracle/`` and is test infrastructure only.
"""
import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from dvmvs.hip import _capi

__all__ = ["cost_volume", "sweep_matrices", "hidden_warp", "relative_pose", "lstm_gates", "depth_reproject", "depth_reproject_lowres",
           "bias_act_", "upsample2x", "depthwise_conv"]


# two-pass tiled sweep (spill list in the workspace): see dvmvs_cost_volume_workspace_bytes_two_pass in the header
COST_VOLUME_TWO_PASS = os.environ.get("DVMVS_COST_VOLUME_TWO_PASS", "1") == "1"


_workspaces = {}


def sweep_workspace(device, B, M, H, W, D):
    """Persistent spill workspace of the two-pass sweep for (device, shape): zero-filled once here and left zeroed by every
    call (contract in include/dvmvs_hip.h).  Returns (tensor, bytes).

    One buffer per device and shape: calls that use it must be ordered with respect to each other, which is the case for
    the one-process-per-GPU, one-stream-per-process model of this package (a hipGraph replay runs on the stream that
    launches it).  Code that issues cost volumes of the same shape on several streams concurrently must pass its own
    workspaces through the C ABI.  Allocated outside any graph capture when the first (eager / warm-up) call happens."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, B, M, H, W, D)
    entry = _workspaces.get(key)
    if entry is None:
        nbytes = _capi.lib().dvmvs_cost_volume_workspace_bytes(B, M, H, W, D)
        entry = (torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device), nbytes)
        _workspaces[key] = entry
    return entry


def drop_sweep_workspace(device, B, M, H, W, D):
    """Forgets the cached workspace of (device, shape).  Called when a cost-volume call fails: its contract (header words zero
    between calls) may no longer hold, and a stale group count would silently corrupt every later volume of that shape."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    _workspaces.pop((index, B, M, H, W, D), None)


def _no_cpu(op):
    raise RuntimeError(f"dvmvs::{op} only runs as a HIP kernel on an MI355X; move the tensors to the GPU. "
                       f"There is deliberately no CPU fallback (the CPU oracle under oracle/ is for tests).")


def _dev_f32(name, *tensors):
    for t in tensors:
        if t.device.type != "cuda":
            _no_cpu(name)
        if t.dtype != torch.float32:
            raise TypeError(f"dvmvs::{name}: expected float32 tensors, got {t.dtype}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return t.data_ptr()


# ----------------------------------------------------------------------------------------------------------------------
# fused plane-sweep cost volume
# ----------------------------------------------------------------------------------------------------------------------
def _check_sweep_matrices(name, Hm, kt, B, M):
    if tuple(Hm.shape) != (B, M, 9) or tuple(kt.shape) != (B, M, 3):
        raise ValueError(f"dvmvs::{name}: expected Hm [{B},{M},9] and kt [{B},{M},3] (dvmvs.pose_algebra.sweep_matrices), "
                         f"got {tuple(Hm.shape)} and {tuple(kt.shape)}")


def sweep_work_list_words(B: int, H: int, W: int, D: int) -> int:
    """32-bit words of a sweep work list for this shape (dvmvs_sweep_work_list_bytes / 4)."""
    return int(_capi.lib().dvmvs_sweep_work_list_bytes(int(B), int(H), int(W), int(D))) // 4


def sweep_work_list_host(Hm: Tensor, kt: Tensor, H: int, W: int, D: int, min_depth: float, max_depth: float, variant: int, out: Optional[Tensor] = None):
    """The tiled sweep's work list for HOST matrices ``Hm`` [B,M,9] / ``kt`` [B,M,3] (dvmvs_sweep_work_list: (tile, chunk) pairs whose
    sample boxes had to be halved are cut into plane sub-ranges that separate workgroups process in parallel), as an int32 host tensor
    (``out``: a caller-owned one, e.g. a slice of pinned staging memory).  ``variant``: the launch's (3 = wide configuration)."""
    Hm, kt = Hm.contiguous(), kt.contiguous()
    if Hm.device.type != "cpu" or Hm.dtype != torch.float32 or kt.dtype != torch.float32:
        raise ValueError("sweep_work_list_host needs the float32 HOST copies of the sweep matrices")
    B, M = Hm.shape[0], Hm.shape[1]
    words = sweep_work_list_words(B, H, W, D)
    if out is None:
        out = torch.empty(words, dtype=torch.int32)
    if out.device.type != "cpu" or out.dtype != torch.int32 or out.numel() < words or not out.is_contiguous():
        raise ValueError(f"work list buffer must be a contiguous int32 host tensor of at least {words} words")
    used = _capi.lib().dvmvs_sweep_work_list(Hm.data_ptr(), kt.data_ptr(), B, M, int(H), int(W), int(D), float(min_depth), float(max_depth),
                                             1 if variant in (3, 5) else 0, out.data_ptr(), out.numel() * 4)
    if used < 0:
        _capi.check(used, "dvmvs_sweep_work_list")
    return out


def sweep_plan_host(Hm: Tensor, kt: Tensor, H: int, W: int, D: int, min_depth: float, max_depth: float, variant: int, out: Tensor,
                    allow_mfma: bool = False) -> int:
    """Configuration choice (``variant`` 0; or 2 / 3 as given) and work list in one walk (dvmvs_sweep_plan): fills ``out`` (an int32 host
    tensor of ``sweep_work_list_words``) and returns the variant to launch with."""
    Hm, kt = Hm.contiguous(), kt.contiguous()
    if Hm.device.type != "cpu" or Hm.dtype != torch.float32 or kt.dtype != torch.float32:
        raise ValueError("sweep_plan_host needs the float32 HOST copies of the sweep matrices")
    if out.device.type != "cpu" or out.dtype != torch.int32 or not out.is_contiguous() or out.numel() < sweep_work_list_words(Hm.shape[0], H, W, D):
        raise ValueError("work list buffer must be a contiguous int32 host tensor of sweep_work_list_words entries")
    if allow_mfma and int(variant) == 0:
        # ... with variant 6 (the correlate-then-interpolate sweep) as a candidate: for callers with 32-channel channels-last measurement maps
        chosen = _capi.lib().dvmvs_sweep_plan6(Hm.data_ptr(), kt.data_ptr(), Hm.shape[0], Hm.shape[1], int(H), int(W), int(D), float(min_depth),
                                               float(max_depth), out.data_ptr(), out.numel() * 4)
    else:
        chosen = _capi.lib().dvmvs_sweep_plan(Hm.data_ptr(), kt.data_ptr(), Hm.shape[0], Hm.shape[1], int(H), int(W), int(D), float(min_depth),
                                              float(max_depth), {4: 2, 5: 3}.get(int(variant), int(variant)), out.data_ptr(), out.numel() * 4)
    if chosen < 0:
        _capi.check(chosen, "dvmvs_sweep_plan")
    return chosen


def _work_list_ptr(work_list, image1, B, H, W, D):
    if work_list is None:
        return None
    if work_list.device != image1.device or work_list.dtype != torch.int32 or not work_list.is_contiguous() or \
            work_list.numel() < sweep_work_list_words(B, H, W, D):
        raise ValueError("dvmvs::cost_volume: work_list must be a contiguous int32 tensor of dvmvs_sweep_work_list_bytes on the features' device")
    return work_list.data_ptr()


def cost_volume(image1: Tensor, image2s: Sequence[Tensor], Hm: Tensor, kt: Tensor, min_depth: float, max_depth: float, n_depth_levels: int,
                dot_product: bool, variant: int, work_list: Optional[Tensor] = None) -> Tensor:
    """``Hm`` [B,M,9] = K R K^-1 and ``kt`` [B,M,3] = K t per (batch item, measurement frame): dvmvs.pose_algebra.  ``work_list``: the
    device copy of ``sweep_work_list_host``'s result for these matrices (optional; the tiled sweep then cuts long workgroups).
    Differentiable in the feature maps (dot-product mode)."""
    if work_list is None:      # (the registered op takes a tensor: an empty one = no list)
        work_list = torch.empty(0, dtype=torch.int32, device=image1.device)
    return _cost_volume_op(image1, image2s, Hm, kt, min_depth, max_depth, n_depth_levels, dot_product, variant, work_list)


@torch.library.custom_op("dvmvs::cost_volume", mutates_args=(), device_types="cuda")
def _cost_volume_op(image1: Tensor, image2s: Sequence[Tensor], Hm: Tensor, kt: Tensor,
                    min_depth: float, max_depth: float, n_depth_levels: int, dot_product: bool, variant: int, work_list: Tensor) -> Tensor:
    _dev_f32("cost_volume", image1, Hm, kt, *image2s)
    M = len(image2s)
    if M == 0:
        raise ValueError("dvmvs::cost_volume: need at least one measurement feature map")
    B, C, H, W = image1.shape
    _check_sweep_matrices("cost_volume", Hm, kt, B, M)
    for t in image2s:
        if t.shape != image1.shape:
            raise ValueError(f"dvmvs::cost_volume: measurement features {tuple(t.shape)} != reference {tuple(image1.shape)}")
    image1 = image1.contiguous()
    # Measurement maps that are already channels-last in memory (the frame engine caches them that way) are passed as NHWC;
    # anything else is made NCHW-contiguous, the reference's layout.
    nhwc_ok = bool(dot_product) and variant != 1 and C % 4 == 0 and C > 1 and (H * W >= 64 * 64 or (variant in (6, 7) and C <= 32))
    nhwc = nhwc_ok and all(t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous() for t in image2s)
    image2s = list(image2s) if nhwc else [t.contiguous() for t in image2s]
    Hm, kt = Hm.contiguous(), kt.contiguous()
    out = torch.empty((B, n_depth_levels, H, W), dtype=torch.float32, device=image1.device)
    lib = _capi.lib()
    workspace, ws_bytes = sweep_workspace(image1.device, B, M, H, W, n_depth_levels) if COST_VOLUME_TWO_PASS else (None, 0)
    items = _work_list_ptr(work_list if work_list.numel() else None, image1, B, H, W, n_depth_levels)
    with torch.cuda.device(image1.device):
        rc = lib.dvmvs_cost_volume_planned_fwd(
            _ptr(image1), _capi.pointer_array([_ptr(t) for t in image2s]), _ptr(Hm), _ptr(kt), _ptr(out),
            B, M, C, H, W, n_depth_levels, float(min_depth), float(max_depth), int(bool(dot_product)), int(variant),
            _capi.LAYOUT_NHWC if nhwc else _capi.LAYOUT_NCHW, _ptr(workspace) if workspace is not None else None, ws_bytes, items, _stream(image1))
    if rc != 0 and workspace is not None:
        drop_sweep_workspace(image1.device, B, M, H, W, n_depth_levels)
    _capi.check(rc, "dvmvs_cost_volume_planned_fwd")
    return out


@_cost_volume_op.register_fake
def _(image1, image2s, Hm, kt, min_depth, max_depth, n_depth_levels, dot_product, variant, work_list):
    B, C, H, W = image1.shape
    return image1.new_empty((B, n_depth_levels, H, W))


@_cost_volume_op.register_kernel("cpu")
def _(image1, image2s, Hm, kt, min_depth, max_depth, n_depth_levels, dot_product, variant, work_list):
    _no_cpu("cost_volume")


def _cost_volume_setup(ctx, inputs, output):
    image1, image2s, Hm, kt, min_depth, max_depth, n_depth_levels, dot_product, variant = inputs[:9]
    if not dot_product and (image1.requires_grad or any(t.requires_grad for t in image2s)):
        raise NotImplementedError("dvmvs::cost_volume: gradients are implemented for dot_product=True only")
    ctx.M = len(image2s)
    ctx.depth_range = (min_depth, max_depth, n_depth_levels)
    ctx.save_for_backward(image1, Hm, kt, *image2s)


def _cost_volume_backward(ctx, grad):
    saved = ctx.saved_tensors
    M = ctx.M
    image1, Hm, kt = saved[0], saved[1], saved[2]
    image2s = list(saved[3:3 + M])
    min_depth, max_depth, D = ctx.depth_range
    grad = grad.contiguous()
    B, C, H, W = image1.shape
    image1c = image1.contiguous()
    image2c = [t.contiguous() for t in image2s]
    g1 = torch.empty_like(image1c)
    flags = ctx.needs_input_grad[1]   # a tensor-list input: one flag per measurement map
    need2 = any(flags) if isinstance(flags, (list, tuple)) else bool(flags)
    g2 = [torch.zeros_like(t) for t in image2c] if need2 else []
    with torch.cuda.device(image1.device):
        rc = _capi.lib().dvmvs_cost_volume_bwd(
            _ptr(grad), _ptr(image1c), _capi.pointer_array([_ptr(t) for t in image2c]), _ptr(Hm.contiguous()), _ptr(kt.contiguous()),
            _ptr(g1), _capi.pointer_array([_ptr(t) for t in g2] if need2 else [None] * M),
            B, M, C, H, W, D, float(min_depth), float(max_depth), _stream(image1))
    _capi.check(rc, "dvmvs_cost_volume_bwd")
    return g1, (g2 if need2 else [None] * M), None, None, None, None, None, None, None, None


torch.library.register_autograd("dvmvs::cost_volume", _cost_volume_backward, setup_context=_cost_volume_setup)


@torch.library.custom_op("dvmvs::sweep_matrices", mutates_args=(), device_types="cuda")
def sweep_matrices(pose1: Tensor, pose2s: Sequence[Tensor], K: Tensor) -> Tuple[Tensor, Tensor]:
    """"exact" pose algebra: (Hm [B,M,9], kt [B,M,3]) in fp64 on the device, rounded once (dvmvs_sweep_matrices)."""
    _dev_f32("sweep_matrices", pose1, K, *pose2s)
    B, M = pose1.shape[0], len(pose2s)
    if M == 0 or tuple(pose1.shape) != (B, 4, 4) or tuple(K.shape) != (B, 3, 3) or any(tuple(p.shape) != (B, 4, 4) for p in pose2s):
        raise ValueError("dvmvs::sweep_matrices: expected pose1 [B,4,4], M >= 1 measurement poses [B,4,4] and K [B,3,3]")
    pose1, K = pose1.contiguous(), K.contiguous()
    pose2s = [p.contiguous() for p in pose2s]
    Hm = torch.empty((B, M, 9), dtype=torch.float32, device=pose1.device)
    kt = torch.empty((B, M, 3), dtype=torch.float32, device=pose1.device)
    with torch.cuda.device(pose1.device):
        rc = _capi.lib().dvmvs_sweep_matrices(_ptr(pose1), _capi.pointer_array([_ptr(p) for p in pose2s]), _ptr(K), _ptr(Hm), _ptr(kt),
                                              B, M, _stream(pose1))
    _capi.check(rc, "dvmvs_sweep_matrices")
    return Hm, kt


@sweep_matrices.register_fake
def _(pose1, pose2s, K):
    B, M = pose1.shape[0], len(pose2s)
    return pose1.new_empty((B, M, 9)), pose1.new_empty((B, M, 3))


@sweep_matrices.register_kernel("cpu")
def _(pose1, pose2s, K):
    _no_cpu("sweep_matrices")


# ----------------------------------------------------------------------------------------------------------------------
# hidden-state warp
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("dvmvs::hidden_warp", mutates_args=(), device_types="cuda")
def hidden_warp(image_src: Tensor, depth_dst: Tensor, src_trans_dst: Tensor, camera_matrix: Tensor,
                zero_invalid: bool) -> Tensor:
    _dev_f32("hidden_warp", image_src, depth_dst, src_trans_dst, camera_matrix)
    B, C, H, W = image_src.shape
    image_src, depth_dst = image_src.contiguous(), depth_dst.contiguous()
    src_trans_dst, camera_matrix = src_trans_dst.contiguous(), camera_matrix.contiguous()
    out = torch.empty_like(image_src)
    with torch.cuda.device(image_src.device):
        rc = _capi.lib().dvmvs_hidden_warp_fwd(_ptr(image_src), _ptr(depth_dst), _ptr(src_trans_dst), _ptr(camera_matrix),
                                               _ptr(out), B, C, H, W, int(bool(zero_invalid)), _stream(image_src))
    _capi.check(rc, "dvmvs_hidden_warp_fwd")
    return out


@hidden_warp.register_fake
def _(image_src, depth_dst, src_trans_dst, camera_matrix, zero_invalid):
    return torch.empty_like(image_src)


@hidden_warp.register_kernel("cpu")
def _(image_src, depth_dst, src_trans_dst, camera_matrix, zero_invalid):
    _no_cpu("hidden_warp")


def _hidden_warp_setup(ctx, inputs, output):
    image_src, depth_dst, src_trans_dst, camera_matrix, _ = inputs
    ctx.shape = tuple(image_src.shape)
    ctx.save_for_backward(depth_dst, src_trans_dst, camera_matrix)


def _hidden_warp_backward(ctx, grad):
    depth_dst, src_trans_dst, camera_matrix = ctx.saved_tensors
    B, C, H, W = ctx.shape
    grad = grad.contiguous()
    gsrc = torch.zeros_like(grad)
    with torch.cuda.device(grad.device):
        # the validity mask is NOT applied to the gradient, as in the reference (convlstm.py:41 edits .data)
        rc = _capi.lib().dvmvs_hidden_warp_bwd(_ptr(grad), _ptr(depth_dst.contiguous()), _ptr(src_trans_dst.contiguous()),
                                               _ptr(camera_matrix.contiguous()), _ptr(gsrc), B, C, H, W, _stream(grad))
    _capi.check(rc, "dvmvs_hidden_warp_bwd")
    return gsrc, None, None, None, None


torch.library.register_autograd("dvmvs::hidden_warp", _hidden_warp_backward, setup_context=_hidden_warp_setup)


# ----------------------------------------------------------------------------------------------------------------------
# inverse(a) @ c for 4x4 poses, fp64 on the device ("exact" pose algebra; the default path is dvmvs.pose_algebra's host fp32)
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("dvmvs::relative_pose", mutates_args=(), device_types="cuda")
def relative_pose(a: Tensor, c: Tensor) -> Tensor:
    _dev_f32("relative_pose", a, c)
    if a.shape != c.shape or a.shape[-2:] != (4, 4) or a.dim() != 3:
        raise ValueError(f"dvmvs::relative_pose: expected two [B,4,4] tensors, got {tuple(a.shape)} and {tuple(c.shape)}")
    a, c = a.contiguous(), c.contiguous()
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        rc = _capi.lib().dvmvs_relative_pose(_ptr(a), _ptr(c), _ptr(out), a.shape[0], _stream(a))
    _capi.check(rc, "dvmvs_relative_pose")
    return out


@relative_pose.register_fake
def _(a, c):
    return torch.empty_like(a)


@relative_pose.register_kernel("cpu")
def _(a, c):
    _no_cpu("relative_pose")


# ----------------------------------------------------------------------------------------------------------------------
# ConvLSTM gate fusion
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("dvmvs::lstm_gates", mutates_args=(), device_types="cuda")
def lstm_gates(combined_conv: Tensor, c_cur: Tensor) -> Tuple[Tensor, Tensor]:
    _dev_f32("lstm_gates", combined_conv, c_cur)
    B, hidden, H, W = c_cur.shape
    if combined_conv.shape != (B, 4 * hidden, H, W):
        raise ValueError(f"dvmvs::lstm_gates: conv output {tuple(combined_conv.shape)} does not match state {tuple(c_cur.shape)}")
    combined_conv, c_cur = combined_conv.contiguous(), c_cur.contiguous()
    h_next, c_next = torch.empty_like(c_cur), torch.empty_like(c_cur)
    with torch.cuda.device(c_cur.device):
        rc = _capi.lib().dvmvs_lstm_gates_fwd(_ptr(combined_conv), _ptr(c_cur), _ptr(h_next), _ptr(c_next), B, hidden, H, W,
                                              _stream(c_cur))
    _capi.check(rc, "dvmvs_lstm_gates_fwd")
    return h_next, c_next


@lstm_gates.register_fake
def _(combined_conv, c_cur):
    return torch.empty_like(c_cur), torch.empty_like(c_cur)


@lstm_gates.register_kernel("cpu")
def _(combined_conv, c_cur):
    _no_cpu("lstm_gates")


def _lstm_gates_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _lstm_gates_backward(ctx, grad_h, grad_c):
    combined_conv, c_cur = ctx.saved_tensors
    B, hidden, H, W = c_cur.shape
    cc, cs = combined_conv.contiguous(), c_cur.contiguous()
    gh = grad_h.contiguous() if grad_h is not None else None
    gc = grad_c.contiguous() if grad_c is not None else None
    grad_cc, grad_cs = torch.empty_like(cc), torch.empty_like(cs)
    with torch.cuda.device(cs.device):
        rc = _capi.lib().dvmvs_lstm_gates_bwd(_ptr(gh) if gh is not None else None, _ptr(gc) if gc is not None else None,
                                              _ptr(cc), _ptr(cs), _ptr(grad_cc), _ptr(grad_cs), B, hidden, H, W, _stream(cs))
    _capi.check(rc, "dvmvs_lstm_gates_bwd")
    return grad_cc, grad_cs


torch.library.register_autograd("dvmvs::lstm_gates", _lstm_gates_backward, setup_context=_lstm_gates_setup)


# ----------------------------------------------------------------------------------------------------------------------
# depth re-projection (forward splat)
# ----------------------------------------------------------------------------------------------------------------------
def _reproject(name, transformation, previous_depth, full_K, half_K, factor):
    _dev_f32(name, transformation, previous_depth, full_K, half_K)
    B, one, Hf, Wf = previous_depth.shape
    if one != 1:
        raise ValueError(f"dvmvs::{name}: previous depth must be [B,1,H,W], got {tuple(previous_depth.shape)}")
    if tuple(transformation.shape) != (B, 4, 4):
        raise ValueError(f"dvmvs::{name}: transformation must be [{B},4,4] (inverse(reference_pose) @ measurement_pose), "
                         f"got {tuple(transformation.shape)}")
    args = [t.contiguous() for t in (transformation, previous_depth, full_K, half_K)]
    out = torch.empty((B, 1, Hf // 2, Wf // 2), dtype=torch.float32, device=previous_depth.device)
    low = None
    if factor > 0:
        low = torch.empty((B, 1, (Hf // 2) // factor, (Wf // 2) // factor), dtype=torch.float32, device=previous_depth.device)
    with torch.cuda.device(previous_depth.device):
        rc = _capi.lib().dvmvs_depth_reproject_fwd(*[_ptr(t) for t in args], _ptr(out), _ptr(low) if low is not None else None,
                                                   int(factor), B, Hf, Wf, _stream(previous_depth))
    _capi.check(rc, "dvmvs_depth_reproject_fwd")
    return out, low


@torch.library.custom_op("dvmvs::depth_reproject", mutates_args=(), device_types="cuda")
def depth_reproject(transformation: Tensor, previous_depth: Tensor, full_K: Tensor, half_K: Tensor) -> Tensor:
    """``transformation`` [B,4,4] = inverse(reference_pose) @ measurement_pose (dvmvs.pose_algebra.relative_pose)."""
    return _reproject("depth_reproject", transformation, previous_depth, full_K, half_K, 0)[0]


@depth_reproject.register_fake
def _(transformation, previous_depth, full_K, half_K):
    B, _, Hf, Wf = previous_depth.shape
    return previous_depth.new_empty((B, 1, Hf // 2, Wf // 2))


@depth_reproject.register_kernel("cpu")
def _(transformation, previous_depth, full_K, half_K):
    _no_cpu("depth_reproject")


@torch.library.custom_op("dvmvs::depth_reproject_lowres", mutates_args=(), device_types="cuda")
def depth_reproject_lowres(transformation: Tensor, previous_depth: Tensor, full_K: Tensor, half_K: Tensor,
                           factor: int) -> Tuple[Tensor, Tensor]:
    """Splat plus the nearest /factor decimation the fusionnet call site applies next (one C-ABI call)."""
    if factor <= 0:
        raise ValueError("dvmvs::depth_reproject_lowres: factor must be positive")
    out, low = _reproject("depth_reproject_lowres", transformation, previous_depth, full_K, half_K, factor)
    return out, low


@depth_reproject_lowres.register_fake
def _(transformation, previous_depth, full_K, half_K, factor):
    B, _, Hf, Wf = previous_depth.shape
    return (previous_depth.new_empty((B, 1, Hf // 2, Wf // 2)),
            previous_depth.new_empty((B, 1, (Hf // 2) // factor, (Wf // 2) // factor)))


@depth_reproject_lowres.register_kernel("cpu")
def _(transformation, previous_depth, full_K, half_K, factor):
    _no_cpu("depth_reproject_lowres")


# ----------------------------------------------------------------------------------------------------------------------
# frame-path epilogues (inference only: no autograd formulas are registered)
# ----------------------------------------------------------------------------------------------------------------------
ACTIVATIONS = {"none": 0, "relu": 1, "sigmoid": 2}


RESIDUAL_NONE, RESIDUAL_SAME, RESIDUAL_NEAREST_UP2 = 0, 1, 2


@torch.library.custom_op("dvmvs::bias_act_", mutates_args=("x",), device_types="cuda")
def bias_act_(x: Tensor, bias: Tensor, activation: int, residual: Tensor, residual_mode: int) -> None:
    """In place: x[b,c] = act(x[b,c] + bias[c]) (+ residual) for a contiguous NCHW tensor.  ``bias`` / ``residual`` may be
    empty (numel 0); residual_mode 1: same shape, 2: half resolution, nearest-up-sampled on the fly."""
    _dev_f32("bias_act_", x)
    if not x.is_contiguous() or x.dim() != 4:
        raise ValueError("dvmvs::bias_act_: expected a contiguous NCHW tensor")
    B, C, H, W = x.shape
    if bias.numel() not in (0, C):
        raise ValueError(f"dvmvs::bias_act_: bias has {bias.numel()} entries for {C} channels")
    res_ptr = None
    if residual_mode != RESIDUAL_NONE:
        want = (B, C, H, W) if residual_mode == RESIDUAL_SAME else (B, C, H // 2, W // 2)
        if tuple(residual.shape) != want:
            raise ValueError(f"dvmvs::bias_act_: residual {tuple(residual.shape)} does not match {want}")
        residual = residual.contiguous()
        res_ptr = _ptr(residual)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_bias_act_inplace(_ptr(x), _ptr(bias.contiguous()) if bias.numel() else None, res_ptr, int(residual_mode),
                                                B, C, H, W, int(activation), _stream(x))
    _capi.check(rc, "dvmvs_bias_act_inplace")


@bias_act_.register_kernel("cpu")
def _(x, bias, activation, residual, residual_mode):
    _no_cpu("bias_act_")


@torch.library.custom_op("dvmvs::upsample2x", mutates_args=(), device_types="cuda")
def upsample2x(x: Tensor) -> Tensor:
    _dev_f32("upsample2x", x)
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_upsample2x_fwd(_ptr(x), _ptr(out), 0, None, 0, B, C, H, W, _stream(x))
    _capi.check(rc, "dvmvs_upsample2x_fwd")
    return out


@upsample2x.register_fake
def _(x):
    B, C, H, W = x.shape
    return x.new_empty((B, C, 2 * H, 2 * W))


@upsample2x.register_kernel("cpu")
def _(x):
    _no_cpu("upsample2x")


@torch.library.custom_op("dvmvs::upsample2x_bwd", mutates_args=(), device_types="cuda")
def upsample2x_bwd(grad_out: Tensor) -> Tensor:
    """Adjoint of ``upsample2x`` as a gather (no atomics: bit-reproducible), dvmvs_upsample2x_bwd."""
    _dev_f32("upsample2x_bwd", grad_out)
    grad_out = grad_out.contiguous()
    B, C, OH, OW = grad_out.shape
    grad_in = torch.empty((B, C, OH // 2, OW // 2), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        rc = _capi.lib().dvmvs_upsample2x_bwd(_ptr(grad_out), _ptr(grad_in), B, C, OH // 2, OW // 2, _stream(grad_out))
    _capi.check(rc, "dvmvs_upsample2x_bwd")
    return grad_in


@upsample2x_bwd.register_fake
def _(grad_out):
    B, C, OH, OW = grad_out.shape
    return grad_out.new_empty((B, C, OH // 2, OW // 2))


torch.library.register_autograd("dvmvs::upsample2x", lambda ctx, grad: upsample2x_bwd(grad), setup_context=lambda ctx, inputs, output: None)


@torch.library.custom_op("dvmvs::depthwise_conv_train", mutates_args=(), device_types="cuda")
def depthwise_conv_train(x: Tensor, weight: Tensor, stride: int) -> Tensor:
    """Depthwise k x k convolution (weight [C,1,k,k], padding k//2, no bias) with gradients: the training-time form of
    ``depthwise_conv`` (MIOpen runs these MnasNet layers, forward and backward, through its naive reference kernels)."""
    _dev_f32("depthwise_conv_train", x, weight)
    x, weight = x.contiguous(), weight.contiguous()
    B, C, H, W = x.shape
    k = weight.shape[-1]
    if tuple(weight.shape) != (C, 1, k, k):
        raise ValueError(f"dvmvs::depthwise_conv_train: weight {tuple(weight.shape)} is not depthwise for {C} channels")
    pad = k // 2
    out = torch.empty((B, C, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_depthwise_conv_fwd(_ptr(x), _ptr(weight), None, None, 0, _ptr(out), B, C, H, W, k, int(stride), 0, _stream(x))
    _capi.check(rc, "dvmvs_depthwise_conv_fwd")
    return out


@depthwise_conv_train.register_fake
def _(x, weight, stride):
    B, C, H, W = x.shape
    k = weight.shape[-1]
    pad = k // 2
    return x.new_empty((B, C, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1))


@torch.library.custom_op("dvmvs::depthwise_conv_bwd", mutates_args=(), device_types="cuda")
def depthwise_conv_bwd(grad_out: Tensor, x: Tensor, weight: Tensor, stride: int, need_input: bool, need_weight: bool) -> Tuple[Tensor, Tensor]:
    """(grad_x, grad_weight) of ``depthwise_conv_train`` (gathers / a fixed-order reduction: no atomics); an unneeded one is empty."""
    _dev_f32("depthwise_conv_bwd", grad_out, x, weight)
    grad_out, x, weight = grad_out.contiguous(), x.contiguous(), weight.contiguous()
    B, C, H, W = x.shape
    k = weight.shape[-1]
    gx = torch.empty_like(x) if need_input else x.new_empty(0)
    gw = torch.empty_like(weight) if need_weight else x.new_empty(0)
    lib = _capi.lib()
    scratch = lib.dvmvs_depthwise_conv_bwd_workspace_bytes(B, C, H, W, k, int(stride)) if need_weight else 0
    workspace = torch.empty(scratch // 4, dtype=torch.float32, device=x.device) if scratch else None
    with torch.cuda.device(x.device):
        rc = lib.dvmvs_depthwise_conv_bwd(_ptr(grad_out), _ptr(x), _ptr(weight), _ptr(gx) if need_input else None,
                                          _ptr(gw) if need_weight else None, _ptr(workspace) if workspace is not None else None,
                                          B, C, H, W, k, int(stride), _stream(x))
    _capi.check(rc, "dvmvs_depthwise_conv_bwd")
    return gx, gw


@depthwise_conv_bwd.register_fake
def _(grad_out, x, weight, stride, need_input, need_weight):
    return (torch.empty_like(x) if need_input else x.new_empty(0)), (torch.empty_like(weight) if need_weight else x.new_empty(0))


def _depthwise_train_setup(ctx, inputs, output):
    x, weight, stride = inputs
    ctx.stride = stride
    ctx.save_for_backward(x, weight)


def _depthwise_train_backward(ctx, grad):
    x, weight = ctx.saved_tensors
    need_input, need_weight = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    gx, gw = depthwise_conv_bwd(grad, x, weight, ctx.stride, need_input, need_weight)
    return (gx if need_input else None), (gw if need_weight else None), None


torch.library.register_autograd("dvmvs::depthwise_conv_train", _depthwise_train_backward, setup_context=_depthwise_train_setup)


@torch.library.custom_op("dvmvs::depthwise_conv", mutates_args=(), device_types="cuda")
def depthwise_conv(x: Tensor, weight: Tensor, bias: Tensor, stride: int, activation: int, pre_bias: Tensor, pre_relu: bool) -> Tensor:
    """Depthwise k x k convolution (weight [C,1,k,k], padding k//2) + bias (numel 0 = none) + activation, one HIP launch.  With
    ``pre_relu`` the input is the raw output of the preceding 1x1 convolution and relu(x + pre_bias[c]) (numel 0 = no bias) is applied
    to it on the fly."""
    _dev_f32("depthwise_conv", x, weight)
    x, weight = x.contiguous(), weight.contiguous()
    B, C, H, W = x.shape
    k = weight.shape[-1]
    if tuple(weight.shape) != (C, 1, k, k):
        raise ValueError(f"dvmvs::depthwise_conv: weight {tuple(weight.shape)} is not depthwise for {C} channels")
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if pre_bias.numel() not in (0, C):
            raise ValueError(f"dvmvs::depthwise_conv: pre_bias has {pre_bias.numel()} entries for {C} channels")
        rc = _capi.lib().dvmvs_depthwise_conv_fwd(_ptr(x), _ptr(weight), _ptr(bias.contiguous()) if bias.numel() else None,
                                                  _ptr(pre_bias.contiguous()) if pre_bias.numel() else None, int(bool(pre_relu)), _ptr(out),
                                                  B, C, H, W, k, int(stride), int(activation), _stream(x))
    _capi.check(rc, "dvmvs_depthwise_conv_fwd")
    return out


@depthwise_conv.register_fake
def _(x, weight, bias, stride, activation, pre_bias, pre_relu):
    B, C, H, W = x.shape
    k = weight.shape[-1]
    pad = k // 2
    return x.new_empty((B, C, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1))


@depthwise_conv.register_kernel("cpu")
def _(x, weight, bias, stride, activation, pre_bias, pre_relu):
    _no_cpu("depthwise_conv")


# ----------------------------------------------------------------------------------------------------------------------
# destination-passing forms for the frame engine (inference, no autograd, hipGraph-capturable): the producer writes straight
# into a channel slice of a concatenation buffer or into a state buffer, so that torch.cat / copy_ launches disappear from
# the frame.  Plain functions over the C ABI, not torch.library ops: nothing here allocates or needs shape inference.
# ----------------------------------------------------------------------------------------------------------------------
ACTIVATION_SIGMOID_TO_DEPTH = 3


def _slice_batch_stride(name, dst, B, C, H, W):
    """``dst`` must be [B,C,H,W] with dense planes and channels (a channel slice of a contiguous NCHW buffer); returns its batch
    stride in elements."""
    if dst.device.type != "cuda" or dst.dtype != torch.float32:
        raise TypeError(f"dvmvs::{name}: destination must be a float32 HIP tensor")
    if tuple(dst.shape) != (B, C, H, W) or dst.stride(3) != 1 or dst.stride(2) != W or dst.stride(1) != H * W:
        raise ValueError(f"dvmvs::{name}: destination {tuple(dst.shape)} / strides {dst.stride()} is not a channel slice of a "
                         f"contiguous NCHW buffer for [{B},{C},{H},{W}]")
    return dst.stride(0) if B > 1 else C * H * W


def bias_act_into(x: Tensor, dst: Tensor, bias, activation: int, residual=None, residual_mode: int = RESIDUAL_NONE, p0: float = 0.0,
                  p1: float = 0.0) -> Tensor:
    """dst = act(x + bias[c]) (+ residual); ``x`` a dense convolution output, ``dst`` x itself or a channel slice (see above)."""
    _dev_f32("bias_act_into", x)
    if not x.is_contiguous() or x.dim() != 4:
        raise ValueError("dvmvs::bias_act_into: expected a contiguous NCHW source")
    B, C, H, W = x.shape
    stride = _slice_batch_stride("bias_act_into", dst, B, C, H, W)
    if bias is not None and bias.numel() not in (0, C):
        raise ValueError(f"dvmvs::bias_act_into: bias has {bias.numel()} entries for {C} channels")
    res_ptr = None
    if residual_mode != RESIDUAL_NONE:
        want = (B, C, H, W) if residual_mode == RESIDUAL_SAME else (B, C, H // 2, W // 2)
        if tuple(residual.shape) != want or not residual.is_contiguous():
            raise ValueError(f"dvmvs::bias_act_into: residual {tuple(residual.shape)} does not match {want} (contiguous)")
        res_ptr = _ptr(residual)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_bias_act_fwd(_ptr(x), _ptr(dst), stride, _ptr(bias) if bias is not None and bias.numel() else None, res_ptr,
                                            int(residual_mode), B, C, H, W, int(activation), float(p0), float(p1), _stream(x))
    _capi.check(rc, "dvmvs_bias_act_fwd")
    return dst


EUNSUPPORTED = -2     # include/dvmvs_hip.h


def conv_bias_act_into(x: Tensor, weight: Tensor, bias: Tensor, dst, stride: int, padding: int, activation: int):
    """conv2d(x, weight, padding, stride) + bias [+ ReLU] as ONE MIOpen fusion plan (dvmvs_conv_bias_act_fwd), written into ``dst``
    (a dense tensor or a channel slice of a concatenation buffer; None: a new tensor).  Returns the output, or None when MIOpen has
    no fused plan for the problem (the caller then runs convolution + dvmvs_bias_act_fwd)."""
    _dev_f32("conv_bias_act_into", x, weight, bias)
    if activation not in (ACTIVATIONS["none"], ACTIVATIONS["relu"]):
        raise ValueError("dvmvs::conv_bias_act_into: activation must be none or relu")
    x, weight = x.contiguous(), weight.contiguous()
    B, Cin, H, W = x.shape
    Cout, Cin_w, K, K2 = weight.shape
    if Cin_w != Cin or K != K2 or bias.numel() != Cout:
        raise ValueError(f"dvmvs::conv_bias_act_into: weight {tuple(weight.shape)} / bias {bias.numel()} do not fit input {tuple(x.shape)}")
    Ho, Wo = (H + 2 * padding - K) // stride + 1, (W + 2 * padding - K) // stride + 1
    if dst is None:
        dst = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    batch_stride = _slice_batch_stride("conv_bias_act_into", dst, B, Cout, Ho, Wo)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_conv_bias_act_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(dst), batch_stride, B, Cin, H, W, Cout, K,
                                                 int(stride), int(padding), int(activation), _stream(x))
    if rc == EUNSUPPORTED:
        return None
    _capi.check(rc, "dvmvs_conv_bias_act_fwd")
    return dst


def upsample2x_into(x: Tensor, dst: Tensor, pre_bias=None, pre_activation: int = 0) -> Tensor:
    """x2 bilinear (align_corners) up-sampling of ``x`` into ``dst``; with ``pre_activation`` (ACTIVATIONS) ``x`` is a raw convolution
    output and act(x + pre_bias[c]) is applied to the taps on the fly."""
    _dev_f32("upsample2x_into", x)
    x = x.contiguous()
    B, C, H, W = x.shape
    stride = _slice_batch_stride("upsample2x_into", dst, B, C, 2 * H, 2 * W)
    if pre_bias is not None and pre_bias.numel() not in (0, C):
        raise ValueError(f"dvmvs::upsample2x_into: pre_bias has {pre_bias.numel()} entries for {C} channels")
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_upsample2x_fwd(_ptr(x), _ptr(dst), stride, _ptr(pre_bias) if pre_bias is not None and pre_bias.numel() else None,
                                              int(pre_activation), B, C, H, W, _stream(x))
    _capi.check(rc, "dvmvs_upsample2x_fwd")
    return dst


def upsample2x_pair_into(x1: Tensor, dst1: Tensor, x2: Tensor, dst2: Tensor, pre_bias2=None, pre_activation2: int = 0) -> None:
    """``upsample2x_into(x1, dst1)`` and ``upsample2x_into(x2, dst2, pre_bias2, pre_activation2)`` for two maps of the same size in ONE launch
    (dvmvs_upsample2x_pair_fwd: a decoder level's feature map and its depth head); the same values bit for bit."""
    _dev_f32("upsample2x_pair_into", x1, x2)
    x1, x2 = x1.contiguous(), x2.contiguous()
    B, C1, H, W = x1.shape
    C2 = x2.shape[1]
    if tuple(x2.shape) != (B, C2, H, W):
        raise ValueError(f"dvmvs::upsample2x_pair_into: the two maps must have the same batch and size, got {tuple(x1.shape)} and {tuple(x2.shape)}")
    stride1 = _slice_batch_stride("upsample2x_pair_into", dst1, B, C1, 2 * H, 2 * W)
    stride2 = _slice_batch_stride("upsample2x_pair_into", dst2, B, C2, 2 * H, 2 * W)
    if pre_bias2 is not None and pre_bias2.numel() not in (0, C2):
        raise ValueError(f"dvmvs::upsample2x_pair_into: pre_bias2 has {pre_bias2.numel()} entries for {C2} channels")
    with torch.cuda.device(x1.device):
        rc = _capi.lib().dvmvs_upsample2x_pair_fwd(_ptr(x1), _ptr(dst1), stride1, C1, _ptr(x2), _ptr(dst2), stride2,
                                                   _ptr(pre_bias2) if pre_bias2 is not None and pre_bias2.numel() else None, int(pre_activation2), C2, B, H, W,
                                                   _stream(x1))
    _capi.check(rc, "dvmvs_upsample2x_pair_fwd")


def lstm_gates_into(combined_conv: Tensor, c_state: Tensor, h_state: Tensor) -> None:
    """State update in place: c_state <- c', h_state <- h' (the kernel reads a row of c completely before it writes it)."""
    _dev_f32("lstm_gates_into", combined_conv, c_state, h_state)
    B, hidden, H, W = c_state.shape
    if tuple(combined_conv.shape) != (B, 4 * hidden, H, W) or tuple(h_state.shape) != tuple(c_state.shape):
        raise ValueError("dvmvs::lstm_gates_into: shapes do not match")
    if not (combined_conv.is_contiguous() and c_state.is_contiguous() and h_state.is_contiguous()):
        raise ValueError("dvmvs::lstm_gates_into: expected contiguous tensors")
    with torch.cuda.device(c_state.device):
        rc = _capi.lib().dvmvs_lstm_gates_fwd(_ptr(combined_conv), _ptr(c_state), _ptr(h_state), _ptr(c_state), B, hidden, H, W, _stream(c_state))
    _capi.check(rc, "dvmvs_lstm_gates_fwd")


# ---- 3x3 convolutions on the bottleneck maps (csrc/bottleneck_conv.hip): weight-streaming fp32 MFMA GEMM, deterministic split-K ----
def bottleneck_conv_splits(B: int, C_out: int, C_in: int, H_in: int, W_in: int, stride: int) -> int:
    """Number of K-splits (partial sums) dvmvs_bottleneck_conv_fwd produces for this problem; 0 when the kernel does not take it
    (other map sizes, C_in not a multiple of 16): the caller stays on MIOpen."""
    n = _capi.lib().dvmvs_bottleneck_conv_splits(int(B), int(C_out), int(C_in), int(H_in), int(W_in), int(stride))
    return n if n > 0 else 0


def bottleneck_conv_pack(weight: Tensor) -> Tensor:
    """[C_out, C_in, 3, 3] convolution weights re-packed once into the MFMA A-operand order the kernel streams."""
    _dev_f32("bottleneck_conv_pack", weight)
    C_out, C_in, kh, kw = weight.shape
    nbytes = _capi.lib().dvmvs_bottleneck_conv_packed_bytes(C_out, C_in) if (kh, kw) == (3, 3) else 0
    if nbytes == 0:
        raise ValueError(f"dvmvs::bottleneck_conv_pack: need a 3x3 kernel and C_in % 16 == 0, got {tuple(weight.shape)}")
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _capi.lib().dvmvs_bottleneck_conv_pack(_ptr(weight.contiguous()), _ptr(packed), C_out, C_in, _stream(weight))
    _capi.check(rc, "dvmvs_bottleneck_conv_pack")
    return packed


def bottleneck_conv_into(x: Tensor, packed: Tensor, C_out: int, stride: int, partials: Tensor, upsample: bool = False) -> int:
    """3x3, padding 1 convolution of ``x`` [B,C_in,H,W] with ``bottleneck_conv_pack``-ed weights as K-split partial sums into
    ``partials`` (at least splits * B * C_out * H_out * W_out floats, laid out [split][B][C_out][H_out*W_out]).  Returns the number of
    splits; the consumer adds them in ascending order (``partial_sums_bias_act_into`` / ``lstm_gates_partials_into``).
    ``upsample``: the layer convolves the 2x bilinear up-sampling (align_corners=True, ``upsample2x``'s values bit for bit) of ``x``
    [B,C_in,H/2,W/2], interpolated while the input is staged (dvmvs_bottleneck_conv_up2x_fwd: the 16x20 map, stride 1)."""
    _dev_f32("bottleneck_conv_into", x, packed, partials)
    B, C_in, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    splits = bottleneck_conv_splits(B, C_out, C_in, H, W, stride) if not upsample or (H, W, stride) == (16, 20, 1) else 0
    if splits == 0:
        raise ValueError(f"dvmvs::bottleneck_conv_into: problem {tuple(x.shape)} -> {C_out} (stride {stride}{', up-sampled' if upsample else ''}) "
                         f"is not one of the bottleneck shapes")
    if not x.is_contiguous() or partials.numel() < splits * B * C_out * (H // stride) * (W // stride) or not partials.is_contiguous():
        raise ValueError("dvmvs::bottleneck_conv_into: expected a contiguous input and a partial-sum buffer of splits * B * C_out * H_out * W_out floats")
    with torch.cuda.device(x.device):
        if upsample:
            rc = _capi.lib().dvmvs_bottleneck_conv_up2x_fwd(_ptr(x), _ptr(packed), _ptr(partials), B, C_in, H, W, int(C_out), _stream(x))
        else:
            rc = _capi.lib().dvmvs_bottleneck_conv_fwd(_ptr(x), _ptr(packed), _ptr(partials), B, C_in, H, W, int(C_out), int(stride), _stream(x))
    _capi.check(rc, "dvmvs_bottleneck_conv_fwd")
    return splits


def partial_sums_bias_act_into(partials: Tensor, n_partials: int, dst: Tensor, bias, activation: int, shape) -> Tensor:
    """dst = act(sum of the partial sums + bias[c]); ``dst`` a dense [B,C,H,W] tensor or a channel slice of a concatenation buffer."""
    _dev_f32("partial_sums_bias_act_into", partials, dst)
    B, C, H, W = shape
    stride = _slice_batch_stride("partial_sums_bias_act_into", dst, B, C, H, W)
    with torch.cuda.device(dst.device):
        rc = _capi.lib().dvmvs_partial_sums_bias_act_fwd(_ptr(partials), int(n_partials), _ptr(dst), stride,
                                                         _ptr(bias) if bias is not None and bias.numel() else None, B, C, H * W, int(activation),
                                                         _stream(dst))
    _capi.check(rc, "dvmvs_partial_sums_bias_act_fwd")
    return dst


# ---- dense 3x3 / 5x5 convolutions of the larger maps (csrc/direct_conv.hip): direct fp32 MFMA convolution, bias + ReLU fused ----
def direct_conv_tile(B: int, C_in: int, H: int, W: int, C_out: int, kernel_size: int, stride: int) -> int:
    """0 when dvmvs_direct_conv_fwd does not take the problem (the caller keeps its library convolution); else the number of
    16-channel output tiles per wave (1 or 2) to pack the weights for."""
    return _capi.lib().dvmvs_direct_conv_tile(int(B), int(C_in), int(H), int(W), int(C_out), int(kernel_size), int(stride))


def direct_conv_pack(weight: Tensor, n_tile: int) -> Tensor:
    """[C_out, C_in, k, k] convolution weights re-packed once into the MFMA B-operand order the kernel streams."""
    _dev_f32("direct_conv_pack", weight)
    C_out, C_in, kh, kw = weight.shape
    nbytes = _capi.lib().dvmvs_direct_conv_packed_bytes(C_out, C_in, kh, int(n_tile)) if kh == kw else 0
    if nbytes == 0:
        raise ValueError(f"dvmvs::direct_conv_pack: need a 3x3 or 5x5 kernel and C_out % (16 * n_tile) == 0, got {tuple(weight.shape)}, n_tile {n_tile}")
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _capi.lib().dvmvs_direct_conv_pack(_ptr(weight.contiguous()), _ptr(packed), C_out, C_in, kh, int(n_tile), _stream(weight))
    _capi.check(rc, "dvmvs_direct_conv_pack")
    return packed


def direct_conv_into(x: Tensor, packed: Tensor, n_tile: int, bias, dst: Tensor, C_out: int, kernel_size: int, stride: int, activation: int,
                     dst_nhwc=None) -> Tensor:
    """dst = act(conv2d(x, W, padding = k // 2, stride) + bias) with ``direct_conv_pack``-ed weights; ``dst`` a dense [B,C_out,H/s,W/s]
    tensor or a channel slice of a concatenation buffer; activation "none" or "relu".  ``dst_nhwc``: a second destination that receives the
    same values channels-last ([B,C_out,H/s,W/s] in torch.channels_last memory format), written in the same epilogue."""
    _dev_f32("direct_conv_into", x, packed, dst)
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("dvmvs::direct_conv_into: expected a contiguous NCHW input")
    B, C_in, H, W = x.shape
    batch_stride = _slice_batch_stride("direct_conv_into", dst, B, int(C_out), H // stride, W // stride)
    if bias is not None and bias.numel() not in (0, C_out):
        raise ValueError(f"dvmvs::direct_conv_into: bias has {bias.numel()} entries for {C_out} channels")
    if dst_nhwc is not None:
        _dev_f32("direct_conv_into", dst_nhwc)
        if tuple(dst_nhwc.shape) != (B, int(C_out), H // stride, W // stride) or not dst_nhwc.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("dvmvs::direct_conv_into: dst_nhwc must be a dense channels-last tensor of the output's shape")
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_direct_conv_dual_fwd(_ptr(x), 0, _ptr(packed), int(n_tile), _ptr(bias) if bias is not None and bias.numel() else None,
                                                    _ptr(dst), batch_stride, _ptr(dst_nhwc) if dst_nhwc is not None else None, B, C_in, H, W, int(C_out),
                                                    int(kernel_size), int(stride), int(activation), _stream(x))
    _capi.check(rc, "dvmvs_direct_conv_dual_fwd")
    return dst


# ---- 1x1 convolutions (csrc/pointwise_conv.hip): fp32 MFMA GEMM from the NCHW map, bias + ReLU + residual in its store path ----
def pointwise_conv_supported(B: int, C_in: int, H: int, W: int, C_out: int, activation: int, residual_mode: int) -> bool:
    """Whether dvmvs_pointwise_conv_fwd takes the problem (else the caller keeps its library convolution + epilogue launch)."""
    return bool(_capi.lib().dvmvs_pointwise_conv_supported(int(B), int(C_in), int(H), int(W), int(C_out), int(activation), int(residual_mode)))


def pointwise_conv_pack(weight: Tensor) -> Tensor:
    """[C_out,C_in,1,1] (or [C_out,C_in]) -> the kernel's B-operand order; once per layer (the weights are constants at inference)."""
    _dev_f32("pointwise_conv_pack", weight)
    if weight.dim() not in (2, 4) or (weight.dim() == 4 and tuple(weight.shape[2:]) != (1, 1)):
        raise ValueError(f"dvmvs::pointwise_conv_pack: need a [C_out,C_in,1,1] weight, got {tuple(weight.shape)}")
    C_out, C_in = int(weight.shape[0]), int(weight.shape[1])
    nbytes = _capi.lib().dvmvs_pointwise_conv_packed_bytes(C_out, C_in)
    packed = torch.empty(nbytes // 4, device=weight.device, dtype=torch.float32)
    with torch.cuda.device(weight.device):
        rc = _capi.lib().dvmvs_pointwise_conv_pack(_ptr(weight.contiguous()), _ptr(packed), C_out, C_in, _stream(weight))
    _capi.check(rc, "dvmvs_pointwise_conv_pack")
    return packed


def pointwise_conv_into(x: Tensor, packed: Tensor, bias, dst: Tensor, C_out: int, activation: int, residual=None, residual_mode: int = 0,
                        splits: int = 0) -> Tensor:
    """dst = act(conv1x1(x, W) + bias) + residual with ``pointwise_conv_pack``-ed weights; ``dst`` a dense [B,C_out,H,W] tensor or a channel
    slice of a concatenation buffer; activation "none" or "relu"; ``residual`` / ``residual_mode`` as ``bias_act_into`` (1: same shape,
    2: half resolution, nearest-up-sampled)."""
    _dev_f32("pointwise_conv_into", x, packed, dst)
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("dvmvs::pointwise_conv_into: expected a contiguous NCHW input")
    B, C_in, H, W = x.shape
    C_out = int(C_out)
    if packed.numel() * 4 != _capi.lib().dvmvs_pointwise_conv_packed_bytes(C_out, C_in):
        raise ValueError(f"dvmvs::pointwise_conv_into: the packed weights are not those of a {C_in} -> {C_out} layer")
    batch_stride = _slice_batch_stride("pointwise_conv_into", dst, B, C_out, H, W)
    if bias is not None and bias.numel() not in (0, C_out):
        raise ValueError(f"dvmvs::pointwise_conv_into: bias has {bias.numel()} entries for {C_out} channels")
    residual_stride = 0
    if residual_mode:
        _dev_f32("pointwise_conv_into", residual)
        shape = (B, C_out, H, W) if residual_mode == 1 else (B, C_out, H // 2, W // 2)
        if tuple(residual.shape) != shape:
            raise ValueError(f"dvmvs::pointwise_conv_into: residual of shape {tuple(residual.shape)}, mode {residual_mode} expects {shape}")
        residual_stride = _slice_batch_stride("pointwise_conv_into", residual, *shape)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_pointwise_conv_fwd(_ptr(x), 0, _ptr(packed), _ptr(bias) if bias is not None and bias.numel() else None,
                                                  _ptr(residual) if residual_mode else None, residual_stride, int(residual_mode), _ptr(dst), batch_stride,
                                                  B, C_in, H, W, C_out, int(activation), int(splits), _stream(x))
    _capi.check(rc, "dvmvs_pointwise_conv_fwd")
    return dst


def conv_head_into(x: Tensor, weight: Tensor, bias, dst: Tensor, activation: int = 0, p0: float = 0.0, p1: float = 0.0) -> Tensor:
    """3x3, padding 1 convolution with ONE output channel (the decoder's depth heads): dst [B,1,H,W] = act(conv(x, weight) + bias);
    bias None and activation 0: the raw convolution output (the consumer applies bias + activation)."""
    _dev_f32("conv_head_into", x, weight, dst)
    if x.dim() != 4 or not x.is_contiguous() or tuple(weight.shape) != (1, x.shape[1], 3, 3) or not weight.is_contiguous():
        raise ValueError(f"dvmvs::conv_head_into: expected a contiguous NCHW input and a [1,{x.shape[1]},3,3] weight, got {tuple(x.shape)}, {tuple(weight.shape)}")
    B, C_in, H, W = x.shape
    batch_stride = _slice_batch_stride("conv_head_into", dst, B, 1, H, W)
    with torch.cuda.device(x.device):
        rc = _capi.lib().dvmvs_conv_head_fwd(_ptr(x), 0, _ptr(weight), _ptr(bias) if bias is not None and bias.numel() else None, _ptr(dst),
                                             batch_stride, B, C_in, H, W, int(activation), float(p0), float(p1), _stream(x))
    _capi.check(rc, "dvmvs_conv_head_fwd")
    return dst


def lstm_gates_partials_into(conv_partials: Tensor, n_partials: int, c_state: Tensor, h_state: Tensor) -> None:
    """``lstm_gates_into`` on a convolution output that arrives as ``n_partials`` partial sums ([split][B][4*hidden][H*W])."""
    _dev_f32("lstm_gates_partials_into", conv_partials, c_state, h_state)
    B, hidden, H, W = c_state.shape
    if conv_partials.numel() < n_partials * B * 4 * hidden * H * W or tuple(h_state.shape) != tuple(c_state.shape):
        raise ValueError("dvmvs::lstm_gates_partials_into: shapes do not match")
    if not (conv_partials.is_contiguous() and c_state.is_contiguous() and h_state.is_contiguous()):
        raise ValueError("dvmvs::lstm_gates_partials_into: expected contiguous tensors")
    with torch.cuda.device(c_state.device):
        rc = _capi.lib().dvmvs_lstm_gates_partials_fwd(_ptr(conv_partials), int(n_partials), _ptr(c_state), _ptr(h_state), _ptr(c_state), B, hidden, H, W,
                                                       _stream(c_state))
    _capi.check(rc, "dvmvs_lstm_gates_partials_fwd")


def hidden_warp_into(image_src: Tensor, depth_dst: Tensor, src_trans_dst: Tensor, camera_matrix: Tensor, zero_invalid: bool, dst: Tensor) -> Tensor:
    _dev_f32("hidden_warp_into", image_src, depth_dst, src_trans_dst, camera_matrix, dst)
    B, C, H, W = image_src.shape
    if not (image_src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (B, C, H, W)):
        raise ValueError("dvmvs::hidden_warp_into: source and destination must be contiguous [B,C,H,W]")
    with torch.cuda.device(image_src.device):
        rc = _capi.lib().dvmvs_hidden_warp_fwd(_ptr(image_src), _ptr(depth_dst.contiguous()), _ptr(src_trans_dst.contiguous()),
                                               _ptr(camera_matrix.contiguous()), _ptr(dst), B, C, H, W, int(bool(zero_invalid)), _stream(image_src))
    _capi.check(rc, "dvmvs_hidden_warp_fwd")
    return dst


def depth_reproject_lowres_into(transformation: Tensor, previous_depth: Tensor, full_K: Tensor, half_K: Tensor, zbuffer: Tensor, out_lowres: Tensor,
                                factor: int) -> Tensor:
    """Splat + decimate with a caller-owned z-buffer [B,Hf/2,Wf/2] that is all-zero before and after the call (two launches)."""
    _dev_f32("depth_reproject_lowres_into", transformation, previous_depth, full_K, half_K, zbuffer, out_lowres)
    B, one, Hf, Wf = previous_depth.shape
    if one != 1 or zbuffer.numel() != B * (Hf // 2) * (Wf // 2) or not zbuffer.is_contiguous() or not out_lowres.is_contiguous() or \
            out_lowres.numel() != B * ((Hf // 2) // factor) * ((Wf // 2) // factor):
        raise ValueError("dvmvs::depth_reproject_lowres_into: buffer shapes do not match the previous depth")
    with torch.cuda.device(previous_depth.device):
        rc = _capi.lib().dvmvs_depth_reproject_lowres_fwd(_ptr(transformation.contiguous()), _ptr(previous_depth.contiguous()), _ptr(full_K.contiguous()),
                                                          _ptr(half_K.contiguous()), _ptr(zbuffer), _ptr(out_lowres), int(factor), B, Hf, Wf,
                                                          _stream(previous_depth))
    _capi.check(rc, "dvmvs_depth_reproject_lowres_fwd")
    return out_lowres


def depth_reproject_estimate_into(transformation: Tensor, previous_depth: Tensor, full_K: Tensor, half_K: Tensor, estimate: Tensor,
                                  estimate_to_clear: Optional[Tensor], factor: int) -> Tensor:
    """The frame path's re-projection in one launch (dvmvs_depth_reproject_estimate_fwd): splat straight into ``estimate``
    [B,1,H/2/f,W/2/f] (all-zero on entry), zero-filling ``estimate_to_clear`` (the buffer of the frame after next) on the way."""
    _dev_f32("depth_reproject_estimate_into", transformation, previous_depth, full_K, half_K, estimate)
    B, one, Hf, Wf = previous_depth.shape
    shape = (B, 1, Hf // 2 // factor, Wf // 2 // factor)
    if one != 1 or tuple(estimate.shape) != shape or not estimate.is_contiguous() or not previous_depth.is_contiguous() or \
            (estimate_to_clear is not None and (tuple(estimate_to_clear.shape) != shape or not estimate_to_clear.is_contiguous() or
                                                estimate_to_clear.data_ptr() == estimate.data_ptr())):
        raise ValueError("dvmvs::depth_reproject_estimate_into: expected contiguous [B,1,H,W] depth and distinct [B,1,H/2/f,W/2/f] estimate buffers")
    with torch.cuda.device(previous_depth.device):
        rc = _capi.lib().dvmvs_depth_reproject_estimate_fwd(_ptr(transformation.contiguous()), _ptr(previous_depth), _ptr(full_K.contiguous()),
                                                            _ptr(half_K.contiguous()), _ptr(estimate),
                                                            _ptr(estimate_to_clear) if estimate_to_clear is not None else None, int(factor), B, Hf, Wf,
                                                            _stream(previous_depth))
    _capi.check(rc, "dvmvs_depth_reproject_estimate_fwd")
    return estimate


def copy_batch(pairs) -> None:
    """``dst.copy_(src)`` for up to eight (dst, src) pairs of equally laid out dense float32 device tensors in ONE launch (dvmvs_copy_batch).
    The caller guarantees what the C entry requires: same shape and strides per pair, dense storage, 16-byte aligned, a multiple of 4
    elements, no overlap between pairs (``batchable`` checks one pair)."""
    n = len(pairs)
    srcs, dsts, counts = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_longlong * n)()
    for j, (dst, src) in enumerate(pairs):
        srcs[j], dsts[j], counts[j] = src.data_ptr(), dst.data_ptr(), dst.numel()
    device = pairs[0][0].device
    with torch.cuda.device(device):
        rc = _capi.lib().dvmvs_copy_batch(srcs, dsts, counts, n, _stream(pairs[0][0]))
    _capi.check(rc, "dvmvs_copy_batch")


def batchable(dst: Tensor, src: Tensor) -> bool:
    """Whether ``dst.copy_(src)`` is a flat copy dvmvs_copy_batch can take."""
    return (dst.is_cuda and src.is_cuda and dst.device == src.device and dst.dtype == torch.float32 and src.dtype == torch.float32
            and dst.shape == src.shape and dst.numel() % 4 == 0 and dst.numel() > 0 and dst.data_ptr() % 16 == 0 and src.data_ptr() % 16 == 0
            and ((dst.is_contiguous() and src.is_contiguous())
                 or (dst.dim() == 4 and dst.is_contiguous(memory_format=torch.channels_last) and src.is_contiguous(memory_format=torch.channels_last))))


def nchw_to_nhwc_into(src: Tensor, dst: Tensor) -> Tensor:
    """``dst`` (a channels-last [B,C,H,W] tensor) = ``src`` (contiguous NCHW), one HIP launch (dvmvs_nchw_to_nhwc: C <= 64, a multiple of 4)."""
    _dev_f32("nchw_to_nhwc_into", src, dst)
    B, C, H, W = src.shape
    if tuple(dst.shape) != (B, C, H, W) or not src.is_contiguous() or not dst.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("dvmvs::nchw_to_nhwc_into: expected a contiguous NCHW source and a channels-last destination of the same shape")
    with torch.cuda.device(src.device):
        rc = _capi.lib().dvmvs_nchw_to_nhwc(_ptr(src), _ptr(dst), B, C, H, W, _stream(src))
    _capi.check(rc, "dvmvs_nchw_to_nhwc")
    return dst


def cost_volume_into(image1: Tensor, image2s, Hm: Tensor, kt: Tensor, min_depth: float, max_depth: float, dst: Tensor, variant: int = 0,
                     work_list: Optional[Tensor] = None) -> Tensor:
    """Dot-product cost volume written into ``dst`` [B,D,H,W] (contiguous: for B == 1 a channel slice of a larger buffer is)."""
    _dev_f32("cost_volume_into", image1, Hm, kt, dst, *image2s)
    B, C, H, W = image1.shape
    M, D = len(image2s), dst.shape[1]
    _check_sweep_matrices("cost_volume_into", Hm, kt, B, M)
    nhwc = C > 1 and all(t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous() for t in image2s)
    if not (image1.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (B, D, H, W) and (nhwc or all(t.is_contiguous() for t in image2s))):
        raise ValueError("dvmvs::cost_volume_into: expected contiguous NCHW tensors (measurement maps: all NCHW or all channels-last)")
    workspace, ws_bytes = sweep_workspace(image1.device, B, M, H, W, D) if COST_VOLUME_TWO_PASS else (None, 0)
    items = _work_list_ptr(work_list, image1, B, H, W, D)
    with torch.cuda.device(image1.device):
        rc = _capi.lib().dvmvs_cost_volume_planned_fwd(_ptr(image1), _capi.pointer_array([_ptr(t) for t in image2s]), _ptr(Hm.contiguous()),
                                                       _ptr(kt.contiguous()), _ptr(dst), B, M, C, H, W, D, float(min_depth), float(max_depth), 1, int(variant),
                                                       _capi.LAYOUT_NHWC if nhwc else _capi.LAYOUT_NCHW, _ptr(workspace) if workspace is not None else None, ws_bytes, items, _stream(image1))
    if rc != 0 and workspace is not None:
        drop_sweep_workspace(image1.device, B, M, H, W, D)
    _capi.check(rc, "dvmvs_cost_volume_planned_fwd")
    return dst
