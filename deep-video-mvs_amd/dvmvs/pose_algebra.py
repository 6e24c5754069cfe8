"""The small pose algebra in front of the hot path: three tiny matrices per frame, and WHERE they are evaluated.

The reference derives, with fp32 torch ops inside its hot functions,

* the sweep constants  ``E = inverse(pose2) @ pose1``, ``Hm = K R K^-1``, ``kt = K t``   (/root/reference/dvmvs/utils.py:51-56)
* the splat transform  ``inverse(reference_pose) @ measurement_pose``                   (utils.py:121)
* the warp transform   ``inverse(previous_pose) @ current_pose``                        (dvmvs/convlstm.py:30)

Camera-to-world poses carry translations of metres, so the fp32 inverse-times-pose has ~5e-7 m of cancellation error in the
relative translation -- up to 3e-4 px on the 0.25 m plane, far more than every other rounding on the path, and the depth
networks amplify it.  "Depth identical to the reference" therefore needs the reference's very matrices, not better ones.
The HIP kernels take the matrices as device arrays (include/dvmvs_hip.h, ABI 3) and this module produces them:

``reference`` (default)
    the reference's own expressions, fp32, on the HOST (LAPACK ``getrf``/``getrs`` behind ``torch.inverse``, ATen's small
    ``bmm``): bit-identical to what the reference computes when it runs on CPU tensors, which is the run the golden
    fixtures under tests/golden/ were captured from.  Poses that already live on the host (they come from ``poses.txt`` /
    a tracker) cost no synchronisation; device tensors are copied back first -- the reference's own ``torch.inverse`` on a
    device tensor synchronises as well (its LU error check).  A few tens of microseconds of host time per frame.
    (On a GPU the reference would use the device LU and ``Kt * (1 / depth)``: another, equally arbitrary fp32 rounding that
    cannot be pinned here.  The CPU rounding is the one that can, so it is the one reproduced.)
``exact``
    fp64 on the device, rounded once (``dvmvs_sweep_matrices`` / ``dvmvs_relative_pose``): no host involvement, closer to
    the real-number result than either fp32 evaluation.  Opt-in: ``DVMVS_POSE_ALGEBRA=exact`` or ``mode="exact"``.

``auto``  (round 6)
    per call: ``exact`` when every pose / intrinsics tensor of the call already lives on the device -- how the reference's own scripts hand them
    over (fusionnet/run-testing.py:127-149 moves poses and K to the device before the loop) -- and ``reference`` for host tensors.  No call
    synchronises the device then; device-resident callers get the real-number matrices rounded once instead of the CPU's fp32 LAPACK rounding
    (the reference on a GPU would get a third, equally arbitrary one).  Measured depth rel-L1 of this mode against the reference fixtures on the 17
    golden frames: bench.py ``module_surface.rel_l1`` / DESIGN.md section 5.  Opt-in: ``DVMVS_POSE_ALGEBRA=auto`` or ``mode="auto"``.

This is host-side set-up of a few 4x4 products, not a fallback of the hot path: every per-pixel operation stays in the
HIP kernels, which raise if the library is missing.
"""
import os
import warnings
from typing import List, Sequence, Tuple

import torch

MODES = ("reference", "exact", "auto")
MODE = os.environ.get("DVMVS_POSE_ALGEBRA", "reference")
if MODE not in MODES:
    raise ValueError(f"DVMVS_POSE_ALGEBRA must be one of {MODES}, got {MODE!r}")


def _mode(mode):
    mode = MODE if mode is None else mode
    if mode not in MODES:
        raise ValueError(f"pose algebra mode must be one of {MODES}, got {mode!r}")
    return mode


def _resolve(mode, tensors):
    """"auto" -> "exact" when all of the call's small tensors are on the device, else "reference"; other modes as they are."""
    mode = _mode(mode)
    if mode != "auto":
        return mode
    return "exact" if all(t.device.type != "cpu" for t in tensors) else "reference"


_WARNED_DEVICE_POSES = False


def to_host(t: torch.Tensor) -> torch.Tensor:
    """``t`` as a host tensor.  A device tensor is copied back, which synchronises the device: poses and intrinsics come from
    the host (``poses.txt``, a tracker) and should be handed over as host tensors; the first device tensor seen in "reference"
    mode is reported once (``DVMVS_POSE_ALGEBRA=exact`` keeps everything on the device instead)."""
    global _WARNED_DEVICE_POSES
    if t.device.type == "cpu":
        return t
    if not _WARNED_DEVICE_POSES:
        _WARNED_DEVICE_POSES = True
        warnings.warn("dvmvs.pose_algebra (mode 'reference'): a pose / intrinsics tensor lives on the device and is copied back to "
                      "the host for the reference's fp32 pose algebra -- a device synchronisation per call; pass host tensors, or "
                      "use DVMVS_POSE_ALGEBRA=exact", RuntimeWarning, stacklevel=3)
    return t.cpu()


_host = to_host      # (older name, kept for callers inside the package)


# ----------------------------------------------------------------------------------------------------------------------
# reference rounding: fp32 torch ops on host tensors
# ----------------------------------------------------------------------------------------------------------------------
def relative_pose_host(a: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """inverse(a) @ c for [B,4,4] host tensors, in their own dtype (fp32 in practice): utils.py:121, convlstm.py:30."""
    return torch.bmm(torch.inverse(a), c)


def sweep_matrices_host(pose1: torch.Tensor, pose2s: Sequence[torch.Tensor], K: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Hm [B,M,9], kt [B,M,3]) on the host, each measurement frame through the reference's expressions (utils.py:51-56).

    The M frames are evaluated one by one, exactly as the reference's loop over measurement frames does (utils.py:93-105):
    LAPACK and ATen's small-matrix bmm work matrix by matrix, so batching would not change a bit, but nothing is gained
    by relying on that."""
    B = pose1.shape[0]
    K_inverse = torch.inverse(K)
    Hm, kt = [], []
    for pose2 in pose2s:
        extrinsic = relative_pose_host(pose2, pose1)                   # measurement camera <- reference camera
        rotation, translation = extrinsic[:, 0:3, 0:3], extrinsic[:, 0:3, 3].unsqueeze(-1)
        kt.append(K.bmm(translation).reshape(B, 3))
        Hm.append(K.bmm(rotation).bmm(K_inverse).reshape(B, 9))
    return torch.stack(Hm, dim=1).contiguous(), torch.stack(kt, dim=1).contiguous()


def sweep_variant_host(Hm: torch.Tensor, kt: torch.Tensor, height: int, width: int, n_depth_levels: int, min_depth: float, max_depth: float) -> int:
    """Which configuration of the LDS-tiled sweep suits this keyframe geometry: 2 (default: 48 KB sample boxes, three 256-thread
    workgroups per CU) or 3 (wide-baseline: 72 KB boxes, 512-thread workgroups).  Decided on the HOST from the host copies of the
    matrices (``Hm`` [B,M,9], ``kt`` [B,M,3], as ``sweep_matrices_host`` returns them) by the library's model of the kernel's run
    plan (``dvmvs_sweep_select_variant``, include/dvmvs_hip.h: a few hundred flops per workgroup, ~20 us) -- no device round trip.
    Either configuration is bit-reproducible; the choice is a deterministic function of the matrices."""
    from dvmvs.hip import _capi
    Hm, kt = Hm.contiguous(), kt.contiguous()
    if Hm.device.type != "cpu" or Hm.dtype != torch.float32 or kt.dtype != torch.float32:
        raise ValueError("sweep_variant_host needs the float32 HOST copies of the sweep matrices")
    variant = _capi.lib().dvmvs_sweep_select_variant(Hm.data_ptr(), kt.data_ptr(), Hm.shape[0], Hm.shape[1], int(height), int(width),
                                                     int(n_depth_levels), float(min_depth), float(max_depth))
    if variant < 0:
        _capi.check(variant, "dvmvs_sweep_select_variant")
    return variant


# ----------------------------------------------------------------------------------------------------------------------
# public: matrices on the device the kernels run on
# ----------------------------------------------------------------------------------------------------------------------
def sweep_matrices(pose1, pose2s, K, device, mode=None, with_host=False):
    """Sweep constants of ``cost_volume_fusion`` as device tensors (Hm [B,M,9], kt [B,M,3]) on ``device``; with ``with_host`` also
    their host copies (None in "exact" mode, where they never exist on the host) for ``sweep_variant_host``."""
    pose2s = list(pose2s)
    if _resolve(mode, [pose1, K] + pose2s) == "reference":
        Hm, kt = sweep_matrices_host(_host(pose1), [_host(p) for p in pose2s], _host(K))
        return (Hm.to(device), kt.to(device), (Hm, kt)) if with_host else (Hm.to(device), kt.to(device))
    if with_host:
        from dvmvs.hip import ops
        return ops.sweep_matrices(pose1.to(device), [p.to(device) for p in pose2s], K.to(device)) + (None,)
    from dvmvs.hip import ops
    return ops.sweep_matrices(pose1.to(device), [p.to(device) for p in pose2s], K.to(device))


def relative_pose(a, c, device, mode=None) -> torch.Tensor:
    """inverse(a) @ c as a [B,4,4] device tensor on ``device``."""
    if _resolve(mode, [a, c]) == "reference":
        return relative_pose_host(_host(a), _host(c)).to(device)
    from dvmvs.hip import ops
    return ops.relative_pose(a.to(device), c.to(device))
