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

This is host-side set-up of a few 4x4 products, not a fallback of the hot path: every per-pixel operation stays in the
HIP kernels, which raise if the library is missing.
"""
import os
from typing import List, Sequence, Tuple

import torch

MODES = ("reference", "exact")
MODE = os.environ.get("DVMVS_POSE_ALGEBRA", "reference")
if MODE not in MODES:
    raise ValueError(f"DVMVS_POSE_ALGEBRA must be one of {MODES}, got {MODE!r}")


def _mode(mode):
    mode = MODE if mode is None else mode
    if mode not in MODES:
        raise ValueError(f"pose algebra mode must be one of {MODES}, got {mode!r}")
    return mode


def _host(t: torch.Tensor) -> torch.Tensor:
    return t if t.device.type == "cpu" else t.cpu()


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


# ----------------------------------------------------------------------------------------------------------------------
# public: matrices on the device the kernels run on
# ----------------------------------------------------------------------------------------------------------------------
def sweep_matrices(pose1, pose2s, K, device, mode=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sweep constants of ``cost_volume_fusion`` as device tensors (Hm [B,M,9], kt [B,M,3]) on ``device``."""
    pose2s = list(pose2s)
    if _mode(mode) == "reference":
        Hm, kt = sweep_matrices_host(_host(pose1), [_host(p) for p in pose2s], _host(K))
        return Hm.to(device), kt.to(device)
    from dvmvs.hip import ops
    return ops.sweep_matrices(pose1.to(device), [p.to(device) for p in pose2s], K.to(device))


def relative_pose(a, c, device, mode=None) -> torch.Tensor:
    """inverse(a) @ c as a [B,4,4] device tensor on ``device``."""
    if _mode(mode) == "reference":
        return relative_pose_host(_host(a), _host(c)).to(device)
    from dvmvs.hip import ops
    return ops.relative_pose(a.to(device), c.to(device))
