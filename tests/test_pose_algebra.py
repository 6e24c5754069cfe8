"""CPU: the host side of the ABI-3 boundary -- dvmvs.pose_algebra's "reference" mode must hand the kernels the very matrices the
reference computes (/root/reference/dvmvs/utils.py:51-56, :121; dvmvs/convlstm.py:30).  The oracle restates those lines and is
pinned to the imported reference by tests/golden (PINNING_REPORT.json), so bit-equality with the oracle's fp32 evaluation is
bit-equality with the reference's CPU run."""
import numpy as np
import pytest
import torch

import dvmvs_oracle as orc
import synthetic as syn

PAIRS = [(9, (6, 0)), (141, (135, 130)), (202, (196, 188)), (250, (249, 240)), (13, (12, 9, 3))]


def test_reference_mode_is_bit_identical_to_the_reference_expressions():
    from dvmvs import pose_algebra
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    for r, ms in PAIRS:
        p1, p2s = syn.pose(r), [syn.pose(m) for m in ms]
        Hm, kt = pose_algebra.sweep_matrices(p1, p2s, halfK, "cpu", "reference")
        assert Hm.dtype == kt.dtype == torch.float32 and tuple(Hm.shape) == (1, len(ms), 9) and tuple(kt.shape) == (1, len(ms), 3)
        for m, p2 in enumerate(p2s):
            H_ref, k_ref = orc.plane_sweep_setup(p1, p2, halfK)
            assert torch.equal(Hm[:, m], H_ref.reshape(1, 9)) and torch.equal(kt[:, m], k_ref.reshape(1, 3)), (r, ms[m])
        T = pose_algebra.relative_pose(p1, p2s[0], "cpu", "reference")
        assert torch.equal(T, orc.relative_pose(p1, p2s[0]))


def test_batched_poses_give_the_per_item_matrices():
    """B > 1 (lock-step sequences, training batches): every batch item's matrices equal its batch-1 evaluation bit for bit
    (LAPACK and ATen's small bmm work matrix by matrix)."""
    from dvmvs import pose_algebra
    K = torch.cat([syn.scaled_K(syn.full_K(), 2.0) * torch.tensor([1.0 + 0.01 * b]) for b in range(3)])
    K[:, 2, 2] = 1.0
    p1 = torch.cat([syn.pose(i) for i in (9, 141, 202)])
    p2s = [torch.cat([syn.pose(i) for i in (6, 135, 196)]), torch.cat([syn.pose(i) for i in (0, 130, 188)])]
    Hm, kt = pose_algebra.sweep_matrices_host(p1, p2s, K)
    for b in range(3):
        H1, k1 = pose_algebra.sweep_matrices_host(p1[b:b + 1], [p[b:b + 1] for p in p2s], K[b:b + 1])
        assert torch.equal(Hm[b:b + 1], H1) and torch.equal(kt[b:b + 1], k1)
    T = pose_algebra.relative_pose_host(p1, p2s[0])
    for b in range(3):
        assert torch.equal(T[b:b + 1], pose_algebra.relative_pose_host(p1[b:b + 1], p2s[0][b:b + 1]))


def test_fp32_pose_algebra_error_is_what_the_header_says():
    """The reason the matrices are an argument: fp32 inverse(pose2) @ pose1 is ~1e-7..1e-6 m off in the relative translation,
    which K t / depth turns into > 1e-4 px on the nearest plane -- larger than any other rounding on the path."""
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    worst = 0.0
    for r, ms in PAIRS:
        for m in ms:
            _, k32 = orc.plane_sweep_setup(syn.pose(r), syn.pose(m), halfK)
            _, k64 = orc.plane_sweep_setup(syn.pose(r).double(), syn.pose(m).double(), halfK.double())
            worst = max(worst, float((k32.double() - k64).abs().max()) / 0.25)      # pixels at the 0.25 m plane (Z ~ 1)
    assert 2e-5 < worst < 2e-3, worst


def test_modes_and_errors():
    from dvmvs import pose_algebra
    with pytest.raises(ValueError):
        pose_algebra.relative_pose(syn.pose(9), syn.pose(6), "cpu", "fast")
    with pytest.raises(RuntimeError, match="MI355X"):     # "exact" is a device kernel: no CPU evaluation
        pose_algebra.sweep_matrices(syn.pose(9), [syn.pose(6)], syn.full_K(), "cpu", "exact")
    assert pose_algebra.MODE == "reference"
    assert np.isfinite(pose_algebra.sweep_matrices_host(syn.pose(9), [syn.pose(6)], syn.full_K())[0].numpy()).all()
