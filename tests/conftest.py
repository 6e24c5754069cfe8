import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device visible")
    return torch.device("cuda:0")


@pytest.fixture
def fixture_host_algebra(monkeypatch):
    """For tests that compare against fixtures CAPTURED ON THE FIXTURE HOST (tests/golden/*.npz from the reference run): replays
    that host's fp32 pose algebra (tests/golden/host_pose_algebra.npz) in place of the local LAPACK, in the product's host-side
    algebra (dvmvs.pose_algebra) and in the oracle, so that the kernels and the CPU pipelines are fed the very matrices the
    reference run computed.  Why: fp32 LAPACK's last bits depend on the CPU (tests/synthetic.py, "host pose algebra").
    A pose pair the table has not seen falls back to the local evaluation and fails the test at teardown."""
    import torch
    import dvmvs_oracle as orc
    import synthetic as syn
    from dvmvs import pose_algebra
    table = syn.FixtureHostAlgebra()
    misses = []
    local = dict(sweep=pose_algebra.sweep_matrices_host, rel=pose_algebra.relative_pose_host, orc_setup=orc.plane_sweep_setup,
                 orc_rel=orc.relative_pose)

    def replay(lookup, fallback, args, fp32_only=True):
        if fp32_only and (orc.POSE_ALGEBRA_DTYPE is not None or args[0].dtype != torch.float32):
            return fallback(*args)
        try:
            return lookup(*args)
        except KeyError:
            misses.append(tuple(tuple(a.shape) if hasattr(a, "shape") else len(a) for a in args))
            return fallback(*args)

    monkeypatch.setattr(pose_algebra, "sweep_matrices_host", lambda p1, p2s, K: replay(table.sweep_matrices_host, local["sweep"], (p1, p2s, K)))
    monkeypatch.setattr(pose_algebra, "relative_pose_host", lambda a, c: replay(table.relative_pose_host, local["rel"], (a, c)))
    monkeypatch.setattr(orc, "plane_sweep_setup", lambda p1, p2, K: replay(table.plane_sweep_setup, local["orc_setup"], (p1, p2, K)))
    monkeypatch.setattr(orc, "relative_pose", lambda a, c: replay(table.relative_pose_host, local["orc_rel"], (a, c)))
    yield table
    assert not misses, f"pose pairs missing from tests/golden/host_pose_algebra.npz (add them to syn.golden_algebra_pairs): {misses[:5]}"
