"""CPU, world size 2 (gloo): bench.py's own timing harness -- warm-up, barrier-bracketed timed region, MAX over ranks -- driven
with a stub step, i.e. the N > 1 code path the driver runs with `python -m torch.distributed.run ... bench.py --gpus N`, minus the
GPU.  (No multi-GPU node was available in rounds 1-3: this is the only execution that path gets here.)"""
import importlib.util
import json
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = load_bench()
    calls = []

    def step(i):                       # rank 1 is the slow one: the job's time is ITS time
        calls.append(i)
        time.sleep(0.004 * (1 + 2 * rank))

    elapsed = bench.timed_region(step, warmup=3, steps=10, world=world, device=torch.device("cpu"))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"elapsed": elapsed, "calls": calls}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_timed_region_world_size_two(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    assert r0["calls"] == r1["calls"] == list(range(13))                 # 3 warm-up + exactly 10 timed steps, numbered through
    assert r0["elapsed"] == r1["elapsed"]                                # the MAX over ranks, on every rank
    assert 10 * 0.012 <= r0["elapsed"] < 10 * 0.012 + 0.5                # the slow rank's 10 x 12 ms, not the fast rank's 10 x 4 ms


def test_timed_region_single_process():
    bench = load_bench()
    seen = []
    elapsed = bench.timed_region(lambda i: seen.append(i), warmup=2, steps=5, world=1, device=torch.device("cpu"),
                                 before=lambda: seen.append("before"), after=lambda: seen.append("after"))
    assert seen == [0, 1, "before", 2, 3, 4, 5, 6, "after"] and 0.0 <= elapsed < 0.5


def test_bench_py_gpus_2_launches_itself(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must not exit -- it re-executes itself under
    torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous) and rank 0 prints ONE JSON line.  Here: the stub step on CPU / gloo
    (DVMVS_BENCH_STUB_STEP_MS), the same launch path."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DVMVS_BENCH_STUB_STEP_MS"] = "3"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["data"] == "stub"
    assert rec["ms_per_step"] >= 6.0          # the slow rank (2 x 3 ms per step) decides
    # the launcher shape the driver uses still works unchanged
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1
    # a world size that contradicts --gpus is still an error
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "does not match" in bad.stderr
