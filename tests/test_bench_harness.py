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
