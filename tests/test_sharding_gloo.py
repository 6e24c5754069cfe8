"""CPU, world_size 2 over gloo: the sequence-sharded N>1 path (ownership, the throughput reduction used by bench.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.join(root, "deep-video-mvs_amd"))
    from dvmvs import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned = sharding.run_sharded(7, lambda s: s * s)
    frames, seconds, fps = sharding.reduce_throughput(local_frames=10 * len(owned), local_seconds=1.0 + rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, sorted(owned))
    if rank == 0:
        torch.save({"owned": gathered, "frames": frames, "seconds": seconds, "fps": fps}, os.path.join(out_dir, "result.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sequence_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "result.pt"), weights_only=False)
    assert res["owned"] == [[0, 2, 4, 6], [1, 3, 5]]                 # every sequence exactly once
    assert res["frames"] == 70.0 and res["seconds"] == 2.0 and abs(res["fps"] - 35.0) < 1e-9   # SUM frames / MAX time


def test_ownership_is_a_partition():
    import sys
    from dvmvs import sharding
    for n in (0, 1, 5, 8, 13):
        for world in (1, 2, 4, 8):
            seen = sorted(s for r in range(world) for s in sharding.sequences_for_rank(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sharding.sequences_for_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
