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


class _StubEngine:
    """Stands in for DepthEngine on the CPU: the sharded scene runner only needs the frame contract."""
    cache_features = False

    def __init__(self, rank):
        self.device, self.rank, self._feature_cache, self.sequences, self.resets = torch.device("cpu"), rank, {}, 0, 0

    def new_sequence(self):
        self.sequences += 1
        self.announced = None

    def reset(self):
        self.resets += 1

    announced = None

    def step(self, image, pose, measurement_images, measurement_poses, full_K, frame_id=None, measurement_ids=None,
             next_reference_image=None, next_frame_id=None, next_reference_pose=None, next_measurement_poses=None, next_measurement_ids=None):
        assert tuple(image.shape) == (1, 3, 256, 320) and len(measurement_images) == len(measurement_poses) == 2
        # feature look-ahead of the offline runner: the frame announced by the previous call is the one that comes, with its image
        if self.announced is not None:
            assert self.announced[0] == frame_id and torch.equal(self.announced[1], image)
        self.announced = None if next_reference_image is None else (next_frame_id, next_reference_image.clone())
        return torch.full((1, 256, 320), 1.0 + self.rank + 0.01 * frame_id)


def _scene_worker(rank, world, port, out_dir):
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.join(root, "deep-video-mvs_amd"))
    from dvmvs import runner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scenes = [os.path.join(out_dir, f"scene{i}") for i in range(3)]
    indices = [os.path.join(out_dir, f"index{i}") for i in range(3)]
    built = []

    def make_engine():
        built.append(_StubEngine(rank))
        return built[-1]

    results, (frames, seconds, fps) = runner.predict_sharded(make_engine, scenes, indices, evaluate=False)
    summary = {s: [float(p.mean()) for p in r[0]] for s, r in results.items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, (summary, len(built), built[0].sequences if built else 0, built[0].resets if built else 0))
    if rank == 0:
        torch.save({"gathered": gathered, "frames": frames, "fps": fps}, os.path.join(out_dir, "scenes.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_scene_runner(tmp_path):
    """BASELINE.json configs[3] on the CPU: three scene folders over two gloo ranks through runner.predict_sharded (real scene
    folders, keyframe index files with a TRACKING LOST line, real pre-processing; a stub engine in place of the GPU one)."""
    from test_runner import _write_scene
    for i in range(3):
        _write_scene(os.path.join(str(tmp_path), f"scene{i}"), 8)
        with open(os.path.join(str(tmp_path), f"index{i}"), "w") as f:
            f.write("00003.png 00002.png 00001.png\nTRACKING LOST\n00005.png 00003.png 00002.png\n" + ("00007.png 00005.png 00003.png\n" if i == 1 else ""))
    world = 2
    mp.spawn(_scene_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "scenes.pt"), weights_only=False)
    (s0, built0, seq0, resets0), (s1, built1, seq1, resets1) = res["gathered"]
    assert sorted(s0) == [0, 2] and sorted(s1) == [1]                       # scene s runs on rank s % 2, exactly once
    assert built0 == 1 and built1 == 1 and seq0 == 2 and seq1 == 1         # one engine per rank, one new_sequence() per scene
    assert resets0 == 2 and resets1 == 1                                  # the TRACKING LOST lines
    assert [round(v, 2) for v in s0[0]] == [1.03, 1.05] and [round(v, 2) for v in s1[1]] == [2.03, 2.05, 2.07]
    assert res["frames"] == 7.0 and res["fps"] > 0


def _more_ranks_than_scenes_worker(rank, world, port, out_dir):
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.join(root, "deep-video-mvs_amd"))
    from dvmvs import runner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    built = []

    def make_engine():
        built.append(_StubEngine(rank))
        return built[-1]

    results, (frames, seconds, fps) = runner.predict_sharded(make_engine, [os.path.join(out_dir, "scene0")], [os.path.join(out_dir, "index0")],
                                                              evaluate=False)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sorted(results), len(built), frames))
    if rank == 0:
        torch.save(gathered, os.path.join(out_dir, "few_scenes.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_without_a_scene_still_joins_the_reduction(tmp_path):
    """more ranks than scenes (8 GPUs, 5 scenes).  The idle rank builds no engine; the throughput reduction must pick
    its device from the BACKEND, not from the engine, or that rank hands RCCL a CPU tensor and the others block in all_reduce.
    With gloo both are CPU tensors, so what this checks is that every rank reaches the collective and sees the job's totals."""
    from test_runner import _write_scene
    _write_scene(os.path.join(str(tmp_path), "scene0"), 8)
    with open(os.path.join(str(tmp_path), "index0"), "w") as f:
        f.write("00003.png 00002.png 00001.png\n00005.png 00003.png 00002.png\n")
    mp.spawn(_more_ranks_than_scenes_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    (scenes0, built0, frames0), (scenes1, built1, frames1) = torch.load(os.path.join(str(tmp_path), "few_scenes.pt"), weights_only=False)
    assert scenes0 == [0] and scenes1 == [] and built0 == 1 and built1 == 0
    assert frames0 == frames1 == 2.0
