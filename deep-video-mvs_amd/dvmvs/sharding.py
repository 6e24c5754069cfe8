"""Sequence-sharded multi-GPU execution: one process per GPU, independent video sequences, no data-path collective.

Frames of one sequence are strictly ordered (LSTM state, previous depth: /root/reference/dvmvs/fusionnet/run-testing.py:
86-88, 201-202) while sequences are independent, so the only way this workload shards is by sequence: sequence ``s`` is
owned by rank ``s % world``; weights (139 MB) are replicated.  The only communication is the timing/throughput reduction
at the end (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def sequences_for_rank(n_sequences, rank, world):
    """Round-robin ownership: every sequence has exactly one owner, loads differ by at most one sequence."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_sequences, world))


def reduce_throughput(local_frames, local_seconds, device="cpu", group=None):
    """Whole-job frames and the slowest rank's wall time (SUM / MAX all-reduce).  Returns (frames, seconds, frames/s)."""
    if not dist.is_available() or not dist.is_initialized():
        return local_frames, local_seconds, local_frames / local_seconds if local_seconds > 0 else 0.0
    frames = torch.tensor([float(local_frames)], dtype=torch.float64, device=device)
    seconds = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(frames, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(seconds, op=dist.ReduceOp.MAX, group=group)
    total, slowest = frames.item(), seconds.item()
    return total, slowest, total / slowest if slowest > 0 else 0.0


def run_sharded(n_sequences, run_sequence, rank=None, world=None):
    """Calls ``run_sequence(sequence_id) -> result`` for every sequence this rank owns; returns {sequence_id: result}."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return {s: run_sequence(s) for s in sequences_for_rank(n_sequences, rank, world)}
