"""CPU tests of host-side logic of the frame engine that needs no device: the queue that turns a step's input copies into one launch
(dvmvs/engine.py: DepthEngine._copy / _flush_copies).  The device entry (dvmvs_copy_batch) is tested on the GPU in
tests/test_sweep_mfma_gpu.py; here the batching op is replaced by a recorder so that the ORDER and GROUPING of the copies can be checked:
copies whose ranges do not touch are grouped, a copy that reads or overwrites a range a queued copy writes (or overwrites one it reads)
goes into a later group, and the result is always what the same copies give when executed one by one in program order."""
import torch

from dvmvs import engine as engine_module
from dvmvs.engine import DepthEngine


class _Recorder:
    """Stands in for dvmvs.hip.ops: everything dense is 'batchable', a batch is executed pair by pair and its size recorded."""

    def __init__(self):
        self.batches = []

    def batchable(self, dst, src):
        return dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous() and dst.numel() % 4 == 0

    def copy_batch(self, pairs):
        self.batches.append(len(pairs))
        spans = [((d.data_ptr(), d.data_ptr() + 4 * d.numel()), (s.data_ptr(), s.data_ptr() + 4 * s.numel())) for d, s in pairs]
        for i, (di, si) in enumerate(spans):      # the entry's contract: no write range touches another pair's write or read range
            for j, (dj, sj) in enumerate(spans):
                if i != j:
                    assert not (di[0] < dj[1] and dj[0] < di[1]) and not (di[0] < sj[1] and sj[0] < di[1]), "dependent copies in one launch"
        snapshot = [(d, s.clone()) for d, s in pairs]      # (a launch reads every source before any of its writes is visible to the others)
        for d, s in snapshot:
            d.copy_(s)


def _bare_engine(monkeypatch):
    rec = _Recorder()
    monkeypatch.setattr(engine_module, "_ops", rec)
    monkeypatch.setattr(engine_module, "_BATCH_COPIES", True)
    eng = DepthEngine.__new__(DepthEngine)
    eng._copy_queue = []
    return eng, rec


def test_independent_copies_become_one_launch(monkeypatch):
    eng, rec = _bare_engine(monkeypatch)
    pool = torch.zeros(6, 64)
    srcs = [torch.full((64,), float(i + 1)) for i in range(4)]
    for i in range(4):
        eng._copy(pool[i], srcs[i])
    assert rec.batches == []                                  # nothing issued before the flush
    eng._flush_copies()
    assert rec.batches == [4] and eng._copy_queue == []
    for i in range(4):
        assert torch.equal(pool[i], srcs[i])
    eng._copy(pool[4], srcs[0])                               # a single queued copy is a plain copy_
    eng._flush_copies()
    assert rec.batches == [4] and torch.equal(pool[4], srcs[0])


def test_dependent_copies_are_split_in_program_order(monkeypatch):
    """slot <- features; buffer <- slot (read after write), slot <- other (write after read / write after write): the later copy starts a new
    group, and the outcome equals sequential execution."""
    eng, rec = _bare_engine(monkeypatch)
    g = torch.Generator().manual_seed(0)
    mem = torch.zeros(8, 32)
    a, b = torch.randn(32, generator=g), torch.randn(32, generator=g)
    program = [(mem[0], a), (mem[1], mem[0]), (mem[0], b), (mem[2], mem[1]), (mem[3], a), (mem[0], mem[3])]
    expect = torch.zeros(8, 32)
    for (d, s_) in program:
        exp_d = expect[(d.data_ptr() - mem.data_ptr()) // (32 * 4)]
        exp_s = s_ if s_.data_ptr() < mem.data_ptr() or s_.data_ptr() >= mem.data_ptr() + mem.numel() * 4 else expect[(s_.data_ptr() - mem.data_ptr()) // (32 * 4)]
        exp_d.copy_(exp_s.clone())
    for d, s_ in program:
        eng._copy(d, s_)
    eng._flush_copies()
    assert torch.equal(mem, expect)
    assert rec.batches == [3]      # (m0 <- b, m2 <- m1, m3 <- a) went out together; the dependent ones one by one -- the recorder checks every launch


def test_outside_a_step_and_for_odd_tensors_the_copy_is_immediate(monkeypatch):
    eng, rec = _bare_engine(monkeypatch)
    dst, src = torch.zeros(2, 6), torch.ones(2, 6)
    eng._copy_queue = None                                    # not inside step(): executed at once
    eng._copy(dst, src)
    assert torch.equal(dst, src) and rec.batches == []
    eng._copy_queue = []
    queued_dst, queued_src = torch.zeros(8), torch.arange(8.0)
    eng._copy(queued_dst, queued_src)
    odd_dst, odd_src = torch.zeros(3), torch.ones(3)          # not a multiple of 4 elements: flushes the queue, then copies directly
    eng._copy(odd_dst, odd_src)
    assert torch.equal(queued_dst, queued_src) and torch.equal(odd_dst, odd_src) and eng._copy_queue == []
    for i in range(9):                                        # the entry takes eight copies: the ninth starts a new launch
        eng._copy(torch.zeros(4), torch.ones(4))
    assert rec.batches == [8] and len(eng._copy_queue) == 1
