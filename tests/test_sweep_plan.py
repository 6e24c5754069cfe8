"""Host-side run-plan model of the LDS-tiled sweep (dvmvs_sweep_plan_stats / dvmvs_sweep_select_variant, csrc/sweep_tiled.hip:
host_plan_stats) -- runs on the CPU, no GPU needed.  Checked against an independent pure-Python restatement of the plan rule
(corner boxes of a tile over a run of planes, greedy halving, LDS capacity) on real keyframe geometries, and for the properties the
engine relies on: deterministic, 2 or 3 only, the wide configuration never queues more than the default one."""
import ctypes
import math

import numpy as np
import pytest
import torch

import synthetic as syn
from dvmvs import pose_algebra
from dvmvs.hip import _capi

H, W, D = 128, 160, 64
f32 = np.float32


def lib_stats(Hm, kt, configuration, shape=(H, W, D)):
    out = (ctypes.c_longlong * 8)()
    rc = _capi.lib().dvmvs_sweep_plan_stats(Hm.contiguous().data_ptr(), kt.contiguous().data_ptr(), Hm.shape[0], Hm.shape[1], shape[0], shape[1],
                                            shape[2], 0.25, 20.0, configuration, out)
    assert rc == 0
    return list(out)


def fma(a, b, c):      # fmaf through float64: the product is exact there
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def pitch_residue(ax, ay):
    aax, aay = abs(ax), abs(ay)
    c, r = math.ceil(15.0 * aax), math.ceil(15.0 * aay)
    cost0 = 15.0 * aay * max(0.0, 1.0 - aax) + max(0.0, c - 15.0)
    cost1 = max(0.0, c + r - 15.0)
    if not cost1 < cost0:
        return 0
    return 15 if (ax < 0) != (ay < 0) else 1


def python_plan_stats(Hm, kt, cap, tw=32, th=8, dp=8, minseg=2):
    """The kernel's plan rule, one (tile, chunk, frame) at a time, in numpy float32 scalars."""
    M = Hm.shape[0]
    stats = [0] * 8
    inv_base, inv_step = 1.0 / 20.0, (1.0 / 0.25 - 1.0 / 20.0) / (D - 1)
    wn, hn, Wm1, Hm1 = f32(W) * f32(0.5), f32(H) * f32(0.5), f32(W - 1), f32(H - 1)

    def position(ray, k):
        denom = f32(f32(ray[2] + k[2]) + f32(1e-8))
        u, v = f32(f32(ray[0] + k[0]) / denom), f32(f32(ray[1] + k[1]) / denom)
        ix = f32(f32(f32(f32(f32(u - wn) / wn) + f32(1.0)) * f32(0.5)) * Wm1)
        iy = f32(f32(f32(f32(f32(v - hn) / hn) + f32(1.0)) * f32(0.5)) * Hm1)
        return ix, iy, denom

    with np.errstate(all="ignore"):
        for chunk in range(D // dp):
            ktd = [[[f32(kt[m][k] / f32(1.0 / (inv_base + (chunk * dp + j) * inv_step))) for k in range(3)] for j in range(dp)] for m in range(M)]
            for ty in range(math.ceil(H / th)):
                for tx in range(math.ceil(W / tw)):
                    x0, x1 = tx * tw, min(tx * tw + tw - 1, W - 1)
                    y0, y1 = ty * th, min(ty * th + th - 1, H - 1)
                    edge = f32(max(1, x1 - x0))
                    spills = False
                    group_runs = group_queued = 0
                    for m in range(M):
                        h = Hm[m]
                        rays = []
                        for c in range(4):
                            xf, yf = f32(x1 if c & 1 else x0), f32(y1 if c & 2 else y0)
                            rays.append(tuple(fma(h[3 * r + 2], f32(1.0), fma(h[3 * r + 1], yf, f32(h[3 * r] * xf))) for r in range(3)))
                        lo, hint = 0, dp
                        while lo < dp:
                            len0 = min(dp - lo, hint)
                            for candidate in range(4):
                                length = len0
                                for _ in range(candidate):
                                    length = max((length + 1) // 2, minseg)
                                length = min(length, len0)
                                pts = [position(rays[c & 3], ktd[m][lo + length - 1 if c & 4 else lo]) for c in range(8)]
                                finite = all(-1e6 < p[0] < 1e6 and -1e6 < p[1] < 1e6 and p[2] > 1e-6 for p in pts)
                                state, records = 0, 0
                                if finite:
                                    lo_x, hi_x = min(p[0] for p in pts), max(p[0] for p in pts)
                                    lo_y, hi_y = min(p[1] for p in pts), max(p[1] for p in pts)
                                    s = f32(0.05)
                                    if f32(hi_x + s) <= -1 or f32(lo_x - s) >= W or f32(hi_y + s) <= -1 or f32(lo_y - s) >= H:
                                        state = 2
                                    else:
                                        bx0, by0 = max(-1, math.floor(max(f32(lo_x - s), -1))), max(-1, math.floor(max(f32(lo_y - s), -1)))
                                        bx1, by1 = min(W, math.floor(min(f32(hi_x + s), W))) + 1, min(H, math.floor(min(f32(hi_y + s), H))) + 1
                                        RW, RH = bx1 - bx0 + 1, by1 - by0 + 1
                                        step = (f32(f32(pts[1][0] - pts[0][0]) / edge), f32(f32(pts[1][1] - pts[0][1]) / edge))
                                        pitch = RW + ((pitch_residue(float(step[0]), float(step[1])) - RW) & 15)
                                        records = pitch * RH
                                        state = 1 if records <= cap else 0
                                if state != 0 or length <= minseg or candidate == 3:
                                    break
                            if state == 1:
                                stats[0] += 1
                                stats[1] += records
                                group_runs += 1
                            elif state == 2:
                                stats[2] += 1
                            else:
                                stats[3] += 1
                                stats[4] += length
                                group_queued += length
                                spills = True
                            lo += length
                            hint = max(length, minseg)
                    stats[5] += spills
                    stats[6] = max(stats[6], group_runs)        # the longest chain of staged runs of one workgroup ...
                    stats[7] = max(stats[7], group_queued)      # ... and the most planes one workgroup queues for the second pass
    return stats


def matrices(line):
    r, ms = syn.keyframe_index_lines(2)[line]
    return pose_algebra.sweep_matrices_host(syn.pose(r), [syn.pose(m) for m in ms], syn.scaled_K(syn.full_K(), 2.0))


@pytest.mark.parametrize("line", [0, 117, 202])
def test_plan_model_equals_the_python_restatement(line):
    Hm, kt = matrices(line)
    for configuration, cap in ((0, 1024), (1, 1536)):
        assert lib_stats(Hm, kt, configuration) == python_plan_stats(Hm[0].numpy().astype(f32), kt[0].numpy().astype(f32), cap), (line, configuration)


def test_selection_properties_on_the_whole_keyframe_index():
    lib = _capi.lib()
    chosen = []
    for line in range(len(syn.keyframe_index_lines(2))):
        Hm, kt = matrices(line)
        d, w = lib_stats(Hm, kt, 0), lib_stats(Hm, kt, 1)
        assert w[4] <= d[4] and w[5] <= d[5], (line, d, w)              # larger boxes never queue more
        assert d[0] + d[2] + d[3] >= 2 * 640 and sum(d) > 0             # every (tile, chunk, frame) is covered by at least one run
        v = pose_algebra.sweep_variant_host(Hm, kt, H, W, D, 0.25, 20.0)
        assert v in (2, 3) and v == lib.dvmvs_sweep_select_variant(Hm.data_ptr(), kt.data_ptr(), 1, 2, H, W, D, 0.25, 20.0)    # deterministic
        chosen.append(v)
    # easy sideways pairs keep the default configuration, the pairs whose footprints do not fit 48 KB boxes take the wide one
    assert chosen[0] == 2 and chosen[153] == 2
    assert chosen[170] == 3 and chosen[202] == 3
    assert 0 < chosen.count(3) < len(chosen) // 2


def test_plan_model_rejects_bad_arguments():
    Hm, kt = matrices(0)
    out = (ctypes.c_longlong * 8)()
    lib = _capi.lib()
    assert lib.dvmvs_sweep_plan_stats(None, kt.data_ptr(), 1, 2, H, W, D, 0.25, 20.0, 0, out) == -1
    assert lib.dvmvs_sweep_plan_stats(Hm.data_ptr(), kt.data_ptr(), 1, 2, H, W, D, 0.25, 20.0, 2, out) == -1
    assert lib.dvmvs_sweep_plan_stats(Hm.data_ptr(), kt.data_ptr(), 1, 9, H, W, D, 0.25, 20.0, 0, out) == -2
    with pytest.raises(ValueError):
        pose_algebra.sweep_variant_host(Hm.double(), kt.double(), H, W, D, 0.25, 20.0)


def work_list(Hm, kt, variant, shape=(H, W, D)):
    from dvmvs.hip import ops
    words = ops.sweep_work_list_host(Hm, kt, shape[0], shape[1], shape[2], 0.25, 20.0, variant).numpy().astype(np.int64) & 0xffffffff
    n = int(words[0])
    return n, words[2:2 + 2 * n].reshape(n, 2)


@pytest.mark.parametrize("line", [0, 74, 170, 202, 242])
def test_work_list_covers_every_plane_of_every_tile_exactly_once(line):
    """dvmvs_sweep_work_list: the items partition (tile, plane); a (tile, chunk) with a long chain of staged runs is cut, an easy
    geometry is not; the static positions keep their (tile, chunk); at most twice the static count; deterministic."""
    Hm, kt = matrices(line)
    for variant, cap in ((2, 1024), (3, 1536)):
        n, items = work_list(Hm, kt, variant)
        cover = np.zeros((80, D), dtype=np.int64)
        for w0, w1 in items:
            assert (w0 >> 16) == 0
            cover[w0 & 0xffff, (w1 & 0xffff):(w1 & 0xffff) + (w1 >> 16)] += 1
        assert (cover == 1).all(), (line, variant)
        assert 640 <= n <= 1280
        sizes = items[:, 1] >> 16
        assert set(np.unique(sizes)) <= {2, 4, 8}
        stats = lib_stats(Hm, kt, 0 if variant == 2 else 1)
        if stats[6] <= 3:
            assert n == 640 and (sizes == 8).all()                  # nothing to cut
        else:
            assert n > 640 and (sizes < 8).any()
        assert (sizes[:640] > 0).all()
        n2, items2 = work_list(Hm, kt, variant)
        assert n2 == n and np.array_equal(items, items2)
    # the first 640 positions are the static numbering's (tile, chunk) pairs: every pair once
    pairs = {(int(w0 & 0xffff), int(w1 & 0xffff) // 8) for w0, w1 in items[:640]}
    assert len(pairs) == 640


def test_work_list_rejects_bad_arguments():
    from dvmvs.hip import ops
    Hm, kt = matrices(0)
    with pytest.raises(ValueError):
        ops.sweep_work_list_host(Hm.double(), kt.double(), H, W, D, 0.25, 20.0, 2)
    with pytest.raises(ValueError):
        ops.sweep_work_list_host(Hm, kt, H, W, D, 0.25, 20.0, 2, out=torch.empty(16, dtype=torch.int32))
    lib = _capi.lib()
    assert lib.dvmvs_sweep_work_list(Hm.data_ptr(), kt.data_ptr(), 1, 2, H, W, D, 0.25, 20.0, 0, None, 0) == -1
    assert lib.dvmvs_sweep_work_list_bytes(0, H, W, D) == 0


def test_one_walk_plan_equals_selection_plus_work_list():
    """dvmvs_sweep_plan (what the frame engine calls once per keyframe) = dvmvs_sweep_select_variant + dvmvs_sweep_work_list of the
    chosen configuration, on every third keyframe pair of the sample scene; it returns the single-pass variant (4 / 5) exactly when
    the chosen configuration's plan queues nothing for the second pass; a forced configuration is respected."""
    from dvmvs.hip import ops
    out = torch.zeros(ops.sweep_work_list_words(1, H, W, D), dtype=torch.int32)
    chosen = []
    for line in range(0, len(syn.keyframe_index_lines(2)), 3):
        Hm, kt = matrices(line)
        v = ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, 0, out)
        configuration = {2: 2, 3: 3, 4: 2, 5: 3}[v]
        assert configuration == pose_algebra.sweep_variant_host(Hm, kt, H, W, D, 0.25, 20.0)
        queued_runs = lib_stats(Hm, kt, configuration - 2)[3]
        assert (v >= 4) == (queued_runs == 0), (line, v, queued_runs)
        ref = ops.sweep_work_list_host(Hm, kt, H, W, D, 0.25, 20.0, v)
        used = 2 + 2 * int(ref[0])
        assert int(out[0]) == int(ref[0]) and torch.equal(out[:used], ref[:used]), line
        chosen.append(v)
    assert {2, 4} <= set(chosen) and (3 in chosen or 5 in chosen)
    Hm, kt = matrices(0)
    for forced in (2, 3):
        assert ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, forced, out) == forced + 2      # (line 0 queues nothing)
        ref = ops.sweep_work_list_host(Hm, kt, H, W, D, 0.25, 20.0, forced)
        assert torch.equal(out[:2 + 2 * int(ref[0])], ref[:2 + 2 * int(ref[0])])
    Hm, kt = matrices(170)
    assert ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, 2, out) == 2                           # (line 170 queues runs in both)
    with pytest.raises(RuntimeError):
        ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, 1, out)


@pytest.mark.parametrize("M", [4, 8])
def test_work_list_never_writes_past_its_buffer(M):
    """with M >= 4 measurement frames every (tile, chunk) has more staged runs than an item may have, so everything is
    cut, and the capacity guard has to count the ranges still pending on the cut stack.  A canary behind the buffer, random geometries,
    both configurations: the item count never exceeds the capacity, the canary survives, the items still partition (tile, plane)."""
    lib = _capi.lib()
    rng = np.random.default_rng(7 + M)
    shape = (128, 160, 64)
    words = int(lib.dvmvs_sweep_work_list_bytes(1, *shape)) // 4
    capacity = (words - 2) // 2
    poses = torch.from_numpy(syn.sample_poses()).float()
    K = syn.scaled_K(syn.full_K(), 2.0)
    for trial in range(60):
        ref = int(rng.integers(0, poses.shape[0]))
        meas = [int(np.clip(ref + rng.integers(-40, 41), 0, poses.shape[0] - 1)) for _ in range(M)]
        Hm, kt = pose_algebra.sweep_matrices_host(poses[ref:ref + 1], [poses[m:m + 1] for m in meas], K)
        Hm, kt = Hm.contiguous().float(), kt.contiguous().float()
        for configuration in (0, 1):
            buf = np.full(words + 64, 0xDEADBEEF, dtype=np.uint32)
            rc = lib.dvmvs_sweep_work_list(Hm.data_ptr(), kt.data_ptr(), 1, M, *shape, 0.25, 20.0, configuration,
                                           buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), words * 4)
            assert rc > 0, (trial, configuration, rc)
            assert (buf[words:] == 0xDEADBEEF).all(), (trial, configuration, "wrote past the work list")
            n = int(buf[0])
            assert n <= capacity and rc == 2 + 2 * n
            items = buf[2:2 + 2 * n].astype(np.int64).reshape(n, 2)
            cover = np.zeros((80, 64), dtype=np.int64)
            for w0, w1 in items:
                cover[w0 & 0xffff, (w1 & 0xffff):(w1 & 0xffff) + (w1 >> 16)] += 1
            assert (cover == 1).all(), (trial, configuration)


def test_plan6_takes_every_single_item_pair_and_no_batch():
    """dvmvs_sweep_plan6 (what the frame engine calls when its measurement maps are channels-last).  Round 6: the correlate-then-interpolate sweep
    (variant 6) in its persistent form with gather passes is the faster kernel on 255 of the sample scene's 285 keyframe pairs and within 1 - 7 us on
    the rest (round 5: taken below 14 estimated tiles per wave, forward-motion lines 98 - 100 left to the tiled kernel: 150 - 290 us against 73 - 82; now
    37 - 60 us), so every single-item launch gets 6 with an EMPTY work list (a tiled launch on it does nothing) and the tiled plan is not walked; never 6
    for a lock-step batch (the tiled plan and its list, as dvmvs_sweep_plan makes them).  The host-side estimate stays available."""
    from dvmvs.hip import ops
    lib = _capi.lib()
    words = ops.sweep_work_list_words(1, H, W, D)
    for line in (17, 18, 66, 98, 99, 100, 184):
        Hm, kt = matrices(line)
        out = torch.full((words,), 7, dtype=torch.int32)
        v = ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, 0, out, allow_mfma=True)
        assert v == 6 and int(out[0]) == 0 and int(out[1]) == 0, (line, v)
        plain = ops.sweep_plan_host(Hm, kt, H, W, D, 0.25, 20.0, 0, torch.zeros(words, dtype=torch.int32))
        assert plain in (2, 3, 4, 5)
        st = (ctypes.c_double * 4)()
        Hc, kc = Hm.contiguous().float(), kt.contiguous().float()
        assert lib.dvmvs_sweep_mfma_estimate(Hc.data_ptr(), kc.data_ptr(), 1, 2, H, W, D, 0.25, 20.0, st) == 0
        assert 0.0 < st[0] < 1e5 and 0.0 <= st[3] <= 1.0
    Hm, kt = matrices(17)
    both = ops.sweep_plan_host(Hm.repeat(2, 1, 1), kt.repeat(2, 1, 1), H, W, D, 0.25, 20.0, 0, torch.zeros(ops.sweep_work_list_words(2, H, W, D), dtype=torch.int32),
                               allow_mfma=True)
    assert both in (2, 3, 4, 5)
    assert lib.dvmvs_sweep_mfma_estimate(None, kt.contiguous().data_ptr(), 1, 2, H, W, D, 0.25, 20.0, (ctypes.c_double * 4)()) == -1
