#!/usr/bin/env python3
"""Probe (MI355X): where do the rare multi-millisecond steps come from?  One run in ~25 of the driver's 20-step bench command has ONE step whose
device gap is 4 - 6 ms (bench.py: device_ms_between_step_ends).  This runs the bench's headline loop (look-ahead 1, collector off, planning thread,
the bench's switch interval) for many steps in one process and prints every step whose host time or device gap exceeds a threshold, with the host
clock at the checkpoints of DepthEngine.step (engine.step_clock) and the seconds since the process started.

    python tools/hiccup_probe.py [steps, default 20000] [threshold ms, default 1.5]"""
import gc
import os
import sys
import time

T0 = time.perf_counter()
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("DVMVS_SWITCH_INTERVAL", "1e-4")

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    threshold = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    from dvmvs.engine import DepthEngine
    dev = torch.device("cuda:0")
    engine = DepthEngine(*bench.build_modules(), device=dev, max_lookahead=1)
    M, n_images = 2, 32
    images, seq, full_K = bench.synthetic_sequence(0, n_images, 300, M)
    images = [im.to(dev) for im in images]
    n_seq = len(seq) - 2
    events, host, clocks, wraps = [], [], [], set()
    with torch.no_grad():
        for k in range(M):
            engine._half_features(k, images[k % n_images])
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        for i in range(steps):
            k = M + (i % (n_seq - M - 1))
            if k == M and i:
                wraps.add(i)
                engine.new_sequence()      # (the index wraps: a new sequence on the same engine, as a caller with many sequences does)
                for j in range(M):
                    engine._half_features(j, images[j % n_images])
            ahead = dict(next_reference_image=images[(k + 1) % n_images], next_frame_id=k + 1, next_reference_pose=seq[k + 1][0],
                         next_measurement_poses=seq[k + 1][1], next_measurement_ids=[k - j for j in range(M)])
            marks = []
            engine.step_clock = marks
            t0 = time.perf_counter()
            engine.step(images[k % n_images], seq[k][0], None, seq[k][1], full_K, frame_id=k, measurement_ids=[k - 1 - j for j in range(M)], **ahead)
            t1 = time.perf_counter()
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            events.append(ev)
            host.append((t0 - T0, 1e3 * (t1 - t0)))
            clocks.append(marks)
    torch.cuda.synchronize()
    gaps = [0.0] + [events[i - 1].elapsed_time(events[i]) for i in range(1, len(events))]
    gaps = [g if i not in wraps else 0.0 for i, g in enumerate(gaps)]
    ordered = sorted(h for _, h in host)
    print(f"{steps} steps: host ms per step median {ordered[len(ordered) // 2]:.3f}, p99 {ordered[int(0.99 * len(ordered))]:.3f}, max {ordered[-1]:.3f}; "
          f"device gap median {sorted(gaps)[len(gaps) // 2]:.3f} ms, max {max(gaps):.3f}")
    n = 0
    for i, ((t, h), g, marks) in enumerate(zip(host, gaps, clocks)):
        if i in wraps:      # (the eager feature extraction of a new sequence's first frames sits in this gap: not a frame step)
            continue
        if h > threshold or g > threshold:
            parts = "  ".join(f"{name} {1e3 * (tt - marks[j][1]):6.3f}" for j, (name, tt) in enumerate(marks[1:])) if marks else ""
            print(f"  step {i:6d} at {t:8.2f} s: host {h:7.3f} ms, device gap {g:7.3f} ms | {parts}")
            n += 1
            if n > 60:
                break
    print(f"{n} steps above {threshold} ms")


if __name__ == "__main__":
    main()
