"""Probe (MI355X): wall time of every engine step of a short run (host clock at each step's return + one device sync at the end of
every step), to see what the first steps after the graph captures cost.   python tools/step_times_probe.py [lookahead]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    from dvmvs.engine import DepthEngine
    dev = torch.device("cuda:0")
    engine = DepthEngine(*bench.build_modules(), device=dev, max_lookahead=level)
    if "--no-gc" in sys.argv:
        import gc
        gc.disable()
    M, n_images, total = 2, 32, 70
    images, seq, full_K = bench.synthetic_sequence(0, n_images, total + M + 2, M)
    images = [im.to(dev) for im in images]
    rows, clocks, events = [], [], []
    with torch.no_grad():
        for k in range(M):
            engine._half_features(k, images[k % n_images])
        torch.cuda.synchronize()
        for i in range(total):
            k = M + i
            ahead = {}
            if level >= 1:
                ahead = dict(next_reference_image=images[(k + 1) % n_images], next_frame_id=k + 1)
                ahead.update(next_reference_pose=seq[k + 1][0], next_measurement_poses=seq[k + 1][1], next_measurement_ids=[k - j for j in range(M)])
            sync_each = "--free-running" not in sys.argv
            engine.step_clock = clocks
            t0 = time.perf_counter()
            engine.step(images[k % n_images], seq[k][0], None, seq[k][1], full_K, frame_id=k, measurement_ids=[k - 1 - j for j in range(M)], **ahead)
            t1 = time.perf_counter()
            events.append(torch.cuda.Event(enable_timing=True))
            events[-1].record()
            if sync_each:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            rows.append((i, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    torch.cuda.synchronize()
    gaps = [events[i - 1].elapsed_time(events[i]) for i in range(1, len(events))]
    print("device time between the ends of consecutive steps (ms):", " ".join(f"{g:.2f}" for g in gaps))
    print(f"look-ahead {level}: step, host ms inside step(), ms until the device is idle (a sync after every step: no overlap between steps)")
    print("frames per sweep configuration:", engine.sweep_variant_counts, " graphs:", len(engine._graphs), " planned blocks used:", engine.planned_frames_used)
    for (i, h, d), marks in zip(rows, clocks):
        parts = "  ".join(f"{name} {1e3 * (t - marks[j][1]):6.3f}" for j, (name, t) in enumerate(marks[1:]))
        print(f"  {i:3d}  {h:8.3f}  {d:8.3f}   | {parts}")


if __name__ == "__main__":
    main()
