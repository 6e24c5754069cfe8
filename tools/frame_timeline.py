"""One frame of bench.py's timed region as an ordered kernel timeline (which kernels lie on the critical path, where the queue idles).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --mark-region --steps 20 --warmup 5 ...
    python tools/frame_timeline.py /tmp/prof/*/*kernel_trace.csv [frame index, default the middle one] > profiles/rNN_frame_timeline.txt

Frames are cut at dvmvs::copy_batch_kernel (the one launch a frame step makes in front of its graph).  Per kernel: start offset from the
frame's first kernel, duration, idle time of ITS queue before it, queue id, name; then per queue the busy time, and the kernels grouped
by class with the idle time in front of them -- on a one-stream run (--lookahead 0) busy + idle = the frame."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"dvmvs::", "", name)
    if name.startswith("Cijk"):
        m = re.search(r"MT\d+x\d+x\d+", name)
        return "rocblas_gemm_" + (m.group(0) if m else "")
    return name.split("(")[0][:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "trace_marker_kernel" in r["Kernel_Name"]]
    region = rows[marks[0] + 1:marks[-1]] if len(marks) >= 2 else rows
    cuts = [i for i, r in enumerate(region) if "copy_batch_kernel" in r["Kernel_Name"]]
    if len(cuts) < 3:
        raise SystemExit("no frame boundaries (copy_batch_kernel) in the timed region")
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(cuts) // 2
    frame = region[cuts[k]:cuts[k + 1]]
    t0 = int(frame[0]["Start_Timestamp"])
    wall = (int(region[cuts[k + 1]]["Start_Timestamp"]) - t0) / 1e3
    last_end = {}
    busy = defaultdict(float)
    idle_before = defaultdict(lambda: [0, 0.0, 0.0])
    print(f"# frame {k} of {len(cuts)}: {len(frame)} launches, {wall:.1f} us from its first kernel to the next frame's first")
    print("# start_us  dur_us  queue_idle_before_us  queue  kernel")
    for r in frame:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        d = (e - s) / 1e3
        busy[q] += d
        c = idle_before[short(r["Kernel_Name"])]
        c[0] += 1
        c[1] += d
        c[2] += max(gap, 0.0)
        print(f"{(s - t0) / 1e3:8.1f} {d:7.2f} {gap:7.2f}  q{q}  {short(r['Kernel_Name'])}")
    print("# per queue: busy us")
    for q, b in sorted(busy.items()):
        print(f"#   q{q}: {b:.1f}")
    print("# per kernel class: launches, busy us, idle us of its queue in front of them")
    for name, (n, d, g) in sorted(idle_before.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"#   {n:3d} {d:8.1f} {g:8.1f}  {name}")


if __name__ == "__main__":
    main()
