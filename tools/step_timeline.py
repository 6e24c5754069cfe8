"""Kernel-by-kernel timeline of a few steady-state steps out of a rocprofv3 kernel trace of bench.py --mark-region.

    python tools/step_timeline.py <kernel_trace.csv> [first_step] [n_steps] > timeline.txt

Steps are delimited by the ConvLSTM gate kernel (one per frame).  Per kernel: start offset from the first printed kernel (us), duration
(us), gap since the latest end seen so far (us; negative = overlapped with another queue's kernel), hardware queue, grid size, name.
Used to see what the critical path of a frame graph is (which stream's chain ends last, what runs next to the sweep).
"""
import csv
import sys


def main():
    src = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "trace_marker_kernel" in r["Kernel_Name"]]
    if len(marks) >= 2:
        rows = rows[marks[0] + 1:marks[-1]]
    gates = [i for i, r in enumerate(rows) if "lstm_gates" in r["Kernel_Name"]]
    if len(gates) < first + count + 1:
        raise SystemExit(f"only {len(gates)} steps in the marked region")
    part = rows[gates[first] + 1:gates[first + count] + 1]
    t0 = int(part[0]["Start_Timestamp"])
    latest = t0
    queues = sorted({r["Queue_Id"] for r in part})
    print(f"{len(part)} kernels in {count} steps; span {(int(part[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; queues {queues}")
    busy = {q: 0 for q in queues}
    for r in part:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        name = name[:name.index("(")] if "(" in name else name
        busy[r["Queue_Id"]] += e - s
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - latest) / 1e3:7.1f}  q{queues.index(r['Queue_Id'])} {r.get('Grid_Size', r.get('Grid_Size_X', '')):>8} {name[-110:]}")
        latest = max(latest, e)
        if "lstm_gates" in name:
            print("    ---- frame boundary (gates) ----")
    print("busy per queue (us):", {f"q{queues.index(q)}": round(v / 1e3, 1) for q, v in busy.items()})


if __name__ == "__main__":
    main()
