"""Per-kernel summary of the TIMED REGION of a rocprofv3 kernel trace of bench.py.

    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --mark-region ...
    python tools/summarize_trace.py /tmp/prof/bench_kernel_trace.csv profiles/r01_bench_timed_region.csv [steps]

bench.py --mark-region brackets its timed loop with the library's empty marker kernel (dvmvs::trace_marker_kernel);
everything outside the two markers (MIOpen solver search during warm-up, the roofline leg, the CPU baseline) is dropped.
"""
import csv
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    rows = list(csv.DictReader(open(src)))
    name_key = "Kernel_Name"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "trace_marker_kernel" in r[name_key]]
    if not marks:      # (traces taken before round 4 were marked with a cumulative-sum kernel)
        marks = [i for i, r in enumerate(rows) if "scan" in r[name_key].lower() or "cumsum" in r[name_key].lower()]
    if len(marks) < 2:
        raise SystemExit(f"expected two marker kernels, found {len(marks)}")
    # a cumsum may be several kernels: region = after the last kernel of the first group .. before the first of the last
    first_group_end = marks[0]
    while first_group_end + 1 in marks:
        first_group_end += 1
    last_group_start = marks[-1]
    while last_group_start - 1 in marks:
        last_group_start -= 1
    region = rows[first_group_end + 1:last_group_start]
    t0, t1 = int(region[0]["Start_Timestamp"]), int(region[-1]["End_Timestamp"])
    agg = defaultdict(lambda: [0, 0])
    for r in region:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r[name_key]][0] += 1
        agg[r[name_key]][1] += d
    busy = sum(v[1] for v in agg.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfKernelTime", "CallsPerStep", "UsPerStep"])
        for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, n, tot, round(tot / n, 1), round(100.0 * tot / busy, 2), round(n / steps, 2) if steps else "",
                        round(tot / 1e3 / steps, 2) if steps else ""])
        w.writerow(["#region_wall_ns", t1 - t0, "kernel_busy_ns", busy, "kernels", len(region), "steps", steps or ""])
    print(f"timed region: wall {1e-6 * (t1 - t0):.2f} ms, kernel busy {1e-6 * busy:.2f} ms, {len(region)} kernel launches"
          + (f", per step: wall {1e-3 * (t1 - t0) / steps:.1f} us, busy {1e-3 * busy / steps:.1f} us, {len(region) / steps:.1f} launches" if steps else ""))
    for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{100.0 * tot / busy:5.1f}%  {tot / n / 1e3:9.2f} us x {n:6d}  {name[:130]}")


if __name__ == "__main__":
    main()
