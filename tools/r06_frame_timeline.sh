#!/bin/bash
# Kernel timeline of one frame of the bench, as benchmarked (look-ahead 1) and on one stream (look-ahead 0):  tools/r06_frame_timeline.sh gpurun_out/ft
out="${1:-gpurun_out/ft}"; mkdir -p "$out"; export TMPDIR=/tmp; root="$(pwd)"
for la in 1 0; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$root/$out/trace$la" -- \
     python "$root/bench.py" --steps 20 --warmup 5 --lookahead $la --mark-region --no-cpu-baseline --no-rel-l1 --sequences-per-gpu 0 \
     > "$root/$out/bench_la$la.json" 2> "$root/$out/bench_la$la.err")
  trace=$(ls "$out"/trace$la/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$trace" ] && python tools/frame_timeline.py "$trace" > "$out/frame_timeline_lookahead$la.txt" 2>&1
  [ -n "$trace" ] && python tools/summarize_trace.py "$trace" "$out/timed_region_lookahead$la.csv" 20 > "$out/timed_region_lookahead$la.txt" 2>&1
  rm -rf "$out/trace$la"
done
