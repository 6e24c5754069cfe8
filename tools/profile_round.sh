#!/bin/bash
# Round profile on the GPU box (run from the repo root):   tools/profile_round.sh gpurun_out/prof_r02
#  1. rocprofv3 --kernel-trace --stats of the default bench command (timed region bracketed by marker kernels)
#  2. rocprofv3 --pmc passes (SQ mix, LDS conflicts, FETCH_SIZE, WRITE_SIZE -- each counter group in its own run, never mixed
#     with tracing) of the sweep + spill kernels on three keyframe geometries: easy (153), median (118), worst (165)
# Copy what should be judged from <out> into profiles/ afterwards (tools/collect_profiles.py does that).
out="${1:-gpurun_out/prof_r03}"
mkdir -p "$out"
export TMPDIR=/tmp
root="$(pwd)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$root/$out/bench_trace" --output-format csv -- \
   python "$root/bench.py" --steps 200 --warmup 30 --mark-region --no-cpu-baseline --no-rel-l1 --sequences-per-gpu 0 \
   > "$root/$out/bench_under_rocprof.json" 2> "$root/$out/bench_under_rocprof.err")
trace=$(ls "$out"/bench_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
stats=$(ls "$out"/bench_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$trace" ] && python tools/summarize_trace.py "$trace" "$out/bench_timed_region.csv" 200 > "$out/bench_timed_region.txt" 2>&1
[ -n "$stats" ] && cp "$stats" "$out/bench_kernel_stats_whole_run.csv"
rm -rf "$out/bench_trace"
# all counter groups on the easy line; the HBM-traffic passes (FETCH_SIZE, WRITE_SIZE) and the kernel trace on all three
for line in 153 118 165; do
  if [ "$line" = 153 ]; then only=""; else only="3 4"; fi
  PMC_ONLY="$only" tools/pmc_sweep.sh "$out/pmc_line$line" --lines=$line --variants 2 --reps 2
  rm -rf "$out/pmc_line$line"/pass*/ "$out/pmc_line$line"/trace
done
