#!/bin/bash
# Round profile on the GPU box (run from the repo root):   tools/profile_round.sh gpurun_out/prof_r04
#  1. rocprofv3 --kernel-trace --stats of the DRIVER's bench command (--steps 20 --warmup 5; timed region bracketed by marker kernels),
#     once as benchmarked (look-ahead 1: the next keyframe's feature extraction on a second stream, so traced kernel durations include
#     the slowdown of running next to another stream's kernels) and once with --lookahead 0 (one stream)
#  2. the same trace of the sweep alone on the index lines of the timed steps, each with the kernel (dvmvs_sweep_plan6: MFMA sweep or a tiled
#     configuration + work list) and the channels-last measurement maps the engine uses
#     (this is what bench.py's roofline leg times with HIP events)
#  3. rocprofv3 --pmc passes (SQ mix, LDS conflicts, FETCH_SIZE, WRITE_SIZE -- each counter group in its own run, never mixed
#     with tracing) of the sweep + second-pass kernels on three keyframe geometries: easy (153), median (118), hard (165)
#  4. kernel trace of the training step (bench.py --mode train)
# Copy what should be judged from <out> into profiles/ afterwards (tools/collect_profiles.py does that).
out="${1:-gpurun_out/prof_r04}"
steps="${STEPS:-20}"; warmup="${WARMUP:-5}"
mkdir -p "$out"
export TMPDIR=/tmp
root="$(pwd)"
for la in 1 0; do
  tag="lookahead$la"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$root/$out/bench_trace_$tag" --output-format csv -- \
     python "$root/bench.py" --steps $steps --warmup $warmup --lookahead $la --mark-region --no-cpu-baseline --no-rel-l1 --sequences-per-gpu 0 \
     > "$root/$out/bench_under_rocprof_$tag.json" 2> "$root/$out/bench_under_rocprof_$tag.err")
  trace=$(ls "$out"/bench_trace_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
  stats=$(ls "$out"/bench_trace_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$trace" ] && python tools/summarize_trace.py "$trace" "$out/bench_timed_region_$tag.csv" $steps > "$out/bench_timed_region_$tag.txt" 2>&1
  [ -n "$stats" ] && cp "$stats" "$out/bench_kernel_stats_whole_run_$tag.csv"
  rm -rf "$out/bench_trace_$tag"
done
first=$((2 + warmup)); last=$((first + steps - 1))
lines=$(seq -s, $first $last)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$root/$out/sweep_trace" --output-format csv -- \
   python "$root/tools/cv_microbench.py" --lines $lines --variants engine --layouts nhwc --reps 10 > "$root/$out/sweep_timed_lines.log" 2>&1)
cp "$out"/sweep_trace/*/*kernel_stats.csv "$out/sweep_timed_lines_kernel_stats.csv" 2>/dev/null
rm -rf "$out/sweep_trace"
# all counter groups on the easy line; the HBM-traffic passes (FETCH_SIZE, WRITE_SIZE) and the kernel trace on all three
for line in 153 118 165; do
  if [ "$line" = 153 ]; then only=""; else only="3 4"; fi
  PMC_ONLY="$only" tools/pmc_sweep.sh "$out/pmc_line$line" --lines=$line --variants engine --layouts nhwc --reps 2
  rm -rf "$out/pmc_line$line"/pass*/ "$out/pmc_line$line"/trace
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$root/$out/train_trace" --output-format csv -- \
   python "$root/bench.py" --mode train --steps 3 --warmup 1 --mark-region > "$root/$out/train_under_rocprof.json" 2> "$root/$out/train_under_rocprof.err")
trace=$(ls "$out"/train_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$trace" ] && python tools/summarize_trace.py "$trace" "$out/train_timed_region.csv" 3 > "$out/train_timed_region.txt" 2>&1
rm -rf "$out/train_trace"
