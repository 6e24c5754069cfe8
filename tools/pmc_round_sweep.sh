#!/bin/bash
# The sweep-only part of tools/profile_round.sh (steps 2 + 3): kernel trace of the sweep alone on the timed lines and the PMC passes on the
# three keyframe geometries, into <out> (default gpurun_out/prof_r05) -- enough to refresh <round>_cost_volume_pmc.json after a kernel change.
out="${1:-gpurun_out/prof_r05}"; steps="${STEPS:-20}"; warmup="${WARMUP:-5}"
mkdir -p "$out"; export TMPDIR=/tmp; root="$(pwd)"
first=$((2 + warmup)); last=$((first + steps - 1)); lines=$(seq -s, $first $last)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$root/$out/sweep_trace" --output-format csv -- \
   python "$root/tools/cv_microbench.py" --lines $lines --variants engine --layouts nhwc --reps 10 > "$root/$out/sweep_timed_lines.log" 2>&1)
cp "$out"/sweep_trace/*/*kernel_stats.csv "$out/sweep_timed_lines_kernel_stats.csv" 2>/dev/null
rm -rf "$out/sweep_trace"
for line in 153 118 165; do
  if [ "$line" = 153 ]; then only=""; else only="3 4"; fi
  PMC_ONLY="$only" tools/pmc_sweep.sh "$out/pmc_line$line" --lines=$line --variants engine --layouts nhwc --reps 2
  rm -rf "$out/pmc_line$line"/pass*/ "$out/pmc_line$line"/trace
done
