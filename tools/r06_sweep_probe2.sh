#!/bin/bash
# Round 6, GPU box: persistent form of the MFMA sweep (tuning variants 235 / 236) against the shipped kernel: timelines + timings.
set -x
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
T="timeout 600"
for v in 235 236; do
  $T python tools/sweep_mfma_trace.py --lines 0,118,170 --variant $v --waves 8192 > $out/r06_sweep_mfma_trace_stage2_variant$v.txt 2>&1
done
$T python tools/cv_microbench.py --lib tuning --layouts nhwc --lines 0,8,16,24,60,100,118,150,170,202,250 --variants 6,227,235,236,engine --out $out/r06_sweep_mfma_tuning5.json > $out/r06_sweep_mfma_tuning5.txt 2>&1
tail -12 $out/r06_sweep_mfma_tuning5.txt
grep -B2 -A12 "per SIMD" $out/r06_sweep_mfma_trace_variant236.txt | head -60
