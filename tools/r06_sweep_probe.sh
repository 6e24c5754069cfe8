#!/bin/bash
# Round 6, GPU box: (a) per-wave / per-SIMD timeline of the shipped MFMA sweep and of the round-6 configurations, (b) their timings.
#   tools/r06_sweep_probe.sh  (needs `make -C deep-video-mvs_amd/csrc trace tuning`)
set -x
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
T="timeout 600"
$T python tools/sweep_mfma_trace.py --lines 0,118 --variant 6 > $out/r06_sweep_mfma_v3_timeline.txt 2>&1
for v in 224 225 226 228; do
  $T python tools/sweep_mfma_trace.py --lines 0,118 --variant $v > $out/r06_sweep_mfma_trace_variant$v.txt 2>&1
done
$T python tools/cv_microbench.py --lib tuning --layouts nhwc --lines 0,8,16,24,118,170,202 --variants 6,138,224,225,226,227,228,229,230,231,232,234 --out $out/r06_sweep_mfma_tuning1.json > $out/r06_sweep_mfma_tuning1.txt 2>&1
tail -30 $out/r06_sweep_mfma_tuning1.txt
grep -A3 "per SIMD" $out/r06_sweep_mfma_v3_timeline.txt | head -20
