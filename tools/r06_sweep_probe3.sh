#!/bin/bash
# Round 6, GPU box: gather passes in the persistent MFMA sweep (tuning variants 237..242) against variants 6 / 7 and the engine's choice.
set -x
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
T="timeout 900"
$T python tools/cv_microbench.py --lib tuning --layouts nhwc --lines 0,8,16,24,60,97,100,118,150,170,180,192,202,250 --variants 7,6,237,238,239,240,242,engine --out $out/r06_sweep_mfma_tuning6.json > $out/r06_sweep_mfma_tuning6.txt 2>&1
tail -12 $out/r06_sweep_mfma_tuning6.txt
$T python tools/sweep_mfma_trace.py --lines 0,118,170 --variant 238 --waves 8192 > $out/r06_sweep_mfma_trace_gather96.txt 2>&1
