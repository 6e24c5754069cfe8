#!/bin/bash
# Round 6, GPU box: final A/B of the shipped MFMA sweep (variant 6: persistent, quarter tail, gather 256) against its parts switched off, and the
# bottleneck convolution's rolling weight prefetch (CH 0) against round 4's bursts.
set -x
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests/test_sweep_mfma_gpu.py tests/test_hip_parity.py tests/test_sweep_random_geometries_gpu.py tests/test_bottleneck_conv_gpu.py -m gpu -x -q -s) > $out/r06_pytest_gpu_5.txt 2>&1; tail -5 $out/r06_pytest_gpu_5.txt
timeout 900 python tools/cv_microbench.py --lib tuning --layouts nhwc --lines 0,8,16,24,60,97,100,118,150,170,180,202,250 --variants 6,246,248,7,auto --out $out/r06_sweep_mfma_tuning8.json > $out/r06_sweep_mfma_tuning8.txt 2>&1; tail -8 $out/r06_sweep_mfma_tuning8.txt
bash tools/bc_tuning_probe.sh > $out/r06_bottleneck_conv_prefetch.txt 2>&1; cat $out/r06_bottleneck_conv_prefetch.txt
