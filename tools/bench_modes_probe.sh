#!/bin/bash
# frames/s of the headline configuration with MIOpen's solver search (cudnn.benchmark = True) and with its
# immediate mode (DVMVS_BENCH_CUDNN_BENCHMARK=0, bench.py's default)
cd "$(dirname "$0")/.."
for mode in 1 0; do
  DVMVS_BENCH_CUDNN_BENCHMARK=$mode python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-rel-l1 --no-roofline-leg --sequences-per-gpu 0 \
    | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('cudnn.benchmark=$mode', round(r['value'],1), 'frames/s', round(r['ms_per_step'],4), 'ms', r['config']['conv_epilogues_inside_miopen'])"
done
