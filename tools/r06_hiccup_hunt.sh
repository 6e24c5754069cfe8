#!/bin/bash
# Hunts the rare 4 - 6 ms step of the driver's 20-step bench command: N quick runs with the per-step host clock on; a run with a slow step leaves
# gpurun_out/hunt/trace_<i>.json (bench.py: DVMVS_BENCH_STEP_TRACE).   tools/r06_hiccup_hunt.sh [N, default 40]
n="${1:-40}"; out=gpurun_out/hunt; mkdir -p "$out"
for i in $(seq 1 "$n"); do
  DVMVS_BENCH_STEP_TRACE="$out/trace_$i.json" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rel-l1 --sequences-per-gpu 0 --no-secondary 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['device_ms_between_step_ends']; print($i, round(d['value'],1), 'max gap', max(g), 'at', g.index(max(g)))"
done > "$out/runs.txt"
cat "$out/runs.txt"; ls "$out"
