#!/bin/bash
# Round-6 A/B of the engine's third ("auxiliary") stream: bench.py with DVMVS_AUX_STREAM = 0 / warp / heads / 1, twice each.
b() { timeout 300 python bench.py "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_work_ms_per_step'), d['rel_l1']['teacher_forced_max'])"; }
for v in 0 warp heads 1; do echo "== DVMVS_AUX_STREAM=$v"; DVMVS_AUX_STREAM=$v b; DVMVS_AUX_STREAM=$v b; done
