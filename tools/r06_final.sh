#!/bin/bash
# Round-6 closing measurements on the GPU box (run from the repo root):  tools/r06_final.sh gpurun_out/final_r06
out="${1:-gpurun_out/final_r06}"; mkdir -p "$out"
( time timeout 2000 python -m pytest tests -m gpu -q ) > "$out/pytest_gpu.txt" 2>&1
tail -4 "$out/pytest_gpu.txt"
timeout 300 python bench.py > "$out/bench.json" 2> "$out/bench.err"
timeout 300 python bench.py --steps 100 --warmup 10 > "$out/bench_100_steps.json" 2> /dev/null
for i in 1 2 3 4 5; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_work_ms_per_step'))"; done > "$out/bench_five_runs.txt"
timeout 200 python tools/pointwise_probe.py --out "$out/pointwise_conv_layers.txt" > /dev/null 2>&1
DVMVS_HIP_LIB=deep-video-mvs_amd/lib/libdvmvs_hip_tuning.so timeout 100 python tools/bottleneck_layers_probe.py 2>&1 | grep -v amdgpu > "$out/bottleneck_layers.txt"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1
tools/r06_frame_timeline.sh "$out/ft" > /dev/null 2>&1
tools/profile_round.sh gpurun_out/prof_r06 > "$out/profile_round.log" 2>&1
python tools/collect_profiles.py gpurun_out/prof_r06 r06 > /dev/null 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver.json" 2> "$out/bench_driver.err"
for i in $(seq 1 12); do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rel-l1 --sequences-per-gpu 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['device_ms_between_step_ends']; print(round(d['value'],1), round(d['ms_per_step'],4), round(d['host_work_ms_per_step'],3), 'max gap', max(g), 'launches', d['launches_per_frame']['profiled'])"; done > "$out/bench_twelve_runs.txt"
