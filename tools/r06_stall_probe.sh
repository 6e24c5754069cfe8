for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-roofline-leg --no-rel-l1 --no-secondary --no-cpu-baseline --sequences-per-gpu 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['device_ms_between_step_ends']; print(round(d['value'],1), 'max gap', max(g), 'at', g.index(max(g)), 'host', round(d['host_ms_per_step'],3), round(d['host_work_ms_per_step'],3))"
done
