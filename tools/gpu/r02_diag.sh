show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['value'], d['ms_per_step'], d['step_ms_median'], d['step_ms_max'], r['frac'], r['us_per_launch_mean'], (r.get('bracket') or {}).get('us_per_launch_mean'))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | show kev_only
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | show kev_only
BENCH_BRACKET=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | show kev+bracket
python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show kev_only_200
