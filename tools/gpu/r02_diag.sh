show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d['step_ms_median'], d['step_ms_max'], d['host_enqueue_ms'], d['roofline']['frac'])"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | show drv
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | show drv
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | show drv_full
python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show default
python3 bench.py --gpus 1 --steps 2000 --warmup 20 --no-cpu-baseline --no-update-only 2>/dev/null | show long2000_d64
