for i in 1 2 3; do python3 -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-update-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms_median'], d['step_ms_max'], d['roofline']['frac'], d['cpu_baseline']['value'])"; done
