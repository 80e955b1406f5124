cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02s; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms_median'], d['roofline']['frac'])"; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('drv', d['value'], d['ms_per_step'], d['step_ms_median'], d['roofline']['frac'])"
