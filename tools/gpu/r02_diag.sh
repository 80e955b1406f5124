show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d['step_ms_median'])"; }
python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show tune30_20
BENCH_TUNE_MS=200 BENCH_TUNE_ITERS=100 python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show tune200_100
BENCH_TUNE_MS=200 BENCH_TUNE_ITERS=100 python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show tune200_100
python3 bench.py --no-cpu-baseline --no-update-only 2>/dev/null | show tune30_20
python3 bench.py --no-cpu-baseline --no-update-only --no-gemm-tuning 2>/dev/null | show notune
