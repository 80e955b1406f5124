# Round-2 reproduction of the driver's bench command next to the long run.
mkdir -p gpurun_out/r02
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r02/pytest_gpu.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_20_5.json 2> gpurun_out/r02/bench_20_5.err
python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_200_20.json 2>> gpurun_out/r02/bench_20_5.err
python3 bench.py --gpus 1 --steps 20 --warmup 0 --no-cpu-baseline > gpurun_out/r02/bench_20_0.json 2>> gpurun_out/r02/bench_20_5.err
tail -3 gpurun_out/r02/bench_20_5.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('step_ms_median'), d.get('step_ms_max'), d['roofline']['frac'], d.get('roofline_hbm_resident',{}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
