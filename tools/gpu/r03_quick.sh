set -x
mkdir -p gpurun_out/r03
python -m pytest tests/test_step_opts_gpu.py tests/test_hip_parity.py tests/test_samplers_gpu.py -m gpu -q -x 2>&1 | tail -4
python3 tools/stats_variant_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/stats_variant_cost_b.txt
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03/bench_quick.json 2> gpurun_out/r03/bench_quick.err; tail -2 gpurun_out/r03/bench_quick.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms_median'], d['roofline']['frac'], d['roofline']['us_per_step_mean'], d['roofline'].get('with_fused_moments'), d['roofline_unoverlapped']['frac'], d['step_breakdown_us'])
for k,v in d['roofline_hbm_resident']['kernels'].items(): print(k, v['us_per_launch_mean'], v['frac'])
print(d['update_only'])
PY
