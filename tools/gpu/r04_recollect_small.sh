# The pieces of the round-4 evidence that run WITHOUT the plain-graph-launch runtime setting (configs[1] is host-bound through graphs): sinc line +
# its rocprofv3 stats, examples, pytest -m gpu. Through gpurun: bash tools/gpu/r04_recollect_small.sh
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p3; rm -rf $O; mkdir -p $O
python3 bench.py --workload sinc-bnn --steps 3000 --warmup 100 > $O/bench_sinc_bnn.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sinc -o b -- python3 bench.py --workload sinc-bnn --steps 3000 --warmup 100 --no-cpu-baseline > $O/prof_sinc.json 2> $O/prof_sinc.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for ex in examples/*.py; do echo "== $ex"; timeout 300 python3 $ex 2>&1 | tail -4; done > $O/examples.txt 2>&1
python3 -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_check.json 2>> $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04p3/bench_sinc_bnn.json').read().strip().splitlines()[-1]); print(d['value'], d['modes_samples_per_s'], d['cpu_baseline']['value'])
d=json.loads(open('gpurun_out/r04p3/bench_check.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['hip_runtime_env'], d['roofline']['traffic'])
PY
