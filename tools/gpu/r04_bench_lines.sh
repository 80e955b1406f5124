# the four headline bench lines again (after a change that touches only the post-run legs / labels)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# what bench.py asks the HIP runtime for on this workload (pysgmcmc_amd.prefer_plain_graph_launch) -- exported here because under
# rocprofv3 the runtime initialises before python runs
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=gpurun_out/r04p; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_a.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_b.json 2>> $O/bench.err
python3 bench.py > $O/bench_default.json 2>> $O/bench.err
python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-update-only > $O/bench_2000.json 2>> $O/bench.err
python3 bench.py --chains-per-gpu 2 --steps 200 --warmup 20 --no-cpu-baseline --no-update-only > $O/bench_2chains_per_gpu.json 2>> $O/bench.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04p/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('step_ms_median'), d['roofline']['frac'], {k:v for k,v in (d.get('step_breakdown_us') or {}).items() if k!='note'}, d.get('chains_per_gpu',{}).get('two_chains_samples_per_s'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/bench.err
