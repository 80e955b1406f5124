# One round's evidence on the final kernels. Through gpurun from the repo root: R=r06 bash tools/gpu/profiles.sh ; then, here,
# bash tools/collect_profiles.sh r06 copies the summaries from gpurun_out/ into profiles/ (tracked).
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=${R:-r06}; O=gpurun_out/${R}p; rm -rf $O; mkdir -p $O
# (1) FIRST the PMC passes (cold launches of every update kernel at both sizes) so that the bench lines below can read the traffic
# table of THIS build (keyed on the kernel sources' hash): kernel stats + the two PMC passes, separate, as the guide prescribes
python3 -c "import bench; print(bench.kernel_source_hash())" > $O/kernel_source_hash.txt
for N in 10002434 49826818; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe_${N}_stats -o s -- python3 tools/pmc_probe.py $N 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${N}_fetch -o f -- python3 tools/pmc_probe.py $N 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${N}_write -o w -- python3 tools/pmc_probe.py $N 3 > /dev/null 2>&1
done
PMC_KERNEL_SOURCE_HASH=$(cat $O/kernel_source_hash.txt) python3 tools/pmc_traffic.py profiles/${R}_pmc_traffic $O/pmc_10002434 $O/pmc_49826818 > $O/pmc_traffic.log 2>&1
cp profiles/${R}_pmc_traffic.json $O/pmc_traffic.json; cp profiles/${R}_pmc_traffic.md $O/pmc_traffic.md
# (2) the driver's command, twice, the long forms, configs[4]'s and configs[1]'s workloads
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_a.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_b.json 2>> $O/bench.err
python3 bench.py > $O/bench_default.json 2>> $O/bench.err
python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-update-only > $O/bench_2000.json 2>> $O/bench.err
python3 bench.py --chains-per-gpu 2 --steps 200 --warmup 20 --no-cpu-baseline --no-update-only > $O/bench_2chains_per_gpu.json 2>> $O/bench.err
python3 bench.py --workload bnn50m-sgld --steps 100 --warmup 10 > $O/bench_50m_sgld.json 2>> $O/bench.err
python3 bench.py --workload bnn50m-rsghmc --steps 100 --warmup 10 > $O/bench_50m_rsghmc.json 2>> $O/bench.err
python3 bench.py --workload sinc-bnn --steps 3000 --warmup 100 > $O/bench_sinc_bnn.json 2>> $O/bench.err
# the reference's default dtype (pysgmcmc/samplers/base_classes.py:25): the f64 chain's line
python3 bench.py --dtype f64 --steps 100 --warmup 10 > $O/bench_10m_f64.json 2>> $O/bench.err
# N > 1 started WITHOUT a launcher (all ranks on this box's one GPU over gloo: the code path, not xGMI timings)
BENCH_PRIME_STEADY=60 python3 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --all-ranks-on-gpu0 --no-update-only > $O/bench_selflaunch_n2_gloo.json 2>> $O/bench.err
BENCH_PRIME_STEADY=60 python3 bench.py --gpus 8 --steps 20 --warmup 5 --backend gloo --all-ranks-on-gpu0 --no-update-only > $O/bench_selflaunch_n8_gloo.json 2>> $O/bench.err
# (3) kernel-trace stats of the bench commands (program directly after --). Under rocprofv3 the HIP runtime initialises before python
# runs, so what pysgmcmc_amd.configure_for_device_bound_chains() asks of it is exported here
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench10m -o b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-only --no-product-defaults > $O/prof_bench10m.json 2> $O/prof_bench10m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench50m_rsghmc -o b -- python3 bench.py --workload bnn50m-rsghmc --steps 100 --warmup 10 --no-cpu-baseline --no-update-only --no-product-defaults > $O/prof_bench50m_rsghmc.json 2> $O/prof_bench50m_rsghmc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench50m_sgld -o b -- python3 bench.py --workload bnn50m-sgld --steps 100 --warmup 10 --no-cpu-baseline --no-update-only --no-product-defaults > $O/prof_bench50m_sgld.json 2> $O/prof_bench50m_sgld.err
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
# per-dispatch timelines of the 10 M-parameter step, f32 and f64
STEPTRACE_OUT=$O/steptrace bash tools/gpu/step_trace.sh > $O/step_timeline.txt 2>&1
STEPTRACE_ARGS="--dtype f64" STEPTRACE_OUT=$O/steptrace_f64 bash tools/gpu/step_trace.sh > $O/step_timeline_f64.txt 2>&1
cd /tmp 2>/dev/null; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sinc -o b -- python3 bench.py --workload sinc-bnn --steps 3000 --warmup 100 --no-cpu-baseline > $O/prof_sinc.json 2> $O/prof_sinc.err
# keep only the small summaries (the per-dispatch traces are MBs)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
for ex in examples/*.py; do echo "== $ex"; timeout 300 python3 $ex 2>&1 | tail -4; done > $O/examples.txt 2>&1
python3 -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('step_ms_median'), d['roofline']['frac'], d['roofline'].get('traffic'), d.get('roofline_hbm_resident',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('value_product_defaults') or {}).get('value'), d.get('value_ex_exchange'), (d.get('rccl') or {}).get('exposed_ms'), d.get('modes_samples_per_s'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $O/bench.err
