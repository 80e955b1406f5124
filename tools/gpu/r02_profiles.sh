# Round-2 profile refresh on the final kernels. Run through gpurun from the repo root.
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
# (a) kernel-trace stats of the bench commands (program directly after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench10m -o b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/prof_bench10m.json 2> $O/prof_bench10m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench50m_sgld -o b -- python3 bench.py --workload bnn50m-sgld --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_bench50m_sgld.json 2> $O/prof_bench50m_sgld.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench50m_rsghmc -o b -- python3 bench.py --workload bnn50m-rsghmc --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_bench50m_rsghmc.json 2> $O/prof_bench50m_rsghmc.err
# (b) cold launches of every update kernel at both sizes: kernel stats + the two PMC passes
for N in 10002434 49826818; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe_${N}_stats -o s -- python3 tools/pmc_probe.py $N 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${N}_fetch -o f -- python3 tools/pmc_probe.py $N 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${N}_write -o w -- python3 tools/pmc_probe.py $N 3 > /dev/null 2>&1
done
# keep only the small summaries (the per-dispatch traces are MBs)
find $O -name "*kernel_trace.csv" -path "*prof_bench*" -delete
find $O -name "*kernel_trace.csv" -path "*probe_*" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/*/ | head -80
tail -2 $O/prof_bench10m.err
