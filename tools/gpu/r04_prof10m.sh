cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# what bench.py asks the HIP runtime for on this workload (pysgmcmc_amd.prefer_plain_graph_launch) -- exported here because under
# rocprofv3 the runtime initialises before python runs
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=gpurun_out/r04p2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench10m -o b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-only > $O/prof_bench10m.json 2> $O/prof_bench10m.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
grep "stream_quads" $O/prof_bench10m/b_kernel_stats.csv | cut -c1-200
