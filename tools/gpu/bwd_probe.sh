# Evidence run of the fused backward step (delta W^T + tanh' + column sums per row tile). Through gpurun: bash tools/gpu/bwd_probe.sh
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# what bench.py asks the HIP runtime for on this workload (pysgmcmc_amd.prefer_plain_graph_launch) -- exported here because under
# rocprofv3 the runtime initialises before python runs
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=gpurun_out/bwdprobe; rm -rf $O; mkdir -p $O
python3 tools/bwd_fused_probe.py 2>&1 | grep -v amdgpu.ids > $O/default.txt
PROBE_SKIP_CHAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 tools/bwd_fused_probe.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/default.txt; head -8 $O/stats/s_kernel_stats.csv | cut -c1-200
