# Evidence run of the forward-epilogue experiment. Through gpurun: bash tools/gpu/fwd_probe.sh
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fwdprobe; rm -rf $O; mkdir -p $O
python3 tools/fwd_fused_probe.py 2>&1 | grep -v amdgpu.ids > $O/default.txt
for P in 10 8 1 2; do echo "== BNN_DENSE_TANH_PROBE=$P"; BNN_DENSE_TANH_PROBE=$P python3 tools/fwd_fused_probe.py 2>&1 | grep "^same\|^chain"; done > $O/variants.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 tools/fwd_fused_probe.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/default.txt $O/variants.txt; head -8 $O/stats/s_kernel_stats.csv | cut -c1-160
