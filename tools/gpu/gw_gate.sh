#!/bin/bash
# round 6 gate: the weight-gradient harness alone and under rocprofv3 (kernel durations without launch gaps; shape 0 = the two batched 2048 x 2048 products)
O=gpurun_out/gw_gate
mkdir -p $O
tools/gpu/bnn_gw_bf16x3 > $O/harness.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats$s -o s -- $GRAFT_REPO_ROOT/tools/gpu/bnn_gw_bf16x3 $s > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for s in 0 1 2; do echo "== shape $s"; f=$(find $O/stats$s -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6,7 "$f" | grep -v -E "fill|ref_f64" | head -30; done > $O/kernel_stats.txt
cat $O/harness.txt | grep -E "^gW|time per|probes|16-byte|err"
cat $O/kernel_stats.txt
