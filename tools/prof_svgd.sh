# per-kernel rocprofv3 stats of the SVGD step for a few (particles x parameters) shapes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for shape in 50x2 8x10002434 32x10002434 64x10002434; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/svgd_prof_$shape -o svgd -- python3 tools/svgd_rate.py $shape > gpurun_out/svgd_prof_$shape.log 2>&1
  f=$(find gpurun_out/svgd_prof_$shape -name "*kernel_stats.csv" | head -1)
  echo "== $shape"
  if [ -n "$f" ]; then head -9 "$f" | cut -c1-220; fi
done
