# HBM traffic of the SVGD kernels (separate FETCH_SIZE / WRITE_SIZE passes, as for profiles/r01_pmc_traffic.md)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for shape in 16x10002434 64x2000000; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_svgd_${shape}_$c -o p -- python3 tools/svgd_rate.py $shape > gpurun_out/pmc_svgd_${shape}_$c.log 2>&1
  done
done
ls gpurun_out | grep pmc_svgd
