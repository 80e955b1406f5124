# SQ counters of the SVGD kernels for one shape (default 64 x 10 M): where the wave cycles go
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
shape=${1:-64x10002434}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmc_svgd_$shape -o s -- python3 tools/svgd_rate.py $shape > gpurun_out/pmc_svgd_$shape.log 2>&1
tail -3 gpurun_out/pmc_svgd_$shape.log
ls gpurun_out/pmc_svgd_$shape
