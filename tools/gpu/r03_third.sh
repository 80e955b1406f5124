#!/bin/bash
# round 3, third GPU session: new tests (toy chains, full-protocol ESS pin, configs[0]), the whole suite, then the
# ADAPT sweep and the statistics-variant cost
set -x
mkdir -p gpurun_out/r03
python -m pytest tests/test_builtin_target_chains_gpu.py tests/test_reference_outputs_gpu.py tests/test_step_opts_gpu.py -m gpu -q --durations=8 > gpurun_out/r03/pytest_new.txt 2>&1
tail -25 gpurun_out/r03/pytest_new.txt
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_all.txt 2>&1
tail -8 gpurun_out/r03/pytest_gpu_all.txt
python tools/stats_variant_cost.py > gpurun_out/r03/stats_variant_cost.txt 2>&1
cat gpurun_out/r03/stats_variant_cost.txt | grep -v amdgpu.ids
python tools/adapt_sweep.py > gpurun_out/r03/adapt_sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/adapt_sweep.txt | sort -t'(' -k1,1 | head -100
