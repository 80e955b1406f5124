mkdir -p gpurun_out/r02
python3 tools/stats_variant_cost.py
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r02/pytest_gpu.txt
