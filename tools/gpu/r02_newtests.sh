mkdir -p gpurun_out/r02
python3 -m pytest tests/test_config4_gpu.py -m gpu -x -q --durations=5 > gpurun_out/r02/pytest_new.txt 2>&1
tail -30 gpurun_out/r02/pytest_new.txt
