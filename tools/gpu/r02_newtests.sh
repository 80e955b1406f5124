mkdir -p gpurun_out/r02
python3 -m pytest tests/test_diagnostics_gpu.py tests/test_hip_parity.py -m gpu -x -q --durations=5 -k "rccl or two_ranks or bench or rhat" > gpurun_out/r02/pytest_new.txt 2>&1
tail -30 gpurun_out/r02/pytest_new.txt
