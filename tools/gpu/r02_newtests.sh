mkdir -p gpurun_out/r02
python3 -m pytest tests/test_config4_gpu.py tests/test_diagnostics_gpu.py tests/test_relativistic_momentum.py tests/test_hip_parity.py::test_rhat_pack_finish -m gpu -x -q > gpurun_out/r02/pytest_new.txt 2>&1
tail -40 gpurun_out/r02/pytest_new.txt
