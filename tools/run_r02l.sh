python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r02l_pytest.log; tail -12 gpurun_out/r02l_pytest.log
python tools/exp_half_path.py 2>&1 | tail -12
