python -m pytest tests/test_forest_gpu.py tests/test_lotd_gpu.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r02m_pytest.log; tail -12 gpurun_out/r02m_pytest.log
