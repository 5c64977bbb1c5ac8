python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02g_pytest.log; tail -15 gpurun_out/r02g_pytest.log
bash tools/gpu_variants.sh r02g_fix "NR3D_PAIR_FIXED=0,1" "NR3D_PAIR_UNROLL=4,8" "NR3D_PAIR_UNITS=1024,1536"
