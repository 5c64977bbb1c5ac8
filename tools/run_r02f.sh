python -m pytest tests/test_occ_grid_gpu.py tests/test_mlp_gpu.py tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02f_pytest.log; tail -15 gpurun_out/r02f_pytest.log
NR3D_PAIR_EPB_LOG2=12 python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -5
bash tools/gpu_variants.sh r02f_epb "NR3D_PAIR_EPB_LOG2=12,13" "NR3D_LOTD_ACC_UNITS=768,1024,1536,2048"
