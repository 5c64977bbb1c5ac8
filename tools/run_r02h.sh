python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_reference_vectors_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02h_pytest.log; tail -8 gpurun_out/r02h_pytest.log
bash tools/gpu_variants.sh r02h_pl "NR3D_LOTD_FWD_PAIRLANE=0,1" "NR3D_LOTD_LDS_STAGE=0,1" "NR3D_PAIR_UNROLL=4"
