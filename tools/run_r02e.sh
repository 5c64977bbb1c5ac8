python -m pytest tests/test_pack_ops_gpu.py tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02e_pytest.log; tail -15 gpurun_out/r02e_pytest.log
bash tools/gpu_variants.sh r02e_lds "NR3D_LOTD_LDS_STAGE=0,1" "NR3D_LOTD_ACC_UNITS=768"
bash tools/gpu_variants.sh r02e_dbg "NR3D_PAIR_DEBUG=0,1,2,3" "NR3D_LOTD_ACC_UNITS=768"
NR3D_LOTD_ACC_UNITS=768 bash tools/gpu_profile.sh r02e > /dev/null 2>&1; head -14 gpurun_out/r02e/bench_kernel_stats.txt
NR3D_LOTD_ACC_UNITS=768 bash tools/gpu_counters.sh r02e_ctr "k_fwd|k_pair_bin|k_pair_accum" > /dev/null 2>&1
