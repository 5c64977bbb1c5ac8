python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r02k_pytest.log; tail -6 gpurun_out/r02k_pytest.log
bash tools/gpu_profile.sh r02k pmc > /dev/null 2>&1; head -16 gpurun_out/r02k/bench_kernel_stats.txt
bash tools/gpu_counters.sh r02k_ctr "k_fwd|k_pair_bin|k_pair_accum|k_contract" > /dev/null 2>&1
