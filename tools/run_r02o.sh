#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_forest_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02o_pytest.log
cat gpurun_out/r02o_pytest.log
python bench.py --steps 10 --warmup 3 --no-extra 2>/dev/null | tail -1 > gpurun_out/r02o_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r02o_bench.json')); print(d['ms_per_step'], d['kernel_ms'], d['cpu_baseline'])"
