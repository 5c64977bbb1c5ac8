#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_reference_vectors_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02m_pytest.log
cat gpurun_out/r02m_pytest.log
bash tools/gpu_variants.sh r02m_all "NR3D_PAIR_ALL=0,1,0,1"
python bench.py --log2-points 24 --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1
NR3D_PAIR_ALL=0 python bench.py --log2-points 24 --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1
