#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_variants.sh r02v_xcd "NR3D_PAIR_DEBUG=0,4,0,4,0,4"
