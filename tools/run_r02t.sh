#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_pack_ops_gpu.py tests/test_occ_grid_gpu.py tests/test_ray_query_gpu.py tests/test_neus_query_gpu.py tests/test_forest_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r02t_pytest.log
cat gpurun_out/r02t_pytest.log
python - <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
for side in (64, 128, 512):
    r = bench.march_composite_rate(dev, iters=20, side=side, cpu_seconds=0.0)
    print(side * side, r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
PY
