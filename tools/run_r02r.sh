#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pack_ops_gpu.py tests/test_occ_grid_gpu.py tests/test_ray_query_gpu.py tests/test_neus_query_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r02r_pytest.log
cat gpurun_out/r02r_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for sc in 1 0; do
NR3D_PACK_SCAN=$sc python - <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
for side in (64, 128, 512):
    r = bench.march_composite_rate(dev, iters=20, side=side, cpu_seconds=0.0)
    print("scan", os.environ["NR3D_PACK_SCAN"], side * side, r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
PY
done
NR3D_PACK_SCAN_MAX=1000000 python - <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
for side in (128, 512):
    r = bench.march_composite_rate(dev, iters=20, side=side, cpu_seconds=0.0)
    print("scan forced", side * side, r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
PY
