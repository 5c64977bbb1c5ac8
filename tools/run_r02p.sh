#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
NR3D_MARCH_VARIANT=1 timeout 900 python -m pytest tests/test_occ_grid_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02p_pytest.log
cat gpurun_out/r02p_pytest.log
for g in 16 32 64; do for v in 0 1 2 3; do
NR3D_MARCH_VARIANT=$v NR3D_MARCH_GROUP=$g python - <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
r = bench.march_composite_rate(dev, iters=20, side=64, cpu_seconds=0.0)
print(os.environ["NR3D_MARCH_GROUP"], os.environ["NR3D_MARCH_VARIANT"], r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
# a coherent scene: solid ball of radius 0.6 in the same 128^3 grid, same rays
from nr3d_lib_amd.bindings import _occ_grid
from nr3d_lib_amd import _hip as H
grid, o, d, near, far, roi, step = bench._c3_scene(64)
ax = torch.linspace(-1, 1, 128)
X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
ball = ((X * X + Y * Y + Z * Z) < 0.36)
args = [t.to(dev) for t in (o, d, near, far, roi)]
gb = ball.to(dev)
for k in range(3):
    out = _occ_grid.ray_marching(*args, gb, 0, step, 1e10, 0.0, 512, True)
torch.cuda.synchronize()
H.prof_enable("march"); 
for k in range(20):
    out = _occ_grid.ray_marching(*args, gb, 0, step, 1e10, 0.0, 512, True)
torch.cuda.synchronize()
ms, n = H.prof_read("march"); H.prof_enable()
print("  ball scene: march us/iter", round(ms * 1e3 / 20, 2), "samples", int(out[1].shape[0]))
PY
done; done
