"""tools/exp_fused_query.py -- the no-grad pruning query of the full loop alone: fused encode + decode kernel against the two kernels
(6.9 M ray-coherent marched samples of the 262 144-ray pass).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
from nr3d_lib_amd.models.grid_encodings.lotd import lotd as L

dev = torch.device("cuda:0")
model, n, fwd_bwd = bench._full_loop_setup(dev, 512)
rays_o, rays_d, near, far = __import__("demo_field").pinhole_rays(512, dev)
m = model.accel.ray_march(rays_o, rays_d, near, far)
x = m.samples
print("samples", x.shape[0])
for fuse in (True, False):
    L.FUSE_DECODED = fuse
    with torch.no_grad():
        for _ in range(3):
            s = model.query_density(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            s = model.query_density(x)
        torch.cuda.synchronize()
    print("fused" if fuse else "two kernels", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms", float(s.double().sum()))
