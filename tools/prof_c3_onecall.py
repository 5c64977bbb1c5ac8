"""tools/prof_c3_onecall.py -- where one iteration of configs[2] on the one-call path spends its host time (run on the GPU box):
enqueue of the forward call, enqueue of the backward, wait for the totals, views."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nr3d_lib_amd.bindings import _occ_grid

dev = torch.device("cuda:0")
for occ in ("random", "shell"):
    grid_c, o_c, d_c, near_c, far_c, roi_c, step = bench._c3_scene(64, occ)
    grid, o, d, near, far, roi = (t.to(dev) for t in (grid_c, o_c, d_c, near_c, far_c, roi_c))
    n = 4096
    m = _occ_grid.ray_marching_finished(o, d, near, far, roi, grid, 0, step, 1e10, 0.0, 512, True)
    S = m["t_starts"].shape[0]
    sigma, rgb = 10 * torch.rand(S, device=dev), torch.rand(S, 3, device=dev)
    g = [torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, 3, device=dev)]
    acc = [0.0] * 5
    iters = 200
    for it in range(iters + 20):
        if it == 20:
            acc = [0.0] * 5
            torch.cuda.synchronize(); T0 = time.perf_counter()
        t0 = time.perf_counter()
        mc = _occ_grid.ray_marching_composite(o, d, near, far, roi, grid, 0, step, 1e10, 0.0, 512, sigma, rgb, 1e-4, 0.0, True)
        t1 = time.perf_counter()
        mc.backward(*g)
        t2 = time.perf_counter()
        mc.totals()
        t3 = time.perf_counter()
        a, b = mc.view("mask"), mc.grads()[0]
        t4 = time.perf_counter()
        for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += v
    torch.cuda.synchronize(); T1 = time.perf_counter()
    print(occ, "S", S, "us/iter: fwd enqueue %.1f  bwd enqueue %.1f  wait %.1f  views %.1f  | total %.1f" %
          tuple([v / iters * 1e6 for v in acc[:4]] + [(T1 - T0) / iters * 1e6]))
