#!/usr/bin/env python
"""tools/exp_c3_host.py -- where the host time of one configs[2] iteration (4096 rays) goes: wall per iteration with and
without the C-side event hooks, then a cProfile of 200 iterations."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.bindings import _occ_grid, _pack_ops

dev = torch.device("cuda", 0)
side = 64
grid_c, o_c, d_c, near_c, far_c, roi_c, step = bench._c3_scene(side)
grid, o, d, near, far, roi = (t.to(dev) for t in (grid_c, o_c, d_c, near_c, far_c, roi_c))
n = side * side
gen = torch.Generator(device="cpu").manual_seed(8)
state = {}


def one():
    m = _occ_grid.ray_marching_finished(o, d, near, far, roi, grid, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, 512, True)
    tmid, pil = m["t_starts"], m["pack_infos"]
    S = tmid.shape[0]
    if "sigma" not in state:
        state["sigma"] = (10.0 * torch.rand(S, generator=gen)).to(dev)
        state["rgb"] = torch.rand(S, 3, generator=gen).to(dev)
        state["g"] = [torch.randn(n, generator=gen).to(dev), torch.randn(n, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev)]
    alpha = _pack_ops.tau_to_alpha_forward(state["sigma"], m["deltas"])
    vw, mask, depth, rgb = _pack_ops.packed_composite_forward(alpha, tmid, state["rgb"], pil, m["ridx_hit"], n, 1e-4, 0.0, True, packs_tile=True)
    ga, gt, gc = _pack_ops.packed_composite_backward(alpha, vw, tmid, state["rgb"], pil, m["ridx_hit"], 1e-4, 0.0, True, mask, depth,
                                                     state["g"][0], state["g"][1], state["g"][2], None, packs_tile=True)
    return S


def wall(iters=200):
    for _ in range(5): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print("wall ms/iter, hooks off:", round(wall(), 4))
H.prof_enable("march", "composite_fwd", "composite_bwd")
print("wall ms/iter, hooks on :", round(wall(), 4))
H.prof_enable()
print("wall ms/iter, hooks off:", round(wall(), 4))
# host-only phases: time of each binding call as seen by the host (no sync between)
pr = cProfile.Profile()
pr.enable()
for _ in range(200): one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
