"""tools/exp_sorted_step.py -- round 6, review item 9: what would a per-step Morton sort of the 2^20 points buy the headline step?
Upper bound first: the UNCHANGED kernels on the same points given in Morton order (the permutation passes a real implementation needs
are then priced separately: sort + x gather + un-permuting y (128 MB each way) + dL_dx).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg

dev = torch.device("cuda:0")
cfg = gen_ngp_cfg()
meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
N = 1 << 20
gen = torch.Generator().manual_seed(42)
params = torch.empty(meta.n_params).uniform_(-1e-4, 1e-4, generator=gen).to(dev)
x = torch.rand(N, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
g = (torch.randn(N, meta.n_encoded_dims, generator=gen) / 1e4).to(dev)


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    names = ("lotd_fwd", "lotd_fwd_lds", "lotd_contract_dx", "lotd_bin", "lotd_accum", "lotd_direct")
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    for k in names:
        H.prof_read(k)
    H.prof_enable(*names)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    H.prof_enable()
    us = {}
    for k in names:
        tot, cnt = H.prof_read(k)
        us[k] = round(tot / 5 * 1e3, 1)
    return round(ms, 4), us


def step(xx, gg):
    y, j = _lotd.lod_fwd(meta, xx, params, need_input_grad=True)
    return _lotd.lod_bwd(meta, gg, xx, params, j, need_input_grad=True, need_param_grad=True)


print("random order        ", *timed(lambda: step(x, g)))
for bits in (5, 6, 7, 8):
    order = H.spatial_order(x.contiguous(), bits).long()
    xs, gs = x[order].contiguous(), g[order].contiguous()
    print(f"Morton {bits} bits/dim   ", *timed(lambda: step(xs, gs)))
# the passes a real implementation adds
order = H.spatial_order(x.contiguous(), 7)
y = torch.empty(N, 32, device=dev)


def passes():
    o = H.spatial_order(x, 7)                        # keys + sort
    ol = o.long()
    xs = x[ol]                                       # x gather
    ys = y[ol]                                       # stands for un-permuting y (128 MB read + 128 MB write)
    gs = g[ol]                                       # dL_dy into sorted order (or gathered rows inside the kernels)
    return xs, ys, gs
print("permutation passes (sort, x, y, dL_dy as torch gathers): ms", timed(passes)[0])
print("sort alone: ms", timed(lambda: H.spatial_order(x, 7))[0])
