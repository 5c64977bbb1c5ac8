#!/usr/bin/env python
"""tools/exp_buckets.py -- cost of computing dL/dparam in level buckets (headline config, 2^20 points): HIP-event times of
the one-call backward, the bucketed one and each bucket alone.   python tools/exp_buckets.py [lo ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg

cfg = gen_ngp_cfg()
m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
dev = torch.device("cuda", 0)
N = 1 << 20
g = torch.Generator().manual_seed(1)
p = torch.empty(m.n_params).uniform_(-1e-4, 1e-4, generator=g).to(dev)
x = torch.rand(N, 3, generator=g).to(dev)
gy = (torch.randn(N, m.n_encoded_dims, generator=g) / 1e4).to(dev)
y, j = _lotd.lod_fwd(m, x, p, need_input_grad=True)


def timed(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / iters, 4)


out = {"full": timed(lambda: _lotd.lod_bwd(m, gy, x, p, j, need_input_grad=True, need_param_grad=True))}
cuts = [int(a) for a in sys.argv[1:]] or [6]
for c in cuts:
    bk = [(c, 15), (0, c - 1)]
    out[f"buckets@{c}"] = timed(lambda: _lotd.lod_bwd(m, gy, x, p, j, need_input_grad=True, need_param_grad=True, level_buckets=bk))
    for b in bk:
        out[f"only{b}"] = timed(lambda: _lotd.lod_bwd(m, gy, x, p, j, need_input_grad=True, need_param_grad=True, level_buckets=[b]))
out["dx_only"] = timed(lambda: _lotd.lod_bwd(m, gy, x, p, j, need_input_grad=True, need_param_grad=False))
print(json.dumps(out))
