#!/usr/bin/env python
"""tools/exp_second_order.py -- the three second-order outputs of configs[1]'s meta, one at a time (for rocprofv3
--kernel-trace: which kernels each pass launches and what they cost).   python tools/exp_second_order.py [log2_points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg

cfg = gen_ngp_cfg()
m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
N = 1 << int(sys.argv[1] if len(sys.argv) > 1 else 20)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
p = torch.empty(m.n_params).uniform_(-0.1, 0.1, generator=g).to(dev)
x = torch.rand(N, 3, generator=g).to(dev)
dy = torch.randn(N, m.n_encoded_dims, generator=g).to(dev)
v = torch.randn(N, 3, generator=g).to(dev)
_, j = _lotd.lod_fwd(m, x, p, need_input_grad=True)
for name, kw in (("ddLdy", dict(need_dLdinput_ddLdoutput=True, need_dLdinput_dparams=False, need_dLdinput_dinput=False)),
                 ("dparam", dict(need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True, need_dLdinput_dinput=False)),
                 ("dx", dict(need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=False, need_dLdinput_dinput=True))):
    f = lambda: _lotd.lod_bwd_bwd_input(m, v, dy, x, p, j, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:8s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
