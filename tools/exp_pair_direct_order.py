import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
dev = torch.device("cuda:0")
cfg = gen_ngp_cfg()
meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
N = 1668733
gen = torch.Generator().manual_seed(42)
params = torch.empty(meta.n_params).uniform_(-1e-4, 1e-4, generator=gen).to(dev)
x = torch.rand(N, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
g = (torch.randn(N, meta.n_encoded_dims, generator=gen) / 1e4).to(dev)
order = H.spatial_order(x.contiguous(), 8).long()
xs, gs = x[order].contiguous(), g[order].contiguous()
def run(xx, gg):
    for _ in range(3): _lotd.lod_bwd(meta, gg, xx, params, None, need_input_grad=False, need_param_grad=True)
    for k in ("lotd_bin","lotd_accum","lotd_direct"): H.prof_read(k)
    H.prof_enable("lotd_bin","lotd_accum","lotd_direct")
    for _ in range(5): _lotd.lod_bwd(meta, gg, xx, params, None, need_input_grad=False, need_param_grad=True)
    torch.cuda.synchronize(); H.prof_enable()
    return {k: round(H.prof_read(k)[0]/5*1e3,1) for k in ("lotd_bin","lotd_accum","lotd_direct")}
print(os.environ.get("NR3D_PAIR_DIRECT_MERGE","-"), "random", run(x,g), "morton", run(xs,gs))
