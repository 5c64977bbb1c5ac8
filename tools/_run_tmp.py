import torch, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
dev = torch.device("cuda", 0)
cfg = gen_ngp_cfg()
meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
N = 1 << 20
g = torch.Generator().manual_seed(3)
params = torch.empty(meta.n_params).uniform_(-1e-2, 1e-2, generator=g).to(dev)
x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
dL_dy = (torch.randn(N, meta.n_encoded_dims, generator=g) / 1e2).to(dev)
v = torch.randn(N, 3, generator=g).to(dev)
y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
def timed(name, fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / iters * 1e3, 4), "ms")
for smooth in (False, True):
    meta2 = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"], smooth)
    y, j = _lotd.lod_fwd(meta2, x, params, need_input_grad=True)
    print("smoothstep" if smooth else "linear")
    timed(" bwd_bwd ddLdy", lambda: _lotd.lod_bwd_bwd_input(meta2, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=True, need_dLdinput_dparams=False, need_dLdinput_dinput=False))
    timed(" bwd_bwd dparam", lambda: _lotd.lod_bwd_bwd_input(meta2, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True, need_dLdinput_dinput=False))
    timed(" bwd_bwd dx", lambda: _lotd.lod_bwd_bwd_input(meta2, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=False, need_dLdinput_dinput=True))
