"""configs[3]: dL/dparam (first and second order) with the levels regrouped at width <= 4 (wide levels as 4-feature pseudo
levels: 20-byte records) against the plain meta (2-feature pseudo levels, 12-byte records)"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.bindings import _lotd
sys.path.insert(0, os.path.join(sys.path[0], "tools"))
from bench_c4 import RES, FEATS, TYPES
dev = torch.device("cuda", 0)
meta = _lotd.LoDMeta(3, RES, FEATS, TYPES, None)
N = 1 << 22
g = torch.Generator().manual_seed(3)
params = torch.empty(meta.n_params).uniform_(-0.05, 0.05, generator=g).to(dev)
x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
dL_dy = (torch.randn(N, meta.n_encoded_dims, generator=g) / 1e4).to(dev)
gT = dL_dy.t().contiguous()
groups = []
for cap in (2, 4, 8):
    gs = []
    for width in (8, 4, 2):
        gm = _lotd._CMeta()
        H.check(H.lib().nr3d_lotd_meta_regroup(C.byref(meta._c), C.c_uint32(width), C.c_uint32(cap), C.byref(gm)))
        if gm.n_pseudo_levels:
            gs.append((gm, torch.frombuffer(bytearray(bytes(gm)), dtype=torch.uint8).to(dev)))
    groups.append((cap, gs))
H.lib().nr3d_lotd_dparam_workspace_bytes.restype = C.c_uint64
ref = None
for cap, gs in groups:
    need = max(int(H.lib().nr3d_lotd_dparam_workspace_bytes(C.byref(gm), H.u32(N), H.u32(1))) for gm, _ in gs)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    def run():
        dp = torch.zeros(meta.n_params, device=dev)
        for gm, gd in gs:
            H.check(H.lib().nr3d_lotd_bwd_dparam(C.byref(gm), H.ptr(gd), H.u32(N), C.c_int(H.F32), C.c_int(H.F32), H.ptr(gT), H.i64(1), H.i64(N),
                                                 H.ptr(x), H.ptr(params), None, None, H.u32(0), H.u32(1), H.i32(meta.n_levels), H.ptr(dp), H.ptr(ws),
                                                 C.c_uint64(need), H.stream_of(x)))
        return dp
    dp = run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): dp = run()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = dp
    err = float((dp - ref).abs().max() / ref.abs().max())
    print(f"width cap {cap}: groups {[int(gm.n_feat_per_pseudo_lvl) for gm, _ in gs]} pseudo levels {[int(gm.n_pseudo_levels) for gm, _ in gs]}  "
          f"dL/dparam {e0.elapsed_time(e1) / 3:.3f} ms  max rel diff vs cap 2: {err:.2e}")
