"""tools/exp_fwd_slab.py -- headline forward with the Dense levels 2-3 served from LDS by slab (option fwd_lds_stage = 2) or by the
two-lane kernel (1, the default): bit-identity of y / dy_dx and the kernel times (in-library event timers)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
dev = torch.device("cuda", 0)
cfg = gen_ngp_cfg()
meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
g = torch.Generator().manual_seed(1)
for log2n in (20, 22):
    N = 1 << log2n
    params = torch.empty(meta.n_params).uniform_(-1e-1, 1e-1, generator=g).to(dev)
    x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    res = {}
    for mode in (1, 2):
        H.set_option("fwd_lds_stage", mode)
        for _ in range(3):
            y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
        H.prof_read("lotd_fwd"); H.prof_read("lotd_fwd_lds")
        H.prof_enable("lotd_fwd", "lotd_fwd_lds")
        torch.cuda.synchronize(); 
        import time; t0 = time.perf_counter()
        for _ in range(20):
            y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        H.prof_enable()
        a, na = H.prof_read("lotd_fwd"); b, nb = H.prof_read("lotd_fwd_lds")
        res[mode] = (y.clone(), j.clone())
        print(f"2^{log2n} fwd_lds_stage={mode}: wall {wall:.4f} ms; k_fwd_pairlane {a / na * 1e3:.1f} us x {na // 20}; lds kernels {b / 20 * 1e3:.1f} us per call in {nb // 20} intervals")
    print("   bit-identical:", bool(torch.equal(res[1][0], res[2][0])), bool(torch.equal(res[1][1], res[2][1])))
H.set_option("fwd_lds_stage", None)
