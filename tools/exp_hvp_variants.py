#!/usr/bin/env python
"""tools/exp_hvp_variants.py -- timing decomposition of the pair-lane d(dL/dx)/dx kernel on configs[1]'s meta
(NR3D_HVP_DBG: 1 no stores, 2 no gathers, 4 no dL_dy loads, 8 no dL_ddLdx loads; results wrong by design).
    python tools/exp_hvp_variants.py [log2_points]

Needs the experiments build of the library (round 4: measurement knobs are compiled out of the production library):
    make -C nr3d_lib_amd/csrc clean && make -C nr3d_lib_amd/csrc -j8 EXTRA=-DNR3D_EXPERIMENTS"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    N = 1 << int(sys.argv[1])
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    p = torch.empty(m.n_params).uniform_(-0.1, 0.1, generator=g).to(dev)
    x = torch.rand(N, 3, generator=g).to(dev)
    dy = torch.randn(N, m.n_encoded_dims, generator=g).to(dev)
    v = torch.randn(N, 3, generator=g).to(dev)
    f = lambda: _lotd.lod_bwd_bwd_input(m, v, dy, x, p, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=False,
                                        need_dLdinput_dinput=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 10 * 1e3 * (1 << 20) / N:8.1f} us per 2^20")
else:
    lg = sys.argv[1] if len(sys.argv) > 1 else "22"
    for dbg in (0, 1, 2, 4, 8, 3, 6, 12, 7, 15, 14):
        env = dict(os.environ, NR3D_HVP_DBG=str(dbg))
        out = subprocess.run([sys.executable, __file__, lg, "child"], env=env, capture_output=True, text=True).stdout.strip()
        print(f"dbg={dbg:2d} {out}", flush=True)
