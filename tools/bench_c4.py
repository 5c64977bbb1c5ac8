#!/usr/bin/env python
"""tools/bench_c4.py -- BASELINE configs[3]: mixed LoTD [Dense,Dense,VM,VM,VM,CP,CP,CP], cuboid res, n_feats
[4,4,8,4,2,16,8,4], fwd + dL/dx + dL/dparam + second-order d(dL/dx)/d{dL_dy, param, x}.  Per-op HIP-event times.
    python tools/bench_c4.py [--log2-points 22] [--iters 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd.bindings import _lotd

RES = [[32, 24, 16], [64, 48, 32], [128, 96, 64], [256, 192, 128], [512, 384, 256], [1024, 768, 512],
       [2048, 1536, 1024], [4096, 3072, 2048]]
FEATS = [4, 4, 8, 4, 2, 16, 8, 4]
TYPES = ["Dense", "Dense", "VM", "VM", "VM", "CP", "CP", "CP"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-points", type=int, default=22)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    meta = _lotd.LoDMeta(3, RES, FEATS, TYPES, None)
    N = 1 << a.log2_points
    g = torch.Generator().manual_seed(3)
    params = torch.empty(meta.n_params).uniform_(-0.05, 0.05, generator=g).to(dev)
    x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    dL_dy = (torch.randn(N, meta.n_encoded_dims, generator=g) / 1e4).to(dev)
    v = torch.randn(N, 3, generator=g).to(dev)
    ops = {}

    def timed(name, fn):
        out = fn(); out = fn(); out = fn(); torch.cuda.synchronize()   # allocator reaches its steady state (two live output sets)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        ops[name] = round(e0.elapsed_time(e1) / a.iters, 4)
        return out
    y, j = timed("fwd", lambda: _lotd.lod_fwd(meta, x, params, need_input_grad=True))
    timed("bwd_dx", lambda: _lotd.lod_bwd(meta, dL_dy, x, params, j, need_input_grad=True, need_param_grad=False))
    timed("bwd_dparam", lambda: _lotd.lod_bwd(meta, dL_dy, x, params, j, need_input_grad=False, need_param_grad=True))
    timed("bwd_bwd_ddLdy", lambda: _lotd.lod_bwd_bwd_input(meta, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=True,
                                                         need_dLdinput_dparams=False, need_dLdinput_dinput=False))
    timed("bwd_bwd_dparam", lambda: _lotd.lod_bwd_bwd_input(meta, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=False,
                                                          need_dLdinput_dparams=True, need_dLdinput_dinput=False))
    timed("bwd_bwd_dx", lambda: _lotd.lod_bwd_bwd_input(meta, v, dL_dy, x, params, j, need_dLdinput_ddLdoutput=False,
                                                      need_dLdinput_dparams=False, need_dLdinput_dinput=True))
    tot = sum(ops.values())
    print(json.dumps({"workload": f"configs[3] mixed LoTD, 2^{a.log2_points} points", "n_params": meta.n_params,
                      "n_encoded_dims": meta.n_encoded_dims, "ms": ops, "ms_total": round(tot, 3),
                      "mpoints_per_s": round(N / tot / 1e3, 3)}))


if __name__ == "__main__":
    main()
