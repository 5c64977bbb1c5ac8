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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--option", action="append", default=[], help="name=value for nr3d_set_option (A/B runs), repeatable")
    a = ap.parse_args()
    if a.option:
        from nr3d_lib_amd import _hip as H
        for kv in a.option:
            k, val = kv.split("=")
            H.set_option(k, int(val))
    dev = torch.device("cuda", 0)
    meta = _lotd.LoDMeta(3, RES, FEATS, TYPES, None)
    N = 1 << a.log2_points
    g = torch.Generator().manual_seed(3)
    params = torch.empty(meta.n_params).uniform_(-0.05, 0.05, generator=g).to(dev)
    x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    dL_dy = (torch.randn(N, meta.n_encoded_dims, generator=g) / 1e4).to(dev)
    v = torch.randn(N, 3, generator=g).to(dev)
    ops, spread = {}, {}

    def timed(name, fn):
        # median of >= 20 timed iterations after >= 5 warm-ups, HIP events on the op's stream around EACH iteration
        for _ in range(max(a.warmup, 3)):              # (>= 3: the allocator reaches its steady state, two live output sets)
            out = fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for e0, e1 in ev:
            e0.record()
            out = fn()
            e1.record()
        torch.cuda.synchronize()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        ops[name] = round(ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]), 4)
        spread[name] = [round(ms[0], 4), round(ms[-1], 4)]
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
    # FACTORED algorithmic bytes per point (element-granular, no cache credit): a point touches every DISTINCT table entry
    # once -- Dense 8 entries, VM 3 planes x 4 + 3 lines x 2 = 18, CP 3 lines x 2 = 6 entries of F floats per level
    # (SURVEY 8(d) quotes the un-factored count, one gather per corner and factor: 5 632 B instead of 1 936 B here).
    ent = {"Dense": 8, "VM": 18, "CP": 6}
    G = sum(ent[t] * f * 4 for t, f in zip(TYPES, FEATS))                       # all gathers            1 936 B
    Gp = sum(ent[t] * f * 4 for t, f in zip(TYPES, FEATS) if t != "Dense")      # product-type factors   1 680 B
    E4 = 4 * meta.n_encoded_dims
    # per pass: the bytes THIS build's pass moves (element granular).  dL/dx and d(dL/dx)/d(dL_dy) stream the Jacobian the
    # forward stored (3 E4 = 600 B/pt) instead of gathering again, and the forward writes it -- the model follows the
    # build, so no pass can exceed the peak (round-2 review: the gather-based model gave "fractions" of 1.5 and 1.25).
    model = {"fwd": 12 + G + E4 + 3 * E4,                      # x, gathers, y, the stored Jacobian
             "bwd_dx": E4 + 3 * E4 + 12,                       # dL_dy, Jacobian, dL_dx                            812 B
             "bwd_dparam": 12 + E4 + Gp + 2 * G,               # x, dL_dy, other factors of product levels, scatter as RMW
             "bwd_bwd_ddLdy": 12 + 3 * E4 + E4,                # dL_ddLdx, Jacobian, dL_ddLdy                      812 B
             "bwd_bwd_dparam": 24 + E4 + Gp + 2 * G,
             "bwd_bwd_dx": 24 + E4 + G + 12}
    peak = 8000.0
    per = {k: {"ms": ops[k], "algorithmic_bytes_per_point": model[k],
               "achieved": round(model[k] * N / (ops[k] * 1e-3) / 1e9, 1),
               "frac": round(model[k] * N / (ops[k] * 1e-3) / 1e9 / peak, 4)} for k in ops}
    tb = sum(model.values())
    # SURVEY 8(d)'s own figure for this config, un-factored (one gather per corner AND factor, 5 632 B of gathers per pass):
    # fwd 5 844 + bwd 17 120 + d(dL/dx)/dparam 16 864 = 39 828 B/pt.  It covers fwd + dL/dx + dL/dparam + d(dL/dx)/dparam only,
    # so it is priced on those four passes' time; the tables (8 MiB) are cache resident and the kernels gather each DISTINCT
    # entry once, so this model counts bytes nothing has to move: a fraction above 1 flags the model, not the kernels.
    survey_bpp = 39828
    t_survey = ops["fwd"] + ops["bwd_dx"] + ops["bwd_dparam"] + ops["bwd_bwd_dparam"]
    frac_survey = survey_bpp * N / (t_survey * 1e-3) / 1e9 / peak
    print(json.dumps({"workload": f"configs[3] mixed LoTD, 2^{a.log2_points} points", "n_params": meta.n_params,
                      "n_encoded_dims": meta.n_encoded_dims, "iters": a.iters, "warmup": max(a.warmup, 3),
                      "protocol": "per pass: median of the per-iteration HIP-event times (ms_min_max beside it)",
                      "ms": ops, "ms_min_max": spread, "ms_total": round(tot, 3),
                      "mpoints_per_s": round(N / tot / 1e3, 3),
                      "roofline": {"bound": "hbm", "unit": "GB/s", "peak": peak, "model": "factored (distinct table entries per point)",
                                   "algorithmic_bytes_per_point": tb, "achieved": round(tb * N / (tot * 1e-3) / 1e9, 1),
                                   "frac": round(tb * N / (tot * 1e-3) / 1e9 / peak, 4),
                                   "frac_factored": round(tb * N / (tot * 1e-3) / 1e9 / peak, 4),
                                   "frac_survey_unfactored": round(frac_survey, 4),
                                   "survey_unfactored": {"algorithmic_bytes_per_point": survey_bpp, "ms_of_its_four_passes": round(t_survey, 3),
                                                         "flag": ("model over-counts (> 1): cache-resident tables, every distinct entry gathered once"
                                                                  if frac_survey > 1 else "below 1")},
                                   "per_pass": per,
                                   "note": "the 8 MiB of tables are cache resident; HBM carries x, dL_dy, y, the Jacobian and "
                                           "the scatter records.  The forward's fraction prices 1 936 B/point of cache-resident "
                                           "gathers as if they were HBM bytes (SURVEY 8d's no-cache-credit model): a fraction above "
                                           "the ~0.79 that HBM can deliver does not mean HBM moved those bytes.  Likewise the two dL/dparam passes since round 6: "
                                           "their model charges a read-modify-write of every distinct entry per point (2 G), the kernels accumulate "
                                           "those in LDS over sorted points and write each table once"}}))


if __name__ == "__main__":
    main()
