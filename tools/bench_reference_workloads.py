#!/usr/bin/env python
"""tools/bench_reference_workloads.py -- the reference's OWN benchmark workloads on this build, one row per published figure.

BASELINE.md holds the only numbers PJLab-ADG/nr3d_lib publishes for this path: developer timings pasted into its test scripts
(`torch.utils.benchmark.Timer.blocked_autorange()`, GPU not recorded, "tested on 3090" in lotd_hash_only.h:4).  This module runs
the SAME workloads -- same metas, dtypes, sizes, generators and calls -- through this build's drop-in modules and prints, per
row, `{ref_value, ref_hw, ours_us}`:

  (a) LoTD, nr3d_lib/models/grid_encodings/lotd/tests/unit_test.py:13-14,26-33,93-109,116-250: 9 levels
      [34,55,90,140,230,370,600,1000,1600], F = 2, Dense x2 + Hash x7, T = 2^20, float x, HALF params / grads;
      365 365 random points ("360k rand pts") and 3 653 653 points.  The reference's "3.6M real pts" are a private point file
      (surface points: spatially coherent); this box has no such file, so the commented-out alternative of the same script,
      `torch.rand([3653653, 3])`, is used -- uniformly random points, the HARDER case for every gather / scatter kernel.
      Also the Dense x2 + VM x7 meta of the same script (:19-25).
  (b) pack ops, nr3d_lib/graphics/pack_ops/unit_test.py:29-91,697-717,1060-1075,1098-1117,1183-1223: 4096 packs x randint(32,96);
      sort at 4096 x randint(32,64) and 4096 x randint(320,640); packed_alpha_to_vw v1 / v2 (graphics/nerf/nerf_utils.py:29-31).

Protocol: >= 20 timed calls after >= 5 warm-ups; `ours_us` = wall time per call of the whole loop with ONE
synchronisation at its end (what blocked_autorange measures: host + device, back-to-back calls), next to the per-call device
time between HIP events (median, min, max; measured in a SECOND loop, so that the event records are not part of the wall time).
A row whose 20 calls take less than 20 ms is repeated until the timed loop is that long (`iters` of the row says how many).
Hardware differs (RTX 3090 vs MI355X); the workload does not.

    python tools/bench_reference_workloads.py            (prints one JSON object; bench.py embeds it as extra.reference_workloads)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF_HW = "RTX 3090 (inferred: lotd_hash_only.h:4; the test scripts do not record the GPU)"
REF_HW_PACK = "unknown NVIDIA GPU (not recorded)"


def timed(fn, iters=20, warmup=5):
    """wall us per call over >= `iters` back-to-back calls with NOTHING else in the loop and one sync at the end (what the
    reference's Timer.blocked_autorange measures; an event pair per call costs ~8 us of host time, which a 10-us op would be
    charged for), then -- in a second loop -- the per-call device time between events.  Ops shorter than 1 ms are repeated
    until the timed loop is >= 20 ms long (blocked_autorange runs >= 200 ms), at most 2000 calls."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()

    def wall_of(n):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    wall = wall_of(iters)
    if wall * iters < 2e4:
        iters = int(min(2000, max(iters, 2e4 / max(wall, 1.0))))
        wall = wall_of(iters)
    n_ev = min(iters, 100)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    dev = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return dict(ours_us=round(wall, 2), device_us_median=round(float(np.median(dev)), 2),
                device_us_min_max=[round(min(dev), 2), round(max(dev), 2)], iters=iters, warmup=warmup)


def row(name, ref_us, src, fn, ref_hw=REF_HW, note=None, iters=20):
    out = dict(name=name, ref_value_us=ref_us, ref_hw=ref_hw, source=src)
    try:
        out.update(timed(fn, iters=iters))
        if ref_us:
            out["ref_over_ours"] = round(ref_us / out["ours_us"], 2)
    except Exception as ex:                       # a row must not cost the others
        out["error"] = repr(ex)[:300]
    if note:
        out["note"] = note
    return out


def lotd_rows(dev):
    from nr3d_lib_amd.bindings import _lotd
    res = [34, 55, 90, 140, 230, 370, 600, 1000, 1600]
    metas = {
        "dense_hash": (lambda: _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["Hash"] * 7, 2 ** 20, False),
                       "unit_test.py:26-33"),
        "dense_vm": (lambda: _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["VM"] * 7, None, False),
                     "unit_test.py:19-25"),
    }
    # reference figures in us: {meta: {op: (360k, 3.6M)}}  (hash-only kernel figures for Dense+Hash, the default c_hash_only)
    ref = {
        "dense_hash": {"fwd": (340, 1240), "fwd_dydx": (620, 2870), "bwd_dx": (370, 3690), "bwd_dparam": (1040, 36000),
                       "bwdbwd_dparam": (5590, 110100), "bwdbwd_ddLdy": (460, 4700), "bwdbwd_ddLdy_dparam": (6080, 116000),
                       "bwdbwd_all_three": (8780, 124000)},
        "dense_vm": {"fwd": (2590, 5140), "fwd_dydx": (10700, 20200), "bwd_dx": (904, 9150), "bwd_dparam": (12500, 208000),
                     "bwdbwd_dparam": (None, None), "bwdbwd_ddLdy": (None, None), "bwdbwd_ddLdy_dparam": (41700, 678000),
                     "bwdbwd_all_three": (56100, 681000)},
    }
    lines = {"fwd": "unit_test.py:123-135", "fwd_dydx": "unit_test.py:139-152", "bwd_dx": "unit_test.py:156-169",
             "bwd_dparam": "unit_test.py:173-186", "bwdbwd_dparam": "unit_test.py:190-203", "bwdbwd_ddLdy": "unit_test.py:207-220",
             "bwdbwd_ddLdy_dparam": "unit_test.py:224-237", "bwdbwd_all_three": "unit_test.py:241-254"}
    rows = []
    # generate_meta (host only): the 11-level Dense / VM / CPfast meta of unit_test.py:62-68, 20 us in the reference
    gm = lambda: _lotd.LoDMeta(3, [34, 55, 90, 140, 230, 370, 600, 1000, 1600, 2600, 4200], [2] * 11,
                               ["Dense", "Dense"] + ["VM"] * 7 + ["CPfast", "CPfast"])
    t0 = time.perf_counter()
    for _ in range(200):
        gm()
    rows.append(dict(name="generate_meta (host)", ref_value_us=20, ref_hw="host CPU (not recorded)", source="unit_test.py:62-68,116-120",
                     ours_us=round((time.perf_counter() - t0) / 200 * 1e6, 2), iters=200,
                     note="LoDMeta through ctypes: nr3d_lotd_meta_create + the Python object (no device work)"))
    for mname, (mk, msrc) in metas.items():
        meta = mk()
        gen = torch.Generator(device="cpu").manual_seed(42)
        params = (torch.randn(meta.n_params, generator=gen) / 1.0e2).to(dev).half()
        for si, (n, label) in enumerate(((365365, "365365 random points"), (3653653, "3653653 random points (reference: 3.6M real pts)"))):
            x = torch.rand(n, 3, generator=gen).to(dev)
            y, dydx = _lotd.lod_fwd(meta, x, params, None, None, None, None, True)
            grad = (torch.randn(n, meta.n_encoded_dims, generator=gen) / 1.0e4).to(dev).half()
            grad_input = torch.randn(n, 3, generator=gen).to(dev)
            ops = {
                "fwd": lambda: _lotd.lod_fwd(meta, x, params, None, None, None, None, False),
                "fwd_dydx": lambda: _lotd.lod_fwd(meta, x, params, None, None, None, None, True),
                "bwd_dx": lambda: _lotd.lod_bwd(meta, grad, x, params, dydx, None, None, None, None, True, False),
                "bwd_dparam": lambda: _lotd.lod_bwd(meta, grad, x, params, dydx, None, None, None, None, False, True),
                "bwdbwd_dparam": lambda: _lotd.lod_bwd_bwd_input(meta, grad_input, grad, x, params, dydx, None, None, None, None, False, True, False),
                "bwdbwd_ddLdy": lambda: _lotd.lod_bwd_bwd_input(meta, grad_input, grad, x, params, dydx, None, None, None, None, True, False, False),
                "bwdbwd_ddLdy_dparam": lambda: _lotd.lod_bwd_bwd_input(meta, grad_input, grad, x, params, dydx, None, None, None, None, True, True, False),
                "bwdbwd_all_three": lambda: _lotd.lod_bwd_bwd_input(meta, grad_input, grad, x, params, dydx, None, None, None, None, True, True, True),
            }
            for op, fn in ops.items():
                r = row(f"lotd {mname} {op}, {label}", ref[mname][op][si], f"{msrc}; {lines[op]}", fn)
                r["points"] = n
                if "ours_us" in r:
                    r["mpoints_per_s"] = round(n / r["ours_us"], 2)
                rows.append(r)
            del x, y, dydx, grad, grad_input
            torch.cuda.empty_cache()
    # the derived whole-step figure of BASELINE.md 1a (fwd + dy/dx, dL/dx, dL/dparam on the hash-only kernels)
    for label, refv in (("365365", 2030.0), ("3653653", 42600.0)):
        parts = [r for r in rows if r["name"].startswith("lotd dense_hash") and label + " random" in r["name"]
                 and any(k in r["name"] for k in (" fwd_dydx,", " bwd_dx,", " bwd_dparam,"))]
        if len(parts) == 3 and all("ours_us" in r for r in parts):
            tot = sum(r["ours_us"] for r in parts)
            rows.append(dict(name=f"lotd dense_hash derived step (fwd+dy/dx, dL/dx, dL/dparam), {label} points", ref_value_us=refv,
                             ref_hw=REF_HW, source="BASELINE.md 1a (sum of the three rows)", ours_us=round(tot, 2),
                             ref_over_ours=round(refv / tot, 2), mpoints_per_s=round(int(label) / tot, 2)))
    return rows


def forest_rows(dev):
    """nr3d_lib/models/grid_encodings/lotd/tests/unit_test_forest.py:16-23 (meta), :36-58 (forest: level 3, six blocks), :112-116 (inputs),
    :145-181 (timings, 3.6M points).  kaolin builds the reference's octree; here ForestBlockSpace.populate_from_corners."""
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.spatial import ForestBlockSpace
    U = "nr3d_lib/models/grid_encodings/lotd/tests/unit_test_forest.py"
    res = [34, 55, 90, 140, 230, 370, 600, 1000, 1600]
    meta = _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["VM"] * 7)
    space = ForestBlockSpace(device=dev)
    space.populate(mode="from_corners", corners=[[1, 1, 0], [1, 1, 1], [1, 1, 2], [2, 2, 2], [3, 2, 2], [4, 2, 2]], level=3)
    metas = (meta, space.meta)
    gen = torch.Generator(device="cpu").manual_seed(42)
    n = 3653653
    params = (torch.randn(meta.n_params * space.n_trees, generator=gen) / 1.0e2).to(dev).half()
    x = torch.rand(n, 3, generator=gen).to(dev)
    blidx = torch.randint(space.n_trees, (n,), generator=gen).to(dev)
    y, dydx = _lotd.lod_fwd(metas, x, params, blidx, None, None, None, True)
    grad = (torch.randn(n, meta.n_encoded_dims, generator=gen) / 1.0e4).to(dev).half()
    label = f"forest ({space.n_trees} blocks, Dense x2 + VM x7), {n} random points (reference: 3.6M real pts)"
    ops = (("fwd", 9430.0, ":145-150", lambda: _lotd.lod_fwd(metas, x, params, blidx, None, None, None, False)),
           ("fwd_dydx", 39180.0, ":152-157", lambda: _lotd.lod_fwd(metas, x, params, blidx, None, None, None, True)),
           ("bwd_dx", 9070.0, ":159-164", lambda: _lotd.lod_bwd(metas, grad, x, params, dydx, blidx, None, None, None, True, False)),
           ("bwd_dparam", 79050.0, ":166-171", lambda: _lotd.lod_bwd(metas, grad, x, params, dydx, blidx, None, None, None, False, True)),
           ("bwd_dx_dparam", 86580.0, ":173-178", lambda: _lotd.lod_bwd(metas, grad, x, params, dydx, blidx, None, None, None, True, True)))
    rows = []
    for op, refv, lines, fn in ops:
        r = row(f"lotd {op}, {label}", refv, U + lines, fn, iters=20)
        r["points"] = n
        if "ours_us" in r:
            r["mpoints_per_s"] = round(n / r["ours_us"], 2)
        rows.append(r)
    return rows


def pack_rows(dev):
    import nr3d_lib_amd.graphics.pack_ops as po
    from nr3d_lib_amd.bindings import _pack_ops as _backend
    from nr3d_lib_amd.graphics.nerf.nerf_utils import packed_alpha_to_vw_v1, packed_alpha_to_vw_v2
    U = "nr3d_lib/graphics/pack_ops/unit_test.py"
    gen = torch.Generator(device="cpu").manual_seed(7)
    rows = []
    n_per_pack = torch.randint(32, 96, [4096], generator=gen).to(dev)
    pack_infos = po.get_pack_infos_from_n(n_per_pack)
    S = int(n_per_pack.sum())
    feats = torch.randn(S, 1, generator=gen).to(dev)
    rows.append(row("packed_sum, 4096 packs x randint(32,96)", 25.15, U + ":29-39", lambda: po.packed_sum(feats, pack_infos), REF_HW_PACK))
    rows.append(row("packed_cumsum", 23.0, U + ":41-45", lambda: po.packed_cumsum(feats, pack_infos), REF_HW_PACK))
    rows.append(row("packed_cumprod", 14.0, U + ":47-51", lambda: po.packed_cumprod(feats, pack_infos), REF_HW_PACK))
    rows.append(row("packed_diff", 12.56, U + ":81-91", lambda: po.packed_diff(feats, pack_infos), REF_HW_PACK))
    zeros, ones = torch.zeros_like(n_per_pack), torch.ones_like(n_per_pack)
    rows.append(row("interleave_arange_simple", 40.0, U + ":709-713", lambda: po.interleave_arange_simple(n_per_pack), REF_HW_PACK))
    rows.append(row("interleave_arange(0, n, 1)", 60.0, U + ":714-717", lambda: po.interleave_arange(zeros, n_per_pack, ones), REF_HW_PACK))
    near = torch.rand(4096, generator=gen).to(dev)
    far = (100.0 * torch.rand(4096, generator=gen)).clamp_min_(2.0).to(dev)
    r = row("interleave_sample_step_wrt_depth_clamped (backend), 4096 rays, max 512 steps", 219.0, U + ":1183-1203",
            lambda: _backend.interleave_sample_step_wrt_depth_clamped(near, far, 512, 0.01, 0.01, 10.0), REF_HW_PACK)
    try:
        r["samples"] = int(_backend.interleave_sample_step_wrt_depth_clamped(near, far, 512, 0.01, 0.01, 10.0)[0].shape[0])
    except Exception:
        pass
    rows.append(r)
    rows.append(row("interleave_sample_step_wrt_depth_clamped (wrapper, perturb=False)", 231.0, U + ":1215-1218",
                    lambda: po.interleave_sample_step_wrt_depth_clamped(near, far, perturb=False, max_steps=512, dt_gamma=0.01, min_step_size=0.01, max_step_size=10.0), REF_HW_PACK))
    rows.append(row("interleave_sample_step_wrt_depth_clamped (wrapper, perturb=True)", 334.0, U + ":1220-1223",
                    lambda: po.interleave_sample_step_wrt_depth_clamped(near, far, perturb=True, max_steps=512, dt_gamma=0.01, min_step_size=0.01, max_step_size=10.0), REF_HW_PACK))
    for lo, hi, refv, lab in ((32, 64, 334.0, "200k"), (320, 640, 12000.0, "2Mi")):
        n_s = torch.randint(lo, hi, [4096], generator=gen).to(dev)
        pi_s = po.get_pack_infos_from_n(n_s)
        vals = torch.randn(int(n_s.sum()), generator=gen).to(dev)
        r = row(f"packed_sort, 4096 packs x randint({lo},{hi}) ({lab} elements)", refv, U + ":1098-1117", lambda: po.packed_sort(vals, pi_s), REF_HW_PACK,
                note="returns (sorted copy, indices) like the reference wrapper: clone + arange + the sort kernel + one gather")
        r["elements"] = int(vals.shape[0])
        rows.append(r)
        inplace = vals.clone()
        rows.append(row(f"packed_sort kernel alone (in place, ids), {lab} elements", None, "csrc/pack_ops/pack_ops_cuda.cu:2634-2763",
                        lambda: _backend.packed_sort_qsort(inplace, pi_s, True), REF_HW_PACK,
                        note="after the first call the data is sorted: the bitonic network's cost does not depend on the input"))
    # merge_two_packs_sorted: unit_test.py:1060-1075
    nidx_1 = torch.unique(torch.randint(4096, [4096], generator=gen)).to(dev)
    nidx_2 = torch.unique(torch.randint(4096, [8192], generator=gen)).to(dev)
    n1 = torch.randint(32, 96, [nidx_1.numel()], generator=gen).to(dev)
    n2 = torch.randint(32, 96, [nidx_2.numel()], generator=gen).to(dev)
    pi1, pi2 = po.get_pack_infos_from_n(n1), po.get_pack_infos_from_n(n2)
    t1 = po.interleave_linspace(-torch.randn(nidx_1.numel(), generator=gen).abs().to(dev), torch.randn(nidx_1.numel(), generator=gen).abs().to(dev), n1, return_idx=False)
    t2 = po.interleave_linspace(-torch.randn(nidx_2.numel(), generator=gen).abs().to(dev), torch.randn(nidx_2.numel(), generator=gen).abs().to(dev), n2, return_idx=False)
    rows.append(row("merge_two_packs_sorted, ~2.5k + ~3.5k packs, ~400k points", 1100.0, U + ":1060-1080; pack_ops.py:614-615",
                    lambda: po.merge_two_packs_sorted(t1, pi1, nidx_1, t2, pi2, nidx_2, return_val=False), REF_HW_PACK))
    na = torch.randint(32, 96, [4096], generator=gen).to(dev)
    nb = torch.randint(32, 96, [4096], generator=gen).to(dev)
    pia, pib = po.get_pack_infos_from_n(na), po.get_pack_infos_from_n(nb)
    ta = po.interleave_linspace(-torch.ones(4096, device=dev), torch.ones(4096, device=dev), na, return_idx=False)
    tb = po.interleave_linspace(-torch.rand(4096, generator=gen).to(dev), torch.rand(4096, generator=gen).to(dev), nb, return_idx=False)
    rows.append(row("merge_two_packs_sorted_aligned, 4096 + 4096 packs, ~520k points", 200.0, "pack_ops.py:642 (\"200 us @ 4k nuggets & 400k pts\")",
                    lambda: po.merge_two_packs_sorted_aligned(ta, pia, tb, pib), REF_HW_PACK))
    # packed_alpha_to_vw: nerf_utils.py:29-31 -- eval at 1M points (no grad), train at 110k points (forward + backward)
    for n_pts, mode, refs in ((1 << 20, "eval", (73.0, 178.0)), (110_000, "train", (900.0, 186.0))):
        n_pk = max(1, n_pts // 64)
        n_e = torch.randint(32, 96, [n_pk], generator=gen).to(dev)
        pi_e = po.get_pack_infos_from_n(n_e)
        Se = int(n_e.sum())
        alpha = (torch.rand(Se, generator=gen) * 0.2).to(dev)
        up = torch.randn(Se, generator=gen).to(dev)
        for ver, f, refv in (("v1 (cumprod)", packed_alpha_to_vw_v1, refs[0]), ("v2 (fused kernel)", packed_alpha_to_vw_v2, refs[1])):
            if mode == "eval":
                def fn(f=f):
                    with torch.no_grad():
                        return f(alpha, pi_e)
            else:
                def fn(f=f):
                    a = alpha.detach().requires_grad_(True)
                    f(a, pi_e).backward(up)
                    return a.grad
            r = row(f"packed_alpha_to_vw {ver}, {mode}, {Se} points in {n_pk} packs x randint(32,96)", refv, "nr3d_lib/graphics/nerf/nerf_utils.py:29-31", fn, REF_HW_PACK,
                    note="pack lengths are not stated by the reference; randint(32,96) as in its other pack tests")
            rows.append(r)
    return rows


def run(dev=None):
    dev = dev or torch.device("cuda", 0)
    out = dict(protocol="each row: >= 20 timed calls after >= 5 warm-ups; ours_us = wall time per call, back-to-back calls, one "
                        "synchronisation at the end, nothing else in the loop (what Timer.blocked_autorange measures; short ops: as many calls "
                        "as fill 20 ms); device_us_* = HIP events around each call, in a second loop",
               ours_hw="1 x MI355X", rows=[])
    for part in (lotd_rows, forest_rows, pack_rows):
        try:
            torch.cuda.empty_cache()
            out["rows"] += part(dev)
        except Exception as ex:
            out["rows"].append(dict(name=part.__name__, error=repr(ex)[:300]))
    ok = [r for r in out["rows"] if r.get("ref_over_ours")]
    out["rows_with_reference_figure"] = len(ok)
    out["rows_faster_than_reference_figure"] = sum(r["ref_over_ours"] >= 1.0 for r in ok)
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
