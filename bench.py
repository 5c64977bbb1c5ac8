#!/usr/bin/env python
"""bench.py -- headline benchmark of the nr3d hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (N = 1): BASELINE.json configs[1] -- 16-level Hash LoTD (gen_ngp_cfg defaults: T = 2^19, F = 2,
6 Dense + 10 Hash levels, 12 131 648 fp32 params), 2^20 uniformly random points, one "step" =
forward (y and dy/dx) + backward (dL/dx and dL/dparam, one call), all inputs resident in HBM.  N > 1: the same per-GPU batch on every
rank (weak scaling; points are independent so there is no data-path collective) plus ONE RCCL all-reduce of
dL/dparam per step.  Metric: whole-job Mpoints/s.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the timed region) and,
at N = 1, `cpu_baseline` (the CPU oracle -- a port, not the reference -- on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_POINTS_LOG2 = 20


def algorithmic_bytes_per_point(L, F, D=3, C=8):
    """element-granular bytes per point (no cache credit, no sector over-fetch), exactly SURVEY.md section 8(d):
    fwd = 4D + L*C*F*4 + L*F*4                                                   (x, corner gathers, y)         = 1164 B (C2)
    bwd = 4D + L*F*4 + L*C*F*4 + 2*L*C*F*4 + 4D   (x, dL_dy, gather for dL/dx, scatter as RMW, dL_dx)          = 3224 B (C2)
    ("fused, no dy_dx round trip"; this build stores the Jacobian in the forward and streams it in the backward instead
    of re-gathering, 2*L*F*D*4 = 768 B/pt of real traffic that the model does not credit)"""
    E = L * F
    return dict(fwd=4 * D + L * C * F * 4 + E * 4,
                bwd=4 * D + E * 4 + L * C * F * 4 + 2 * L * C * F * 4 + 4 * D)


def kernel_algorithmic_bytes(kernel, n_levels_served, F=2, D=3, C=8):
    """section 8(d)'s per-level figures restricted to what ONE launch of `kernel` processes, per point"""
    Ls = n_levels_served
    if kernel == "lotd_fwd":                            # x + corner gathers + y of the levels it serves
        return 4 * D + Ls * C * F * 4 + Ls * F * 4
    if kernel == "lotd_fwd_lds":
        # the level's table is staged in LDS once per workgroup: the corner values (8(d)'s 64 B per point and level) never
        # come from memory, so crediting them would put this kernel above the peak.  What it moves: x, y and the Jacobian.
        return 4 * D + Ls * F * 4 + Ls * F * D * 4
    if kernel == "lotd_contract_dx":                    # dL_dy + the stored Jacobian (8(d)'s reference-faithful variant) + dL_dx
        return Ls * F * 4 + Ls * F * D * 4 + 4 * D
    if kernel in ("lotd_bin", "lotd_accum"):            # x + dL_dy ... scatter as read-modify-write, half to each stage
        return (4 * D + Ls * F * 4 + 2 * Ls * C * F * 4) // 2
    if kernel == "lotd_direct":
        # the levels that skip the records accumulate in LDS: the scatter (8(d)'s 2 x 64 B read-modify-write per point and
        # level) never leaves the CU, so crediting it would put this kernel above the peak.  What it moves at least: x and the
        # level's two dL_dy columns per level (a level of several buckets reads them once per bucket).
        return Ls * (4 * D + F * 4)
    raise KeyError(kernel)


# timers of include/nr3d_hip.h (NR3D_PROF_*) -> kernel names as rocprofv3 prints them (the default configuration)
PROF_KERNELS = {"lotd_fwd": "k_fwd_pairlane<true, float, 2>", "lotd_fwd_lds": "k_fwd_lds<true, float, 2>",
                "lotd_contract_dx": "k_contract_dx_rowmajor<3, float>", "lotd_bin": "k_pair_bin<1024, false>",
                "lotd_accum": "k_pair_accum<4, true>", "lotd_direct": "k_pair_direct<true, false>"}
LIVE_TIMER = "lotd_fwd"      # the dominant kernel: timed inside the timed region (2 events per step); the rest in an extra pass
OP_TIMERS = {"fwd": ["lotd_fwd_lds", "lotd_fwd"], "bwd": ["lotd_contract_dx", "lotd_bin", "lotd_accum", "lotd_direct"]}


_LIB_SHA = []


def _lib_sha256():
    if not _LIB_SHA:
        import hashlib
        from nr3d_lib_amd import _hip
        _LIB_SHA.append(hashlib.sha256(open(_hip.LIB_PATH, "rb").read()).hexdigest() if os.path.exists(_hip.LIB_PATH) else None)
    return _LIB_SHA[0]


def pmc_traffic_bytes(kernels, launches=None):
    """HBM-side bytes per step of the named kernels (rocprofv3 names), from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by tools/gpu_profile.sh <tag> pmc on an MI355X; FETCH_SIZE and WRITE_SIZE are
    KiB per dispatch, `launches` = dispatches per step).  Correction per MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
    counts 64 B per 128-B request, so it is doubled -- calibrated on k_transpose, which reads exactly 128 MiB and reports
    65 552 KiB; WRITE_SIZE needs none (k_transpose writes 128 MiB and reports 131 072 KiB).  None when a kernel is absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    data = json.load(open(path))
    # the counters were collected on ONE build of the library (tools/prof_summary.py stamps its sha256): for any other build
    # -- a kernel changed after the PMC passes -- the figure would describe code that no longer runs: null then
    if data.get("__lib_sha256") != _lib_sha256():
        return None
    total = 0.0
    for want in kernels:
        hit = [ctr for name, ctr in data.items() if name.endswith("::" + want) and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr]
        if not hit:
            return None
        total += (2.0 * hit[0]["FETCH_SIZE"] + hit[0]["WRITE_SIZE"]) * 1024.0 * (launches or {}).get(want, 1)
    return int(total)


def cpu_baseline(cfg, n_sample_log2=17, min_seconds=10.0, c3=None):
    """the CPU oracle (oracle/, a C restatement with OpenMP -- kind 'port') on a bounded sample of the workload.
    `c3` = (scene tensors, n_rays, step, seconds): the march + composite leg of the metric instead (configs[2])"""
    import oracle
    cores = oracle.set_num_threads(oracle.host_cores())     # the container's CPU quota, not the host's core count
    if c3 is not None:
        return _cpu_baseline_c3(oracle, cores, *c3)
    m = oracle.lotd_create_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    rng = np.random.default_rng(42)
    n = 1 << n_sample_log2
    x = rng.random((n, 3)).astype(np.float32).clip(1e-6, 1 - 1e-6)
    p = rng.uniform(-1e-4, 1e-4, m.n_params).astype(np.float32)
    g = (rng.standard_normal((n, m.n_encoded_dims)) / 1e4).astype(np.float32)
    oracle.lotd_fwd(m, x[:1024], p, need_dydx=True)     # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        y, j = oracle.lotd_fwd(m, x, p, need_dydx=True)
        oracle.lotd_bwd_dx(m, g, j)
        oracle.lotd_bwd_dparam(m, g, x, p, accum_double=2)      # 2 = float sums over (level, point-slab) tasks: all cores busy
        reps += 1
        el = time.perf_counter() - t0
        if el >= min_seconds or reps >= 64:
            break
    return dict(value=round(reps * n / el / 1e6, 4), unit="Mpoints/s", cores=cores, kind="port",
                sample=f"{reps} x 2^{n_sample_log2} points, same 16-level meta, fwd+dydx + dL/dx + dL/dparam "
                       f"({el:.1f} s of OpenMP CPU work)")


def _cpu_baseline_c3(oracle, cores, scene, n, step, seconds):
    """configs[2] on the host: the oracle's marcher (C, OpenMP) + the reference's pack-op chain (alpha_to_vw, packed_sum x3,
    packed_div; numpy glue) forward and backward.  Returns (grid probes of one march, samples, cpu_baseline dict)."""
    o_c, d_c, near_c, far_c, roi_c, grid_c = scene
    args_np = [t.numpy() for t in (o_c, d_c, near_c, far_c, roi_c)] + [grid_c.numpy()]
    pi_r, ts_r, te_r, ridx_r, gidx_r, probes = oracle.ray_marching(*args_np, 0, step, 1e10, 0.0, 512, True, return_probes=True)
    S_r = ts_r.shape[0]
    rng = np.random.default_rng(8)
    sigma = (10.0 * rng.random(S_r)).astype(np.float32)
    rgb = rng.random((S_r, 3), dtype=np.float32)
    gm, gd, gc = (rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32),
                  rng.standard_normal((n, 3)).astype(np.float32))
    reps, t0 = 0, time.perf_counter()
    while True:
        pi_r, ts_r, te_r, ridx_r, gidx_r = oracle.ray_marching(*args_np, 0, step, 1e10, 0.0, 512, True)
        pil = pi_r.astype(np.int64)
        tt = ts_r[:, 0]
        alpha = (1 - np.exp(-sigma * (te_r - ts_r)[:, 0])).astype(np.float32)
        vw = oracle.packed_alpha_to_vw_forward(alpha, pil, 1e-4, 0.0, False)[0]
        mask = oracle.packed_sum(vw, pil)
        wn = oracle.packed_binary("div", vw, (mask + np.float32(1e-10)).astype(np.float32), pil)
        depth = oracle.packed_sum(wn * tt, pil)
        col = oracle.packed_sum(vw[:, None] * rgb, pil)
        cnt = pil[:, 1]
        rep = lambda a: np.repeat(a, cnt, axis=0)
        s_ = mask + np.float32(1e-10)
        gw = rep(gm) + rep(gd / s_) * tt - rep(gd * depth / s_) + (rep(gc) * rgb).sum(1)
        ga = oracle.packed_alpha_to_vw_backward(vw, gw.astype(np.float32), alpha, pil, 1e-4, 0.0)
        g_t, g_c = rep(gd / s_) * vw, rep(gc) * vw[:, None]
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 1000:
            break
    base = dict(value=round(reps * n / el / 1e6, 4), unit="Mrays/s", cores=cores, kind="port",
                sample=f"{reps} x {n} rays: the oracle's marcher (C, OpenMP) + the pack-op chain "
                       f"(alpha_to_vw, packed_sum x3, packed_div; numpy glue) fwd+bwd, {el:.1f} s of CPU work")
    return int(probes), int(S_r), base


def _c3_scene(side, occupancy="random"):
    """configs[2]'s scene: occ 128^3 = rand > 0.5 (seed 7; the adversarial case: runs of ~2 voxels) or the structured
    variant SURVEY 8(d) names beside it, a sphere shell with ~5 % of the voxels occupied; side^2 pinhole rays from distance 4
    looking at the origin, near / far from the ray-box test with [-1, 1]^3, step 2 sqrt(3) / 512, <= 512 samples per ray"""
    g = torch.Generator(device="cpu").manual_seed(7)
    if occupancy == "shell":
        ax = (torch.arange(128) + 0.5) / 128 * 2 - 1
        r = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).norm(dim=-1)
        grid = (r > 0.62) & (r < 0.66)                            # 5.1 % of the voxels
    else:
        grid = torch.rand(128, 128, 128, generator=g) > 0.5
    n = side * side
    u = torch.linspace(-0.4, 0.4, side)
    uu, vv = torch.meshgrid(u, u, indexing="ij")
    d = torch.stack([uu.flatten(), vv.flatten(), torch.ones(n)], 1)
    d = d / d.norm(dim=1, keepdim=True)
    o = torch.tensor([0.0, 0.0, -4.0]).repeat(n, 1)
    t1, t2 = (-1 - o) / d, (1 - o) / d
    near = torch.minimum(t1, t2).amax(1).clamp_min(0).contiguous()
    far = torch.maximum(t1, t2).amin(1).contiguous()
    far = torch.where(far > near, far, near).contiguous()
    return grid, o, d, near, far, torch.tensor([-1., -1, -1, 1, 1, 1]), 2 * 3 ** 0.5 / 512


def march_composite_rate(dev, iters=20, side=64, cpu_seconds=0.0, occupancy="random"):
    """BASELINE configs[2]: occ 128^3 march + alpha composite forward AND backward, side^2 rays
    (side = 64: the 4096 rays of configs[2] -- bound by the longest ray's dependent chain, marched with 32 lanes per ray
     looking ahead along the t recurrence; side = 512: enough rays to fill the chip, one lane per ray, the throughput
     regime of configs[4]).
    One iteration = ray march (count + emit) -> alpha = 1 - exp(-sigma * delta) -> fused composite (vw, mask, normalised
    depth, rgb) -> its backward for random upstream gradients (dL/dalpha, dL/dt, dL/drgb).
    Roofline (SURVEY 8d byte model, element-granular): march = 32 B/ray (o, d, near, far) + 8 B/ray packed_info +
    16 B/sample emitted (t0, t1, ridx, gidx) + 1 B per grid probe (probes counted by the oracle's marcher on the host);
    composite fwd = per sample alpha 4 + t 4 + rgb 12 read + vw 4 written, per ray pack_infos 16 + index 8 + 20 out;
    composite bwd = per sample alpha 4 + vw 4 + t 4 + rgb 12 read + dalpha 4 + dt 4 + drgb 12 written, per ray 16 + 8 +
    28 in.  At 4096 rays the whole op moves ~20 MB = 2.5 us at peak: it is launch / latency bound (absolute us matter);
    the fraction is information only, as SURVEY 8(d) says."""
    from nr3d_lib_amd import _hip as H
    from nr3d_lib_amd.bindings import _occ_grid, _pack_ops
    grid_c, o_c, d_c, near_c, far_c, roi_c, step = _c3_scene(side, occupancy)
    grid, o, d, near, far, roi = (t.to(dev) for t in (grid_c, o_c, d_c, near_c, far_c, roi_c))
    n = side * side
    gen = torch.Generator(device="cpu").manual_seed(8)
    state = {}

    def two_phase():
        # march (count + emit) and the post-processing every caller applies (hit rays, int64 packs, interval lengths: two
        # launches, the readback shared with the marcher's), then sigma -> alpha in one launch
        m = _occ_grid.ray_marching_finished(o, d, near, far, roi, grid, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, 512, True)
        tmid, pil = m["t_starts"], m["pack_infos"]
        S = tmid.shape[0]
        if "sigma" not in state:       # per-sample inputs of the composite: fixed scene => fixed S
            state["sigma"] = (10.0 * torch.rand(S, generator=gen)).to(dev)
            state["rgb"] = torch.rand(S, 3, generator=gen).to(dev)
            state["g"] = [torch.randn(n, generator=gen).to(dev), torch.randn(n, generator=gen).to(dev),
                          torch.randn(n, 3, generator=gen).to(dev)]
        alpha = _pack_ops.tau_to_alpha_forward(state["sigma"], m["deltas"])
        # the two launches behind graphics.pack_ops.packed_composite and its autograd backward, called directly (at 4096
        # rays the op is launch bound: no autograd graph bookkeeping inside the timed loop)
        vw, mask, depth, rgb = _pack_ops.packed_composite_forward(alpha, tmid, state["rgb"], pil, m["ridx_hit"], n, 1e-4, 0.0, True,
                                                                  packs_tile=True)     # a marcher's packs cover every sample
        ga, gt, gc = _pack_ops.packed_composite_backward(alpha, vw, tmid, state["rgb"], pil, m["ridx_hit"], 1e-4, 0.0, True,
                                                         mask, depth, state["g"][0], state["g"][1], state["g"][2], None,
                                                         packs_tile=True)
        return S, mask, ga

    def one_call():
        # round 6: the same work as ONE forward call (count -> scan -> emit -> alpha + composite, four launches, no host wait in
        # between: per-sample buffers sized by the bound n * 512) + the backward launch, THEN the single readback that sizes the views
        mc = _occ_grid.ray_marching_composite(o, d, near, far, roi, grid, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, 512,
                                              state["sigma"], state["rgb"], 1e-4, 0.0, True)
        mc.backward(state["g"][0], state["g"][1], state["g"][2])
        return mc.totals()[0], mc.view("mask"), mc.grads()[0]          # the readback; views of what the caller looks at
    S0 = two_phase()[0]                # sizes sigma / rgb (fixed scene) and is the cross-check of the one-call path
    fused = n * 512 * 72 <= _occ_grid.FUSED_MARCH_COMPOSITE_MAX_BYTES
    one = one_call if fused else two_phase
    if fused:
        a, b = one_call(), two_phase()
        assert a[0] == b[0] == S0 and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), "one-call march + composite differs from the chain"
    for _ in range(5):               # >= 5 warm-ups; the caching allocator reaches its steady state
        S = one()[0]
    names = ("march", "composite_fwd", "composite_bwd")
    # wall clock first, with the library's event hooks off (they cost ~20 us of host time per iteration,
    # tools/exp_c3_host.py); then the same loop again with the hooks on for the per-kernel times
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    for k in names:
        H.prof_read(k)
    H.prof_enable(*names)
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    H.prof_enable()
    kus = {}
    for k in names:
        tot, cnt = H.prof_read(k)
        kus[k] = tot / iters * 1e3                       # us per iteration (march: count + emit intervals)
    occ_txt = "rand > 0.5" if occupancy == "random" else f"sphere shell, {100 * float(grid_c.float().mean()):.1f} % occupied"
    out = dict(workload=f"configs[2]: occ 128^3 ({occ_txt}) march + fused alpha composite fwd+bwd, {n} rays x <= 512 samples",
               iters=iters, warmup=5,
               samples=int(S), ms_per_iter=round(ms, 4), mrays_per_s=round(n / ms / 1e3, 4),
               kernel_us_per_iter={k: round(v, 2) for k, v in kus.items()},
               route="one call (nr3d_march_composite_fwd) + backward, readback last" if fused else "two-phase chain (count -> readback -> emit -> alpha -> composite)",
               launches_per_iter=("count, scan (+ hit rays, totals), cached emit (+ per-sample epilogue), sigma -> alpha + composite over all "
                                  "rays, composite backward: 5 launches enqueued back to back, ONE device->host readback after the last") if fused else
                                 ("march 2 (count; cached emit + per-sample epilogue) + scan 1 (3 above 32768 rays: packed_info "
                                  "and the hit rays' compaction in the same scan), sigma -> alpha 1, composite 1 + 1 (+ one zero "
                                  "fill of the per-ray outputs); one device->host readback"))
    if cpu_seconds > 0:
        # the oracle's marcher counts the grid probes of the byte model, and is the CPU baseline next to the chain
        probes, S_r, base = cpu_baseline(None, c3=((o_c, d_c, near_c, far_c, roi_c, grid_c), n, step, cpu_seconds))
        b_march = 32 * n + 8 * n + 16 * S_r + probes
        b_fwd = 24 * S_r + 44 * n
        b_bwd = 44 * S_r + 52 * n
        total_b = b_march + b_fwd + b_bwd
        kern_s = sum(kus.values()) * 1e-6
        out["roofline"] = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBPS, algorithmic_bytes=int(total_b), grid_probes=int(probes),
                               achieved=round(total_b / kern_s / 1e9, 2), frac=round(total_b / kern_s / 1e9 / HBM_PEAK_GBPS, 5),
                               achieved_wall=round(total_b / (ms * 1e-3) / 1e9, 2),
                               per_kernel={k: dict(us=round(kus[k], 2), algorithmic_bytes=int(bb),
                                                   achieved=round(bb / (kus[k] * 1e-6) / 1e9, 2))
                                           for k, bb in zip(names, (b_march, b_fwd, b_bwd))},
                               note="launch / latency bound at this size (SURVEY 8d): fraction for information")
        out["cpu_baseline"] = base
    return out


def _full_loop_setup(dev, side=512, shift=0.0, precision="float"):
    """model + rays + one forward/backward of BASELINE configs[4]'s loop on `side`^2 rays (gradients accumulate)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from demo_field import DemoField, pinhole_rays
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer, nerf_ray_query_march_occ
    res = 128
    ax = (torch.arange(res) + 0.5) / res * 2 - 1
    r = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).norm(dim=-1)
    occ = ((r > 0.45) & (r < 0.8)).to(dev)                      # a shell: ~19 % of the voxels occupied
    model = DemoField(occ, 2 * 3 ** 0.5 / 512, max_steps=512, seed=1, device=dev, precision=precision)
    o, d, near, far = pinhole_rays(side, dev, shift=shift)
    n = side * side
    rays = dict(num_rays=n, rays_o=o, rays_d=d, near=near, far=far, rays_inds=torch.arange(n, device=dev))

    def fwd_bwd(count=True):
        vb, det = nerf_ray_query_march_occ(model, rays, with_rgb=True, compression=True)
        out = composite_packed_volume_buffer(vb, n)
        (out["rgb_volume"].mean() + out["depth_volume"].mean()).backward()
        # (the sample counts are the harness's, not the workload's: two reductions + two host syncs -- taken once, outside the timed loop)
        return (int(det["march.num_per_ray"].sum()), int(det["render.num_per_ray"].sum())) if count else None
    return model, n, fwd_bwd


def full_loop_rate(dev, side=512, iters=20, warmup=5, precision="float"):
    """BASELINE configs[4] on one GPU: 16-level Hash encode + occ-grid march + pack composite, forward AND backward
    through nerf_ray_query_march_occ (visibility pruning on) with a tiny random MLP head (tools/demo_field.py)."""
    model, n, fwd_bwd = _full_loop_setup(dev, side, precision=precision)

    def one(count=False):
        model.zero_grad(set_to_none=True)
        return fwd_bwd(count)
    marched, rendered = one(True)
    for _ in range(warmup - 1):
        one()
    # every iteration timed on its own: the driver allocates data-dependent buffers, and an iteration that happens to go
    # back to hipMalloc (caching-allocator miss) costs several ms -- the median is the steady-state figure, the mean is
    # reported next to it
    per_iter = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one()
        torch.cuda.synchronize()
        per_iter.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(per_iter))
    return dict(workload=f"march + prune + 16-level Hash LoTD encode + fused MLP decoders (32-wide) + composite, fwd+bwd, {n} rays"
                         + (" -- half LoTD tables / features + half decoders on the f16 MFMA (the reference's default storage)" if precision == "half" else ", fp32"),
                samples_marched=marched, samples_rendered=rendered, ms_per_iter=round(ms, 3),
                ms_per_iter_mean=round(float(np.mean(per_iter)), 3), ms_per_iter_min_max=[round(min(per_iter), 3), round(max(per_iter), 3)],
                iters=iters, warmup=warmup,
                mrays_per_s=round(n / ms / 1e3, 3), msamples_per_s=round((marched + rendered) / ms / 1e3, 3))


def full_loop_sharded_rate(dev, dist, rank, world, chunks=8, side=512, iters=3):
    """BASELINE configs[4] as stated: 2^24 rays over 8 GPUs = 2^21 rays per GPU, rendered as `chunks` x `side`^2-ray
    forward/backward passes whose gradients accumulate, then ONE all-reduce of all parameter gradients (LoTD tables +
    decoder weights, nr3d_lib_amd.distributed.allreduce_grads) per iteration.  Every rank renders its own rays (camera
    shifted by rank).  A rank that fails locally reports it through a MIN all-reduce before the gradient collective, so
    the ranks leave together instead of hanging."""
    from nr3d_lib_amd.distributed import allreduce_grads

    def all_ok(ok):
        t = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
    state = {}
    try:
        model, n, fwd_bwd = _full_loop_setup(dev, side, shift=0.05 * rank)
        state["ok"] = True
    except Exception as ex:
        state["ok"], state["err"] = False, repr(ex)[:200]
    if not all_ok(state["ok"]):
        return {"error": state.get("err", "setup failed on another rank")}
    counts = [0, 0]
    ar_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def iteration():
        ok = True
        try:
            model.zero_grad(set_to_none=True)
            counts[0] = counts[1] = 0
            for _ in range(chunks):
                m, r = fwd_bwd()
                counts[0] += m; counts[1] += r
        except Exception as ex:
            ok, state["err"] = False, repr(ex)[:200]
        if not all_ok(ok):
            return False
        ar_ev[0].record()
        allreduce_grads([p.grad for p in model.parameters()], single_rank_too=True)
        ar_ev[1].record()
        return True
    if not iteration():
        return {"error": state.get("err", "an iteration failed on another rank")}
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if not iteration():
            return {"error": state.get("err", "an iteration failed on another rank")}
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0, float(counts[0]), float(counts[1])], device=dev, dtype=torch.float64)
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    ms = float(tmax[0].item()) / iters * 1e3
    rays = world * chunks * n
    allreduce_ms = ar_ev[0].elapsed_time(ar_ev[1])            # this rank's last iteration: flatten + RCCL all-reduce + copy back
    return dict(workload=f"configs[4]: march + prune + 16-level Hash LoTD encode + fused MLP decoders + composite, fwd+bwd, "
                         f"{chunks} x {n} rays per GPU x {world} GPUs = {rays} rays per iteration, one all-reduce of all "
                         f"parameter gradients per iteration",
                rays=rays, samples_marched=int(tsum[1].item()), samples_rendered=int(tsum[2].item()), ms_per_iter=round(ms, 3),
                ms_per_chunk=round((ms - allreduce_ms) / chunks, 3), allreduce_ms=round(allreduce_ms, 3),
                mrays_per_s=round(rays / ms / 1e3, 3), iters=iters, world=world)


def c5_shard_1gpu(dev):
    """configs[4]'s per-rank shard on the ONE GPU a driver run has: 8 x 262 144 rays (= the 2^21 rays a rank of the 8-GPU job renders)
    forward + backward with accumulated gradients, then the all-reduce of all parameter gradients on a ONE-RANK RCCL group, set up
    for this figure only and destroyed before the result line is printed.  What it adds to full_loop_1gpu: the accumulation over
    chunks, the gradient flattening and the collective's launch path at their real sizes (46.3 MiB of table gradients)."""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        r = full_loop_sharded_rate(dev, dist, 0, 1)
    finally:
        if own:
            dist.destroy_process_group()
    if "workload" in r:
        r["workload"] = "configs[4]'s per-rank shard on one GPU (one-rank RCCL group): " + r["workload"].split(": ", 1)[1]
    return r


def lotd_large_batch_rate(log2n=24):
    """the headline workload at 2^24 points (the size BASELINE.json's target is quoted on), in a fresh process"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--log2-points", str(log2n), "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=900)
    line = [l[len("BENCH_FULL "):] for l in r.stdout.splitlines() if l.startswith("BENCH_FULL {")]    # the full tables, not the digest
    if not line:
        return {"error": (r.stderr or r.stdout)[-300:]}
    d = json.loads(line[-1])
    rf = d["roofline"]
    bpp = sum(v["algorithmic_bytes_per_point"] for v in rf["per_op"].values())
    med = d["ms_per_step_median"]
    return dict(workload=d["config"]["workload"], steps=d["steps"], warmup=d["warmup"], mpoints_per_s=d["value"],
                ms_per_step=d["ms_per_step"], ms_per_step_median=med, ms_per_step_min_max=d["ms_per_step_min_max"],
                kernel_ms=d["kernel_ms"], whole_step_frac=rf["whole_step_frac"],        # on ms_per_step (wall), as in the headline
                whole_step_frac_event_sum=rf.get("whole_step_frac_event_sum"),
                frac_of_median_step=round(bpp * (1 << log2n) / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                per_kernel={k: {kk: v[kk] for kk in ("avg_us", "launches_per_step", "frac")} for k, v in rf["per_kernel"].items()})


def lotd_half_rate(dev, log2n=20, iters=20):
    """configs[1] with the reference's DEFAULT storage type, (float, half, float): half params / y / dL_dy / dL_dparam, float x
    and dy_dx, fp32 arithmetic (lotd_encoding.h:1501-1504).  Served natively: no whole-table conversion."""
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    N = 1 << log2n
    gen = torch.Generator(device="cpu").manual_seed(43)
    params = torch.empty(meta.n_params).uniform_(-1e-2, 1e-2, generator=gen).to(dev).half()
    x = torch.rand(N, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
    g = (torch.randn(N, meta.n_encoded_dims, generator=gen) * 1e-2).to(dev).half()

    def one():
        y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
        return _lotd.lod_bwd(meta, g, x, params, j, need_input_grad=True, need_param_grad=True)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        dx, dp = one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    assert dp.dtype == torch.float16
    return dict(workload=f"configs[1] with half tables / outputs / gradients (float, half, float), 2^{log2n} points, fwd(+dy/dx) + dL/dx "
                         "+ dL/dparam", native=bool(_lotd._native_half(meta, params, False)), ms_per_step=round(ms, 4),
                mpoints_per_s=round(N / ms / 1e3, 3))


def lotd_second_order_rate(dev, log2n=20, iters=20):
    """the second-order passes on configs[1]'s meta: d(dL/dx)/d{dL_dy, params, x} for a random dL_ddLdx --
    what an eikonal / curvature loss adds to a step"""
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    N = 1 << log2n
    gen = torch.Generator(device="cpu").manual_seed(44)
    params = torch.empty(meta.n_params).uniform_(-1e-2, 1e-2, generator=gen).to(dev)
    x = torch.rand(N, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
    g = (torch.randn(N, meta.n_encoded_dims, generator=gen) * 1e-2).to(dev)
    v = torch.randn(N, 3, generator=gen).to(dev)
    _, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
    ms = {}
    for name, flags in (("ddLdy", (True, False, False)), ("dparam", (False, True, False)), ("dx", (False, False, True)),
                        ("one_call", (True, True, True))):     # all three in ONE call (what autograd issues; one shared copy of dL_dy)
        fn = lambda: _lotd.lod_bwd_bwd_input(meta, v, g, x, params, j, need_dLdinput_ddLdoutput=flags[0],
                                             need_dLdinput_dparams=flags[1], need_dLdinput_dinput=flags[2])
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        ms[name] = round((time.perf_counter() - t0) / iters * 1e3, 4)
    one_call = ms.pop("one_call")
    tot = sum(ms.values())
    return dict(workload=f"configs[1]'s meta, 2^{log2n} points: d(dL/dx)/d(dL_dy), d(dL/dx)/dparam, d(dL/dx)/dx", ms=ms,
                ms_total=round(tot, 4), ms_all_three_in_one_call=one_call, mpoints_per_s=round(N / tot / 1e3, 3))


def c1_dense_rate(dev):
    """BASELINE configs[0]: single Dense level 32^3 x 4 features, 65 536 points, forward only -- the HIP kernel next to a
    pure-PyTorch trilinear sampler on the host cores (grid_sample on the [32,32,32,4] table with the LoTD coordinate
    convention x * (R-2)/(R-1), align_corners=True: the formulation of the reference's CPU helper param_interpolate,
    lotd_helpers.py:274-346, restated here because the reference cannot travel to the GPU box)"""
    import torch.nn.functional as F
    from nr3d_lib_amd.bindings import _lotd
    R, Fe, n = 32, 4, 65536
    g = torch.Generator().manual_seed(0)
    params = torch.randn(R ** 3 * Fe, generator=g)
    x = torch.rand(n, 3, generator=g).clamp_(1e-6, 1 - 1e-6)
    meta = _lotd.LoDMeta(3, [R], [Fe], ["Dense"], None)
    xd, pd = x.to(dev), params.to(dev)
    y = _lotd.lod_fwd(meta, xd, pd)[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        y = _lotd.lod_fwd(meta, xd, pd)[0]
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / 50 * 1e3
    vol = params.view(1, R, R, R, Fe).permute(0, 4, 1, 2, 3).contiguous()            # [1, F, Rx, Ry, Rz]
    rel = ((x * 2 - 1) * ((R - 2.) / (R - 1.)))[:, [2, 1, 0]].view(1, 1, 1, n, 3)     # grid_sample wants (z, y, x) last
    ref = F.grid_sample(vol, rel, align_corners=True, padding_mode="zeros").view(Fe, n).t()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 2.0:
        F.grid_sample(vol, rel, align_corners=True, padding_mode="zeros"); reps += 1
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    return dict(workload="configs[0]: Dense 32^3 x 4, 65536 points, forward", gpu_ms=round(gpu_ms, 4),
                gpu_mpoints_per_s=round(n / gpu_ms / 1e3, 2), cpu_pytorch_ms=round(cpu_ms, 3),
                cpu_pytorch_mpoints_per_s=round(n / cpu_ms / 1e3, 3), cpu_threads=torch.get_num_threads(),
                max_rel_diff=float(f"{err:.2e}"))


def forest_lotd_rate(dev, log2n=20, iters=20, by_block=False):
    """SURVEY 8f rank 4 as an extra figure: LoTD over a forest of 8 blocks (dense level-1 octree, continuity on), the
    NGP config's first 8 levels (4 Dense + 4 Hash) per block, 2^20 points spread over the blocks:
    fwd(+dy/dx) + dL/dx + dL/dparam"""
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    from nr3d_lib_amd.models.spatial import ForestBlockSpace
    cfg = gen_ngp_cfg()
    L = 8
    meta = _lotd.LoDMeta(3, cfg["lod_res"][:L], cfg["lod_n_feats"][:L], cfg["lod_types"][:L], cfg["hashmap_size"])
    space = ForestBlockSpace(device=dev)
    space.populate(mode="dense", level=1)
    metas = (meta, space.meta)
    n = 1 << log2n
    g = torch.Generator().manual_seed(3)
    x = torch.rand(n, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    bi = torch.randint(0, space.n_trees, (n,), generator=g)
    if by_block:          # points grouped by block (what marching through the blocks delivers), not scattered over them
        bi = bi.sort().values
    bi = bi.to(dev)
    params = torch.empty(space.n_trees * meta.n_params).uniform_(-1e-4, 1e-4, generator=g).to(dev)
    gy = (torch.randn(n, meta.n_encoded_dims, generator=g) / 1e4).to(dev)

    def one():
        y, j = _lotd.lod_fwd(metas, x, params, bi, need_input_grad=True)
        return _lotd.lod_bwd(metas, gy, x, params, j, bi, need_input_grad=True, need_param_grad=True)
    for _ in range(5):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / iters * 1e3
    return dict(workload=f"forest LoTD, {space.n_trees} blocks x {L} levels (4 Dense + 4 Hash), 2^{log2n} points "
                         f"{'grouped by block' if by_block else 'scattered over the blocks'}, "
                         f"fwd(+dy/dx) + dL/dx + dL/dparam (binned)", ms_per_iter=round(ms, 3),
                mpoints_per_s=round(n / ms / 1e3, 2))


def mlp_decoder_rate(dev, log2n=22, iters=20, dims=(32, 64, 64, 16), with_torch=True):
    """SURVEY 8f rank 4 (second half) as an extra figure: the fused decoder 32 -> 64 -> 64 -> 16 (ReLU) on 2^22 samples
    (`dims` = (64, 64, 64, 64): the widest shape the blocks take, every tile count 2),
    forward and backward (dL/dx + all dL/dW, dL/db; forward recomputed inside), against the layer-by-layer PyTorch path.
    Roofline: the f32 MFMA (157.3 TFLOP/s dense; MI355X_MICROARCH.md), FLOPs counted on the UNPADDED layer shapes."""
    from nr3d_lib_amd.models.blocks import MLP
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    dims = list(dims)
    n = 1 << log2n
    torch.manual_seed(0)
    net = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=torch.float, device=dev)
    x = torch.randn(n, dims[0], device=dev)
    gy = torch.randn(n, dims[-1], device=dev)
    mac = sum(a * b for a, b in zip(dims[:-1], dims[1:]))

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def fwd():
        with torch.no_grad():
            return net(x)

    def fwd_bwd():
        xr = x.detach().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        net(xr).backward(gy)
    out = {}
    for name, fused in (("fused", True), ("torch", False)):
        if not fused and not with_torch:
            out[name] = None
            continue
        mlp_mod.USE_FUSED = fused
        try:
            out[name] = dict(fwd_ms=round(timed(fwd), 4), fwd_bwd_ms=round(timed(fwd_bwd), 4))
        finally:
            mlp_mod.USE_FUSED = True
    # the same decoder as MLP(dtype=half): the f16-MFMA kernels (csrc/mlp_half.hip) -- the contract of the reference's fast decoder
    # (tcnn FullyFusedMLP: half weights / activations) -- next to the reference-style autocast layer chain
    half = {}
    try:
        torch.manual_seed(0)
        net_h = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=torch.half, device=dev)
        xh, gyh = x.half(), gy.half()

        def fwd_h():
            with torch.no_grad():
                return net_h(xh)

        def fwd_bwd_h():
            xr = xh.detach().requires_grad_(True)
            net_h.zero_grad(set_to_none=True)
            net_h(xr).backward(gyh)
        for name, fused in (("fused", True), ("torch_autocast", False)):
            if not fused and not with_torch:
                continue
            mlp_mod.USE_FUSED = fused
            try:
                half[name] = dict(fwd_ms=round(timed(fwd_h), 4), fwd_bwd_ms=round(timed(fwd_bwd_h), 4))
            finally:
                mlp_mod.USE_FUSED = True
        hf = half["fused"]
        hb = max(hf["fwd_bwd_ms"] - hf["fwd_ms"], 1e-6)
        # memory model (this kernel is HBM bound, not MFMA bound): fwd reads x, writes y; bwd reads x, dL/dy, writes dL/dx (halfs)
        b_f, b_b = 2 * (dims[0] + dims[-1]), 2 * (2 * dims[0] + dims[-1])
        half["roofline"] = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBPS,
                                fwd_achieved=round(b_f * n / (hf["fwd_ms"] * 1e-3) / 1e9, 1), fwd_frac=round(b_f * n / (hf["fwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                bwd_achieved=round(b_b * n / (hb * 1e-3) / 1e9, 1), bwd_frac=round(b_b * n / (hb * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                mfma_tflops_fwd=round(2 * mac * n / (hf["fwd_ms"] * 1e-3) / 1e12, 1),
                                mfma_tflops_bwd=round(2 * 3 * mac * n / (hb * 1e-3) / 1e12, 1), mfma_peak_f16=2500.0)
    except Exception as ex:
        half["error"] = repr(ex)[:300]
    f = out["fused"]
    bwd_ms = f["fwd_bwd_ms"] - f["fwd_ms"]
    peak = 157.3
    # the forward runs on the bf16 MFMA with three-piece splits (csrc/mlp.hip dense_x3; option mlp_x3): six bf16 products per fp32
    # product, so its ceiling in fp32 FLOPs is the dense bf16 peak / 6; the backward runs on the f32 MFMA
    from nr3d_lib_amd import _hip as H
    x3 = bool(H.get_option("mlp_x3"))
    peak_fwd = 2500.0 / 6.0 if x3 else peak
    tf_fwd = 2 * mac * n / (f["fwd_ms"] * 1e-3) / 1e12
    tf_bwd = 2 * 3 * mac * n / (bwd_ms * 1e-3) / 1e12         # recomputed forward + dH chain + dW
    return dict(workload=f"fused MLP {dims} (ReLU), 2^{log2n} samples, fp32 (+ `half`: the same network as MLP(dtype=half) on the f16 MFMA)", fused=f, torch=out["torch"], half=half,
                msamples_per_s_fwd=round(n / f["fwd_ms"] / 1e3, 1), msamples_per_s_fwd_bwd=round(n / f["fwd_bwd_ms"] / 1e3, 1),
                roofline=dict(bound="mfma", unit="TFLOP/s", peak=peak, fwd_route="bf16 MFMA, 3-piece splits (6 products)" if x3 else "f32 MFMA",
                              fwd_peak=round(peak_fwd, 1), fwd_achieved=round(tf_fwd, 1), fwd_frac=round(tf_fwd / peak_fwd, 3),
                              bwd_achieved=round(tf_bwd, 1), bwd_frac=round(tf_bwd / peak, 3)))


def c4_mixed_rate():
    """BASELINE configs[3] as an extra figure (tools/bench_c4.py): mixed Dense/VM/CP LoTD, 2^22 points,
    fwd + dL/dx + dL/dparam + the three second-order passes"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_c4.py"), "--log2-points", "22", "--iters", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}


def reference_workloads(dev):
    """BASELINE.md's own workloads (the reference's unit-test timings, sections 1a / 1b) on this build, one row per published figure:
    tools/bench_reference_workloads.py"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_reference_workloads as brw
    return brw.run(dev)


COMPACT_LIMIT_BYTES = 6000     # the driver parses the LAST stdout line out of an 8 KB tail: it has to stay well under that
L2_GATHER_CEILING_GREQ_S = 255.0   # 8-byte random gathers alone: 62.3 M requests in 245 us (profiles/r03a_fwd_experiments.txt)
FWD_REQUESTS_PER_POINT_LEVEL = 4.25    # k_fwd_pairlane: TCP_TCC_READ_REQ per (point, level) (profiles/r04final_counters.txt)


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _scalars(d, keys):
    """{short: value} for the (short, path...) entries whose value exists and is not an error"""
    out = {}
    for short, *path in keys:
        v = _get(d, *path)
        if v is not None:
            out[short] = v
    return out


def compact_extra(extra):
    """one scalar group per extra figure (the full tables go to bench_extra.json and to the BENCH_FULL line)"""
    c = {}
    for name, item in extra.items():
        if isinstance(item, dict) and "error" in item:
            c[name] = {"error": str(item["error"])[:80]}
    pick = {
        "lotd_2p24_points": [("ms", "ms_per_step_median"), ("mpts", "mpoints_per_s"), ("whole_step_frac", "whole_step_frac"),
                             ("fwd_kernel_frac", "per_kernel", PROF_KERNELS["lotd_fwd"], "frac")],
        "c4_mixed_lotd": [("ms", "ms_total"), ("mpts", "mpoints_per_s"), ("frac_factored", "roofline", "frac_factored"),
                          ("frac_survey_unfactored", "roofline", "frac_survey_unfactored"),
                          ("dparam_ms", "ms", "bwd_dparam"), ("dparam_frac", "roofline", "per_pass", "bwd_dparam", "frac")],
        "full_loop_1gpu": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "full_loop_1gpu_half": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "c5_shard_1gpu": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s"), ("allreduce_ms", "allreduce_ms"), ("rays", "rays")],
        "march_composite": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s"), ("kernel_us", "kernel_us_per_iter"),
                            ("cpu_mrays", "cpu_baseline", "value"), ("cpu_cores", "cpu_baseline", "cores")],
        "march_composite_shell": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "march_composite_262144_rays": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "march_composite_262144_rays_shell": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "c1_dense_fwd": [("gpu_ms", "gpu_ms"), ("cpu_pytorch_ms", "cpu_pytorch_ms"), ("max_rel_diff", "max_rel_diff")],
        "forest_lotd": [("ms", "ms_per_iter"), ("mpts", "mpoints_per_s")],
        "forest_lotd_by_block": [("ms", "ms_per_iter"), ("mpts", "mpoints_per_s")],
        "lotd_half_params": [("ms", "ms_per_step"), ("mpts", "mpoints_per_s")],
        "lotd_second_order": [("ms", "ms_total"), ("ms_one_call", "ms_all_three_in_one_call")],
        "mlp_decoder": [("f32_fwd_ms", "fused", "fwd_ms"), ("f32_fwd_bwd_ms", "fused", "fwd_bwd_ms"),
                        ("f16_fwd_ms", "half", "fused", "fwd_ms"), ("f16_fwd_bwd_ms", "half", "fused", "fwd_bwd_ms"),
                        ("f32_mfma_frac_fwd", "roofline", "fwd_frac"), ("f16_hbm_frac_bwd", "half", "roofline", "bwd_frac")],
        "mlp_decoder_64": [("f32_fwd_bwd_ms", "fused", "fwd_bwd_ms"), ("f16_fwd_bwd_ms", "half", "fused", "fwd_bwd_ms")],
        "march_composite_262144_rays_per_gpu": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s")],
        "full_loop_2p21_rays_per_gpu": [("ms", "ms_per_iter"), ("mrays", "mrays_per_s"), ("rays", "rays")],
    }
    for name, keys in pick.items():
        if name in extra and name not in c:
            c[name] = _scalars(extra[name], keys)
    rw = extra.get("reference_workloads")
    if isinstance(rw, dict) and "rows" in rw:
        rated = [r for r in rw["rows"] if r.get("ref_over_ours") is not None]
        worst = min(rated, key=lambda r: r["ref_over_ours"]) if rated else None
        c["reference_workloads"] = dict(rows=_get(rw, "rows_with_reference_figure"), faster=_get(rw, "rows_faster_than_reference_figure"),
                                        min_ratio=worst and worst["ref_over_ours"], worst_row=worst and str(worst["name"])[:70])
    return c


def compact_line(full):
    """the ONE line the driver parses: headline + roofline + cpu_baseline + a scalar digest of the extras.
    tests/test_bench_line_cpu.py holds its size (COMPACT_LIMIT_BYTES) and that it round-trips through json"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_min_max",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config", {}))
    if isinstance(cfg.get("parallelism"), str):
        cfg["parallelism"] = cfg["parallelism"][:160]
    out["config"] = cfg
    if "kernel_ms" in full:
        out["kernel_ms"] = full["kernel_ms"]
    rf = full.get("roofline")
    if rf:
        r = {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes",
                                "algorithmic_bytes_per_point", "whole_step_frac", "whole_step_frac_event_sum", "request_rate") if k in rf}
        r["timers"] = "in-library HIP events on the launch stream, dominant kernel inside the timed region"
        if "per_kernel" in rf:        # kernel -> [avg us, frac]
            r["per_kernel_us_frac"] = {k.split("<")[0]: [v["avg_us"], v["frac"]] for k, v in rf["per_kernel"].items()}
        out["roofline"] = r
    if "cpu_baseline" in full:
        out["cpu_baseline"] = full["cpu_baseline"]
    if "extra" in full:
        out["extra"] = compact_extra(full["extra"])
        out["extra"]["full_tables"] = "bench_extra.json; BENCH_FULL line above"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT_BYTES:                     # never let a digest cost the headline: drop digests, largest first
        ex = out.get("extra", {})
        for name in sorted(ex, key=lambda k: -len(json.dumps(ex[k]))):
            del ex[name]
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= COMPACT_LIMIT_BYTES:
                break
    return line


def emit(full):
    """full tables first (one line, prefixed so that no parser takes it for the result, and bench_extra.json), the compact line LAST"""
    try:
        with open(os.path.join(ROOT, "bench_extra.json"), "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        pass
    sys.stdout.flush()
    print("BENCH_FULL " + json.dumps(full), flush=True)
    print(compact_line(full), flush=True)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run on this node and
    become that process (the driver's own `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE and never gets here)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def plumbing_check(args, rank, world):
    """--plumbing-check: the launch path without a GPU (tests/test_bench_line_cpu.py): gloo group, the barrier / MAX-over-ranks
    protocol of the timed region around an empty step, rank 0 prints a compact line with value null.  Not a measurement."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"metric": "plumbing check (no GPU work, not a measurement)", "value": None, "unit": "Mpoints/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f32", "data": "none",
              "config": {"workload": "plumbing check", "parallelism": f"dp{world}: gloo, world size {dist.get_world_size()}",
                         "per_rank_ms_per_step": [round(float(v.item()) / max(1, args.steps) * 1e3, 6) for v in every]}})
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra figures (used when profiling the timed loop)")
    ap.add_argument("--log2-points", type=int, default=N_POINTS_LOG2)
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="no in-library HIP-event timers inside the timed region (A/B of their cost; roofline from an extra pass)")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="launch path only (gloo, no GPU, no kernels): ranks start, meet, rank 0 prints a line with value null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): be the launcher
        if not args.plumbing_check and torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
        launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch {args.gpus} ranks, or run bare and let bench.py launch them)")
    if rank != 0:
        # only rank 0 reports: whatever the other ranks' libraries write to stdout (RCCL's banner is flushed at process exit, at a time
        # of its own) must not land behind rank 0's result line
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if args.plumbing_check:
        return plumbing_check(args, rank, world)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("NR3D_BENCH_FORCE_DIST") == "1":   # the latter: exercise the RCCL plumbing on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.distributed import lotd_backward_allreduce
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    N = 1 << args.log2_points
    gen = torch.Generator(device="cpu").manual_seed(42)
    params = torch.empty(meta.n_params).uniform_(-1e-4, 1e-4, generator=gen).to(dev)
    gen_r = torch.Generator(device="cpu").manual_seed(100 + rank)
    x = torch.rand(N, 3, generator=gen_r).clamp_(1e-6, 1 - 1e-6).to(dev)
    dL_dy = (torch.randn(N, meta.n_encoded_dims, generator=gen_r) / 1e4).to(dev)

    names = ("fwd", "bwd")
    ev = {k: [] for k in names}
    reduce_mode, reduce_trial = ["bucketed"], {}

    def step(record):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
        if record: e[0].record()
        y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
        if record: e[1].record()
        # ONE backward call for both gradients (what LoTDFunction.backward does, lotd.py / lotd_torch_api.cu:397-573)
        if dist is None:
            dx, dp = _lotd.lod_bwd(meta, dL_dy, x, params, j, need_input_grad=True, need_param_grad=True)
        elif reduce_mode[0] in ("bucketed", "bucketed3"):
            # the one collective of the path: dL/dparam is computed in two (levels 6..15 = 40 of 46 MiB, then 0..5) or
            # three (11..15, 6..10, 0..5) level buckets and the all-reduce of a finished bucket runs on RCCL's stream
            # while the next one is accumulated; the step ends when all reductions have completed
            # (nr3d_lib_amd/distributed.py)
            dx, dp = lotd_backward_allreduce(_lotd.lod_bwd, meta, dL_dy, x, params, j, need_input_grad=True,
                                             need_param_grad=True,
                                             first_fraction=0.8 if reduce_mode[0] == "bucketed" else (0.4, 0.8))
        else:
            dx, dp = _lotd.lod_bwd(meta, dL_dy, x, params, j, need_input_grad=True, need_param_grad=True)
            dist.all_reduce(dp)
        if record: e[2].record()
        if record:
            for k, a, b in zip(names, e[:2], e[1:]):
                ev[k].append((a, b))
        return y, dx, dp

    # N > 1: bucketed (overlapped) or single all-reduce -- the split costs ~0.1 ms of extra kernel time, the overlap hides
    # most of the 46 MiB reduction; which one wins depends on the fabric, so both are tried on a few untimed steps
    # (before the warmup) and every rank adopts the globally fastest one.  NR3D_BENCH_ALLREDUCE=bucketed|bucketed3|single
    # pins it.
    if dist is not None:
        pin = os.environ.get("NR3D_BENCH_ALLREDUCE", "")
        if pin in ("bucketed", "bucketed3", "single"):
            reduce_mode[0] = pin
        else:
            trial = {}
            for mode in ("bucketed", "bucketed3", "single"):
                reduce_mode[0] = mode
                for _ in range(3):
                    step(False)
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(6):
                    step(False)
                torch.cuda.synchronize()
                t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                trial[mode] = float(t.item()) / 6 * 1e3
            reduce_mode[0] = min(trial, key=trial.get)        # identical on every rank (decided on the all-reduced times)
            reduce_trial.update({k: round(v, 4) for k, v in trial.items()})

    from nr3d_lib_amd import _hip as H
    for _ in range(args.warmup):
        step(False)
    if dist is not None:
        dist.barrier()
    # The dominant kernel (the forward gather kernel) is bracketed by a HIP-event pair on its launch stream INSIDE the timed
    # region (nr3d_prof_enable, include/nr3d_hip.h): `roofline.achieved` comes from these intervals.  Timing every kernel
    # of the step that way costs ~2 % of the step (12 events), so the other kernels are timed on extra, untimed steps.
    live_timers = not args.no_kernel_timers
    for k in PROF_KERNELS:
        H.prof_read(k)
    H.prof_enable(*([LIVE_TIMER] if live_timers else []))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    H.prof_enable()
    live = H.prof_read(LIVE_TIMER) if live_timers else (0.0, 0)
    H.prof_enable(*PROF_KERNELS)
    n_extra = 5
    for _ in range(n_extra):
        step(False)
    torch.cuda.synchronize()
    H.prof_enable()
    kernel_us, kernel_launches = {}, {}
    for k in PROF_KERNELS:
        ms, n = H.prof_read(k)
        if n:
            kernel_us[k] = ms / n * 1e3
            kernel_launches[k] = n / n_extra
    if live[1]:
        kernel_us[LIVE_TIMER] = live[0] / live[1] * 1e3           # the live figure replaces the extra-pass one
    per_rank_ms = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)                          # every rank's own time: what the slowest-rank figure hides
        per_rank_ms = [round(float(v.item()) / args.steps * 1e3, 4) for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: the ray half of the metric, rays sharded like the points (every rank marches + composites its own 262 144
    # rays, no collective on the data path); whole-job rate = all rays / slowest rank
    multi_march = multi_loop = None
    if dist is not None and not args.no_extra and (world > 1 or os.environ.get("NR3D_BENCH_FORCE_DIST") == "1"):
        try:
            r = march_composite_rate(dev, iters=5, side=512)
            ms_local, samples_local = float(r["ms_per_iter"]), float(r["samples"])
        except Exception:          # must not break the collective below
            ms_local, samples_local = float("nan"), 0.0
        t = torch.tensor([ms_local, samples_local], device=dev, dtype=torch.float64)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        ms_all = float(tmax[0].item())
        multi_march = dict(workload=f"occ 128^3 march + alpha composite fwd+bwd, 262144 rays per GPU x {world} GPUs",
                           samples=int(tsum[1].item()), ms_per_iter=round(ms_all, 4),
                           mrays_per_s=round(world * 262144 / ms_all / 1e3, 4) if ms_all == ms_all and ms_all > 0 else None)
        torch.cuda.empty_cache()
        multi_loop = full_loop_sharded_rate(dev, dist, rank, world)

    if rank == 0:
        kms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
        L_all = meta.n_levels
        bpp = algorithmic_bytes_per_point(L_all, 2)
        # dominant kernel = largest time per step; levels it serves: the LDS-staged levels belong to k_fwd_lds
        per_step_us = {k: kernel_us[k] * kernel_launches[k] for k in kernel_us}
        dom = max(per_step_us, key=per_step_us.get)
        # levels the forward stages in LDS (lotd.hip: Dense, 2 features, table <= 96 KiB, N >= 2^18); one launch holds up
        # to two of them (<= 150 KiB together)
        lds_launches = int(round(kernel_launches.get("lotd_fwd_lds", 0)))
        n_lds = sum(1 for t, f, sz in zip(meta.level_types, meta.level_n_feats, meta.level_sizes)
                    if t == 0 and f == 2 and sz * 8 <= 96 * 1024) if lds_launches else 0
        n_dir = int(H.lib().nr3d_lotd_pair_direct_levels(ctypes.byref(meta._cmeta()), ctypes.c_uint32(min(N, 1 << 22))))
        served = {"lotd_fwd": L_all - n_lds, "lotd_fwd_lds": max(1, n_lds // max(1, lds_launches)), "lotd_contract_dx": L_all, "lotd_bin": L_all - n_dir,
                  "lotd_accum": L_all - n_dir, "lotd_direct": max(n_dir, 1)}
        per_kernel = {}
        for k, us in kernel_us.items():
            kb = kernel_algorithmic_bytes(k, served[k])
            # bytes of ONE launch: a kernel that runs once per chunk of the batch (dL/dparam passes of 2^22 points; one
            # launch per LDS-staged level is already in `served`) processes N / launches points each time
            per_launch = max(1.0, kernel_launches[k] / (lds_launches if k == "lotd_fwd_lds" and lds_launches else 1))
            ach = kb * (N / per_launch) / (us * 1e-6) / 1e9
            per_kernel[PROF_KERNELS[k]] = {"avg_us": round(us, 2), "launches_per_step": round(kernel_launches[k], 2),
                                           "algorithmic_bytes_per_point": kb, "achieved": round(ach, 1),
                                           "frac": round(ach / HBM_PEAK_GBPS, 4)}
        dom_name = PROF_KERNELS[dom]
        launches = {PROF_KERNELS[k]: int(round(v)) for k, v in kernel_launches.items()}
        at_default_size = args.log2_points == N_POINTS_LOG2
        out = {
            "metric": "Mpoints/s LoTD fwd+bwd (16-lvl hash) + Mrays/s march+composite, 1 & 8 GPU",
            "value": round(world * N * args.steps / elapsed / 1e6, 3),
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            # per-step GPU time from the event pairs around every step (fwd start -> bwd end): spread of the K timed steps
            "ms_per_step_median": round(float(np.median([a[0].elapsed_time(b[1]) for a, b in zip(ev["fwd"], ev["bwd"])])), 4),
            "ms_per_step_min_max": [round(float(f([a[0].elapsed_time(b[1]) for a, b in zip(ev["fwd"], ev["bwd"])])), 4) for f in (min, max)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 16-level Hash LoTD (gen_ngp_cfg: T=2^19, F=2, 6 Dense + 10 Hash), "
                                   f"2^{args.log2_points} points/GPU, fwd(+dy/dx) + dL/dx + dL/dparam, fp32",
                       "points_per_gpu": N, "n_params": meta.n_params,
                       "parallelism": (f"dp{world}: one process per GPU, RCCL world size {dist.get_world_size()} "
                                       f"(points sharded; all-reduce of dL/dparam, {reduce_mode[0]}"
                                       + (f"; untimed trial ms/step {reduce_trial}" if reduce_trial else "")
                                       + "; kernel_ms.bwd includes the reduction)")
                                      if dist is not None else "single GPU",
                       **({"per_rank_ms_per_step": per_rank_ms, "allreduce_mode": reduce_mode[0]} if dist is not None else {})},
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            # the dominant KERNEL (largest HIP-event time per step), SURVEY 8(d) bytes of the levels one launch serves
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": per_kernel[dom_name]["achieved"],
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": per_kernel[dom_name]["frac"],
                         "traffic": pmc_traffic_bytes([dom_name]) if at_default_size else None,
                         "avg_launch_us": per_kernel[dom_name]["avg_us"],
                         "algorithmic_bytes": per_kernel[dom_name]["algorithmic_bytes_per_point"] * N,
                         "algorithmic_bytes_per_point": per_kernel[dom_name]["algorithmic_bytes_per_point"],
                         "timers": ("in-library HIP events on the launch stream; " + PROF_KERNELS[LIVE_TIMER] + (" inside" if live_timers else " after")
                                    + " the timed region, the other kernels on 5 extra steps"),
                         "per_kernel": per_kernel,
                         "per_op": {k: {"ms": round(kms[k], 4), "algorithmic_bytes_per_point": bpp[k],
                                        "achieved": round(bpp[k] * N / (kms[k] * 1e-3) / 1e9, 1),
                                        "frac": round(bpp[k] * N / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                        "kernels": [PROF_KERNELS[t] for t in OP_TIMERS[k]],
                                        "traffic": pmc_traffic_bytes([PROF_KERNELS[t] for t in OP_TIMERS[k]], launches)
                                        if at_default_size else None}
                                    for k in names},
                         # the whole step against the roofline, on the WALL clock of the timed region (ms_per_step: what `value` is
                         # computed from); beside it the same bytes over the sum of the event-timed op times (no launch gaps)
                         "whole_step_frac": round(sum(bpp.values()) * N / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
                         "whole_step_frac_event_sum": round(sum(bpp.values()) * N / (sum(kms.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        }
        if "lotd_fwd" in kernel_us:
            # the forward gather kernel is not byte bound: its corner gathers are 8 useful bytes per 128-byte L2 line, and what
            # limits it is the rate at which the L2 channels serve lines.  Its second ceiling, next to the HBM one:
            # requests of one launch (4.25 per point and level, the counter figure) / launch time against the rate 8-byte
            # random gathers ALONE reach on this part
            req = FWD_REQUESTS_PER_POINT_LEVEL * served["lotd_fwd"] * N
            greq = req / (kernel_us["lotd_fwd"] * 1e-6) / 1e9
            out["roofline"]["request_rate"] = {"kernel": PROF_KERNELS["lotd_fwd"], "l2_read_requests": int(req),
                                               "achieved_greq_s": round(greq, 1), "ceiling_greq_s": L2_GATHER_CEILING_GREQ_S,
                                               "frac": round(greq / L2_GATHER_CEILING_GREQ_S, 4)}
        if world == 1 and not args.no_extra:
            out["extra"] = {}
            for name, fn in (("march_composite", lambda: march_composite_rate(dev, cpu_seconds=0.0 if args.no_cpu_baseline else 4.0)),
                             ("march_composite_shell", lambda: march_composite_rate(dev, cpu_seconds=0.0 if args.no_cpu_baseline else 2.0, occupancy="shell")),
                             ("march_composite_262144_rays", lambda: march_composite_rate(dev, iters=20, side=512)),
                             ("march_composite_262144_rays_shell", lambda: march_composite_rate(dev, iters=20, side=512, occupancy="shell")),
                             ("c1_dense_fwd", lambda: c1_dense_rate(dev)),
                             ("full_loop_1gpu", lambda: full_loop_rate(dev)),
                             ("full_loop_1gpu_half", lambda: full_loop_rate(dev, precision="half")),
                             ("c5_shard_1gpu", lambda: c5_shard_1gpu(dev)),
                             ("forest_lotd", lambda: forest_lotd_rate(dev)),
                             ("forest_lotd_by_block", lambda: forest_lotd_rate(dev, by_block=True)),
                             ("lotd_half_params", lambda: lotd_half_rate(dev)),
                             ("lotd_second_order", lambda: lotd_second_order_rate(dev)),
                             ("mlp_decoder", lambda: mlp_decoder_rate(dev)),
                             ("mlp_decoder_64", lambda: mlp_decoder_rate(dev, dims=(64, 64, 64, 64), with_torch=False)),
                             ("c4_mixed_lotd", c4_mixed_rate),
                             ("lotd_2p24_points", lambda: lotd_large_batch_rate(24)),
                             ("reference_workloads", lambda: reference_workloads(dev))):
                try:                     # an extra figure must never cost the headline line (or the other extras)
                    torch.cuda.empty_cache()
                    out["extra"][name] = fn()
                except Exception as ex:
                    out["extra"][name] = {"error": repr(ex)[:300]}
        if multi_march is not None:
            out.setdefault("extra", {})["march_composite_262144_rays_per_gpu"] = multi_march
            out["extra"]["full_loop_2p21_rays_per_gpu"] = multi_loop
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        result = out
    else:
        result = None
    if dist is not None:
        dist.destroy_process_group()
    # RCCL writes its banner (version, host, library path) through C stdio, which flushes at exit -- AFTER anything Python printed: on a
    # multi-rank run the last stdout line was "Librccl path : ..." instead of the result.  Tear the group down first, flush C stdio,
    # then print: the compact JSON line is the last thing this process writes.
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if result is not None:
        emit(result)


if __name__ == "__main__":
    main()
