"""GPU parity at BASELINE.json's full sizes.

configs[1] (16-level NGP LoTD, 2^20 points): the whole batch against the OpenMP oracle (it finishes in seconds on the
GPU box's host cores) plus size-independent properties -- gradient-mass conservation per level and feature
(interpolation weights sum to 1, so sum over a level's table of dL/dparam[., f] == sum_i dL/dy[i, level, f]),
linearity of the scatter in dL/dy, and independence of the chunk size used by the atomic-free path.
configs[3] (mixed Dense/VM/CP, cuboid, full resolution): 2^18 points against the oracle, first and second order, and
the stated 2^22 points through properties + an oracle-checked subsample.
configs[2] is covered at full size by tests/test_occ_grid_gpu.py::test_c3_config_bit_exact (march) and
tests/test_pack_ops_gpu.py::test_c3_composite_shape (composite on 4096 packs x <= 512 samples)."""
import numpy as np
import pytest
import torch

from util import REL_TOL, assert_close

pytestmark = pytest.mark.gpu


def _c2(oracle, dev, log2n=20, seed=42):
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    m_ref = oracle.lotd_create_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    rng = np.random.default_rng(seed)
    n = 1 << log2n
    x = rng.random((n, 3), dtype=np.float32).clip(1e-6, 1 - 1e-6)
    p = rng.uniform(-1e-4, 1e-4, m.n_params).astype(np.float32)
    g = (rng.standard_normal((n, m.n_encoded_dims)) / 1e4).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    return _lotd, m, m_ref, (x, p, g), (t(x), t(p), t(g))


def test_c2_full_batch_against_oracle(oracle, dev):
    _lotd, m, m_ref, (x, p, g), (xt, pt, gt) = _c2(oracle, dev)
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    # points that sit within fp32 rounding of a cell face may legitimately land in the neighbouring cell of one level
    # (the oracle and the device round x*(R-2)+0.5 identically, so in practice there are none; allow 1e-6 of rows)
    bad = (np.abs(y.cpu().numpy() - y_ref) > REL_TOL * np.abs(y_ref).max(0)).any(1)          # per output column
    assert bad.mean() <= 1e-6, f"{bad.sum()} of {len(bad)} points differ in y"
    jj = j.reshape(x.shape[0], -1, 3).cpu().numpy()
    badj = (np.abs(jj - j_ref) > REL_TOL * np.abs(j_ref).max((0, 2), keepdims=True)).any((1, 2))
    assert badj.mean() <= 1e-6, f"{badj.sum()} points differ in dy/dx"
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    assert_close(dx, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="dL_dx")
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam", levels=m_ref)


def test_c2_gradient_mass_linearity_and_chunking(oracle, dev, monkeypatch):
    _lotd, m, m_ref, (x, p, g), (xt, pt, gt) = _c2(oracle, dev, seed=43)
    _, dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    d = m_ref.as_dict()
    dp64 = dp.double().cpu().numpy()
    g64 = g.astype(np.float64)
    scale = np.abs(g64).sum(0).max()
    for lvl, (off, size, F) in enumerate(zip(d["level_offsets"], d["level_sizes"], d["level_n_feats"])):
        table = dp64[off:off + size * F].reshape(size, F)
        cols = g64[:, 2 * lvl:2 * lvl + F].sum(0)            # F == 2 and pseudo level == level for this config
        assert np.abs(table.sum(0) - cols).max() <= 1e-5 * scale, f"level {lvl}: gradient mass not conserved"
    # linearity: scatter(2.5 * g) == 2.5 * scatter(g) (power-of-two-free factor, so not an exponent shift only)
    _, dp_s = _lotd.lod_bwd(m, gt * 2.5, xt, pt, None, need_input_grad=False, need_param_grad=True)
    assert_close(dp_s, 2.5 * dp.cpu().numpy(), name="linearity", levels=m_ref)
    # hardware-atomic scatter (the reference's algorithm) agrees with the atomic-free path at full size
    monkeypatch.setattr(_lotd, "USE_BINNED_DPARAM", False)
    _, dp_a = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    assert_close(dp_a, dp.cpu().numpy(), rel=5e-5, name="atomic vs binned", levels=m_ref)     # fp32 atomics: run-dependent rounding


C4_RES = [[32, 24, 16], [64, 48, 32], [128, 96, 64], [256, 192, 128], [512, 384, 256], [1024, 768, 512],
          [2048, 1536, 1024], [4096, 3072, 2048]]
C4_FEATS = [4, 4, 8, 4, 2, 16, 8, 4]
C4_TYPES = ["Dense", "Dense", "VM", "VM", "VM", "CP", "CP", "CP"]


def test_c4_full_resolution_against_oracle(oracle, dev):
    from nr3d_lib_amd.bindings import _lotd
    m = _lotd.LoDMeta(3, C4_RES, C4_FEATS, C4_TYPES, None)
    m_ref = oracle.lotd_create_meta(3, C4_RES, C4_FEATS, C4_TYPES, None)
    assert m.n_pseudo_levels == 25 and m.n_encoded_dims == 50
    rng = np.random.default_rng(3)
    n = 1 << 18
    x = rng.random((n, 3), dtype=np.float32).clip(1e-6, 1 - 1e-6)
    p = rng.uniform(-0.3, 0.3, m.n_params).astype(np.float32)
    g = (rng.standard_normal((n, 50)) / 1e2).astype(np.float32)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    y, j = _lotd.lod_fwd(m, t(x), t(p), need_input_grad=True)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    # a point within fp32 rounding of a cell face may land in the neighbouring cell of a 4096-wide level: such points are
    # counted (at most 1e-5 of the batch) and then LEFT OUT of every comparison below -- nothing is skipped
    bad = (np.abs(y.cpu().numpy() - y_ref) > REL_TOL * np.abs(y_ref).max(0)).any(1)
    assert bad.mean() <= 1e-5, f"{bad.sum()} of {n} points differ in y"
    good = ~bad
    x, g, v, j_ref = x[good], g[good], v[good], j_ref[good]
    xt, pt, gt, vt = t(x), t(p), t(g), t(v)
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert_close(y, y_ref[good], name="y")
    assert_close(j.reshape(x.shape[0], -1, 3), j_ref, name="dy_dx")
    assert _lotd._dparam_workspace(m, x.shape[0], dev)[1] > 0                   # the atomic-free path
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    assert_close(dx, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="dL_dx")
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam", levels=m_ref)
    ddy, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, j, need_dLdinput_ddLdoutput=True,
                                            need_dLdinput_dparams=True, need_dLdinput_dinput=True)
    assert_close(ddy, oracle.lotd_bwd_bwd_ddLdy(m_ref, v, j_ref), name="dL_ddLdy")
    assert_close(dp2, oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="2nd-order dparam", levels=m_ref)
    assert_close(dx2, oracle.lotd_bwd_bwd_dx(m_ref, v, g, x, p), name="2nd-order dx")


def test_c4_2p22_points_properties(oracle, dev, hiplib):
    """configs[3] at its stated size (2^22 points, fwd + bwd + d(dL/dx)/dparam): the oracle would take minutes here, so
    the full batch goes through size-independent properties -- a random 8192-point subsample of every per-point output
    against the oracle (points are independent), gradient-mass conservation on the Dense levels (trilinear weights sum
    to 1), linearity of both parameter scatters in dL/dy, antisymmetry of the second-order scatter in dL/d(dL/dx), and
    independence of the number of passes of the atomic-free path."""
    from nr3d_lib_amd.bindings import _lotd
    m = _lotd.LoDMeta(3, C4_RES, C4_FEATS, C4_TYPES, None)
    m_ref = oracle.lotd_create_meta(3, C4_RES, C4_FEATS, C4_TYPES, None)
    n = 1 << 22
    gen = torch.Generator(device="cpu").manual_seed(3)
    xt = torch.rand(n, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
    pt = torch.empty(m.n_params).uniform_(-0.3, 0.3, generator=gen).to(dev)
    gt = (torch.randn(n, 50, generator=gen) / 1e2).to(dev)
    vt = torch.randn(n, 3, generator=gen).to(dev)
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    ddy, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, j, need_dLdinput_ddLdoutput=True,
                                            need_dLdinput_dparams=True, need_dLdinput_dinput=True)
    # (1) per-point outputs of a subsample against the oracle
    sel = torch.randperm(n, generator=gen)[:8192].sort().values
    xs, gs, vs, ps = xt[sel.to(dev)].cpu().numpy(), gt[sel.to(dev)].cpu().numpy(), vt[sel.to(dev)].cpu().numpy(), pt.cpu().numpy()
    y_ref, j_ref = oracle.lotd_fwd(m_ref, xs, ps, need_dydx=True)
    ys = y[sel.to(dev)].cpu().numpy()
    ok = ~(np.abs(ys - y_ref) > REL_TOL * np.abs(y_ref).max(0)).any(1)     # cell-face points (<= 1e-5 of them) left out
    assert (~ok).mean() <= 1e-3
    k = torch.from_numpy(np.nonzero(ok)[0])
    assert_close(ys[ok], y_ref[ok], name="y (subsample)")
    assert_close(j.reshape(n, -1, 3)[sel.to(dev)].cpu().numpy()[ok], j_ref[ok], name="dy_dx (subsample)")
    assert_close(dx[sel.to(dev)].cpu().numpy()[ok], oracle.lotd_bwd_dx(m_ref, gs, j_ref)[ok], name="dL_dx (subsample)")
    assert_close(ddy[sel.to(dev)].cpu().numpy()[ok], oracle.lotd_bwd_bwd_ddLdy(m_ref, vs, j_ref)[ok], name="dL_ddLdy (subsample)")
    assert_close(dx2[sel.to(dev)].cpu().numpy()[ok], oracle.lotd_bwd_bwd_dx(m_ref, vs, gs, xs, ps)[ok], name="2nd-order dx (subsample)")
    # (2) gradient mass on the Dense levels: sum over the table of dL/dparam[., f] == sum_i dL/dy[i, col(f)]
    d = m_ref.as_dict()
    dp64, col = dp.double(), 0
    gsum = gt.double().sum(0).cpu().numpy()
    scale = float(gt.double().abs().sum(0).max())
    for lvl, (off, size, F, tp) in enumerate(zip(d["level_offsets"], d["level_sizes"], d["level_n_feats"], d["level_types"])):
        if tp == 0:                                                    # Dense
            table = dp64[off:off + size * F].view(size, F).sum(0).cpu().numpy()
            assert np.abs(table - gsum[col:col + F]).max() <= 1e-5 * scale, f"level {lvl}: gradient mass not conserved"
        col += F
    # (3) linearity in dL/dy (first and second order), antisymmetry in dL/d(dL/dx)
    _, dp_s = _lotd.lod_bwd(m, gt * 2.5, xt, pt, None, need_input_grad=False, need_param_grad=True)
    assert_close(dp_s, 2.5 * dp.cpu().numpy(), name="linearity", levels=m_ref)
    _, dp2_s, _ = _lotd.lod_bwd_bwd_input(m, -vt, gt * 2.5, xt, pt, None, need_dLdinput_ddLdoutput=False,
                                          need_dLdinput_dparams=True, need_dLdinput_dinput=False)
    assert_close(dp2_s, -2.5 * dp2.cpu().numpy(), name="2nd-order linearity", levels=m_ref)
    # (4) one pass (2^22-point chunk) against four passes of 2^20 points
    hiplib.nr3d_lotd_set_dparam_chunk_log2(20)
    try:
        _, dp_c = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    finally:
        hiplib.nr3d_lotd_set_dparam_chunk_log2(0)
    assert_close(dp_c, dp.cpu().numpy(), name="chunking", levels=m_ref)


def test_c2_2p24_points_properties(oracle, dev):
    """configs[1] at the size BASELINE.json's target is quoted on (2^24 points; 2 GiB of y, 6 GiB of dy/dx): the oracle would take
    minutes, so the whole batch goes through size-independent properties -- an 8192-point subsample of every per-point output
    against the oracle (points are independent), gradient-mass conservation per level and feature over ALL 2^24 points,
    linearity of the scatter in dL/dy, and invariance to how the batch is cut: the sixteen 2^20-point pieces give bit-identical
    per-point outputs and their parameter gradients sum to the one-call gradient."""
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    m_ref = oracle.lotd_create_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    n, E = 1 << 24, m.n_encoded_dims
    gen = torch.Generator(device=dev).manual_seed(24)            # generated on the device: 2.2 GiB of inputs never touch the host
    xt = torch.rand(n, 3, generator=gen, device=dev).clamp_(1e-6, 1 - 1e-6)
    pt = torch.empty(m.n_params, device=dev).uniform_(-1e-4, 1e-4, generator=gen)
    gt = torch.randn(n, E, generator=gen, device=dev) / 1e4
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    assert y.shape == (n, E) and dx.shape == (n, 3) and dp.shape == (m.n_params,)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dp).all()
    # (1) a subsample of the per-point outputs against the oracle
    sel = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:8192].sort().values.to(dev)
    xs, gs, ps = xt[sel].cpu().numpy(), gt[sel].cpu().numpy(), pt.cpu().numpy()
    y_ref, j_ref = oracle.lotd_fwd(m_ref, xs, ps, need_dydx=True)
    ys = y[sel].cpu().numpy()
    ok = ~(np.abs(ys - y_ref) > REL_TOL * np.abs(y_ref).max(0)).any(1)          # cell-face points left out (counted)
    assert (~ok).mean() <= 1e-3
    assert_close(ys[ok], y_ref[ok], name="y (subsample)")
    assert_close(j.reshape(n, -1, 3)[sel].cpu().numpy()[ok], j_ref[ok], name="dy_dx (subsample)")
    assert_close(dx[sel].cpu().numpy()[ok], oracle.lotd_bwd_dx(m_ref, gs, j_ref)[ok], name="dL_dx (subsample)")
    # (2) gradient mass, all points: sum over a level's table of dL/dparam[., f] == sum_i dL/dy[i, level, f]
    d = m_ref.as_dict()
    dp64 = dp.double()
    gsum = gt.double().sum(0).cpu().numpy()
    scale = float(gt.double().abs().sum(0).max())
    for lvl, (off, size, F) in enumerate(zip(d["level_offsets"], d["level_sizes"], d["level_n_feats"])):
        table = dp64[off:off + size * F].view(size, F).sum(0).cpu().numpy()
        assert np.abs(table - gsum[2 * lvl:2 * lvl + F]).max() <= 1e-5 * scale, f"level {lvl}: gradient mass not conserved"
    # (3) linearity of the scatter in dL/dy
    _, dp_s = _lotd.lod_bwd(m, gt * 2.5, xt, pt, None, need_input_grad=False, need_param_grad=True)
    assert_close(dp_s, 2.5 * dp.cpu().numpy(), name="linearity", levels=m_ref)
    del dp_s
    # (4) the batch in sixteen pieces: per-point outputs bit-identical, parameter gradients add up
    acc = torch.zeros(m.n_params, dtype=torch.float64, device=dev)
    for c in range(16):
        s = slice(c << 20, (c + 1) << 20)
        yc, jc = _lotd.lod_fwd(m, xt[s], pt, need_input_grad=True)
        dxc, dpc = _lotd.lod_bwd(m, gt[s], xt[s], pt, jc, need_input_grad=True, need_param_grad=True)
        assert torch.equal(yc, y[s]) and torch.equal(jc.reshape(-1), j.reshape(n, -1)[s].reshape(-1)) and torch.equal(dxc, dx[s]), \
            f"piece {c}: per-point outputs depend on the batch they are computed in"
        acc += dpc.double()
    assert_close(acc.float(), dp.cpu().numpy(), name="sum of the pieces' dL_dparam", levels=m_ref)


@pytest.fixture(scope="module")
def nccl_group_fullsize(dev):
    import os
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1, device_id=dev)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_c5_per_rank_shard_2p21_rays(dev, nccl_group_fullsize):
    """configs[4]'s per-GPU body at its stated size: 2^21 rays = 8 chunks x 262 144 rays (march -> prune -> 16-level Hash encode ->
    decoders -> composite, forward + backward, gradients accumulated) and ONE all-reduce of all parameter gradients on a one-rank
    RCCL group -- exactly what bench.py's full_loop_sharded_rate times per rank.  No oracle finishes this in seconds, so:
      * the image does not depend on the chunking: the same rays rendered in 32 chunks of 65 536 give bit-identical per-ray
        outputs (rays are independent);
      * the accumulated gradient is the sum of the chunks' own gradients;
      * the marcher's packs are shard-local: chunk c's pack_infos / t are bit-equal to rows [c n, (c+1) n) of ONE march over all
        2^21 rays, shifted by the samples in front of them;
      * the all-reduce (SUM over one rank) leaves the gradients unchanged."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from demo_field import DemoField, pinhole_rays
    from nr3d_lib_amd.distributed import allreduce_grads
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer, nerf_ray_query_march_occ
    res, side, chunks = 128, 512, 8
    ax = (torch.arange(res) + 0.5) / res * 2 - 1
    r = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).norm(dim=-1)
    occ = ((r > 0.45) & (r < 0.8)).to(dev)
    model = DemoField(occ, 2 * 3 ** 0.5 / 512, max_steps=512, seed=1, device=dev)
    n = side * side
    parts = [pinhole_rays(side, dev, shift=0.05 * c) for c in range(chunks)]          # every chunk its own camera
    o, d, near, far = (torch.cat([p[k] for p in parts]).contiguous() for k in range(4))
    N = chunks * n
    assert N == 1 << 21

    def render(lo, hi, backward):
        rays = dict(num_rays=hi - lo, rays_o=o[lo:hi], rays_d=d[lo:hi], near=near[lo:hi], far=far[lo:hi],
                    rays_inds=torch.arange(hi - lo, device=dev))
        vb, det = nerf_ray_query_march_occ(model, rays, with_rgb=True, compression=True)
        out = composite_packed_volume_buffer(vb, hi - lo, device=dev)       # (an empty buffer carries no tensors to take the device from)
        if backward:
            # sums, not means: a chunk's loss must not depend on how many rays the chunk has
            (out["rgb_volume"].sum() * (1.0 / N) + out["depth_volume"].sum() * (1.0 / N)).backward()
        cnt = lambda k: int(det[k].sum()) if k in det else 0          # a piece whose rays all miss the shell: empty buffer, no details
        return {k: v.detach() for k, v in out.items()}, cnt("march.num_per_ray"), cnt("render.num_per_ray")

    # --- the shard as bench.py runs it: 8 chunks, gradients accumulate, one all-reduce ---
    model.zero_grad(set_to_none=True)
    img8, marched, rendered = [], 0, 0
    for c in range(chunks):
        out, m_, r_ = render(c * n, (c + 1) * n, True)
        img8.append(out); marched += m_; rendered += r_
    assert marched > 40_000_000 and rendered > 8_000_000, (marched, rendered)
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(torch.isfinite(g).all() for g in grads.values()) and float(grads["grid"].abs().max()) > 0
    allreduce_grads([p.grad for p in model.parameters()], single_rank_too=True)        # really issued: RCCL, one rank
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, grads[k]), f"one-rank all-reduce changed grad {k}"
    # --- gradient = sum of the chunks' own gradients ---
    acc = {k: torch.zeros_like(g, dtype=torch.float64) for k, g in grads.items()}
    for c in range(chunks):
        model.zero_grad(set_to_none=True)
        render(c * n, (c + 1) * n, True)
        for k, p in model.named_parameters():
            acc[k] += p.grad.double()
    for k in grads:
        assert_close(grads[k].reshape(-1), acc[k].reshape(-1).cpu().numpy(), rel=1e-5 if k == "grid" else 2e-4,
                     name=f"accumulated grad {k} vs sum of per-chunk grads",
                     levels=model.encoding.meta if k == "grid" and hasattr(model.encoding, "meta") else None)
    # --- the image does not depend on the chunking ---
    with torch.no_grad():
        q = n // 4
        for c in range(chunks):
            for s in range(4):
                out, _, _ = render(c * n + s * q, c * n + (s + 1) * q, False)
                for k in ("mask_volume", "depth_volume", "rgb_volume"):
                    assert torch.equal(out[k], img8[c][k][s * q:(s + 1) * q]), f"chunk {c}.{s}: {k} depends on the chunking"
    # --- packs are shard-local ---
    with torch.no_grad():
        whole = model.accel.ray_march(o, d, near, far)
        assert int(whole.pack_infos[:, 1].sum()) == marched
        cnt_all = torch.zeros(N, dtype=torch.int64, device=dev)
        cnt_all[whole.ridx_hit] = whole.pack_infos[:, 1]
        first = torch.cumsum(cnt_all, 0) - cnt_all
        for c in range(chunks):
            part = model.accel.ray_march(o[c * n:(c + 1) * n], d[c * n:(c + 1) * n], near[c * n:(c + 1) * n], far[c * n:(c + 1) * n])
            base = int(first[c * n])
            S = int(part.pack_infos[:, 1].sum())
            in_chunk = (whole.ridx_hit >= c * n) & (whole.ridx_hit < (c + 1) * n)
            assert torch.equal(whole.ridx_hit[in_chunk] - c * n, part.ridx_hit), f"chunk {c}: hit rays differ"
            pi_w = whole.pack_infos[in_chunk]
            assert torch.equal(pi_w[:, 1], part.pack_infos[:, 1]) and torch.equal(pi_w[:, 0] - base, part.pack_infos[:, 0]), \
                f"chunk {c}: pack_infos are not the unsharded ones shifted by the samples in front"
            assert torch.equal(whole.depth_samples[base:base + S], part.depth_samples), f"chunk {c}: t differs"
            assert torch.equal(whole.deltas[base:base + S], part.deltas)
