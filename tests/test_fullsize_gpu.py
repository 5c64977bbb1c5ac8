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
