"""GPU parity: LoTD HIP kernels (through the C ABI / bindings._lotd) vs the CPU oracle.

Integer outputs (grid indices) bit-exact; fp32 values within REL_TOL = 1e-5 of max|ref|
(BASELINE.json north_star).  Param gradients come from fp32 atomics whose order is not deterministic;
their reference is the oracle accumulated in float64 (order-free)."""
import numpy as np
import pytest
import torch

from util import LOTD_CASES, assert_close, assert_equal, lotd_inputs

pytestmark = pytest.mark.gpu

N = 3000


def _setup(oracle, dev, case, n=N, seed=0, n_batch=1):
    from nr3d_lib_amd.bindings import _lotd
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    x, params, dL_dy, v = lotd_inputs(m_ref.as_dict(), n, seed, n_batch=n_batch)
    t = lambda a: torch.from_numpy(a).to(dev)
    return _lotd, m_ref, m, (x, params, dL_dy, v), (t(x), t(params), t(dL_dy), t(v))


@pytest.mark.parametrize("case", list(LOTD_CASES))
def test_fwd_and_jacobian(oracle, dev, case):
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert tuple(y.shape) == y_ref.shape
    assert_close(y, y_ref, name="y")
    assert_close(j.reshape(y_ref.shape[0], -1, m.n_dims_to_encode), j_ref, name="dy_dx")
    y2, j2 = _lotd.lod_fwd(m, xt, pt, need_input_grad=False)
    assert j2 is None
    assert_close(y2, y_ref, name="y(no grad)")
    # row-major Jacobian layout of the reference's generic path
    m.c_permute_dydx = False
    y3, j3 = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert tuple(j3.shape) == (x.shape[0], m.n_encoded_dims * m.n_dims_to_encode) and j3.is_contiguous()
    assert_close(j3.view(j_ref.shape), j_ref, name="dy_dx row-major")


@pytest.fixture(params=["direct", "records"])
def bin_mode(request, hip_option):
    """dL/dparam of the pair path's levels with <= 4 buckets: accumulated straight from (x, dL_dy) in LDS (k_pair_direct,
    default) or through records like the large levels (option pair_direct = 0)"""
    hip_option("pair_direct", "1" if request.param == "direct" else "0")
    return request.param


@pytest.mark.parametrize("case", ["mixed", "mixed_cuboid", "mixed_smooth", "cp_only_4d", "cp_4d", "nplane_4d"])
def test_fwd_split_launch_same_bits(oracle, dev, case, hip_option):
    """the Dense / Hash levels of a meta that also has product-type levels run through the lean Dense / Hash kernel in a
    launch of their own (twice the occupancy): the same code for those levels, so not a bit may change"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=5003, seed=3)
    outs = {}
    for mode in ("1", "0"):
        hip_option("fwd_split", mode)
        y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
        y2, _ = _lotd.lod_fwd(m, xt, pt, need_input_grad=False)
        ym, jm = _lotd.lod_fwd(m, xt, pt, max_level=m.n_levels // 2, need_input_grad=True)
        outs[mode] = (y, j, y2, ym, jm)
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)
    y_ref, _ = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    assert_close(outs["1"][0], y_ref, name="y split launch")


@pytest.mark.parametrize("case", ["mixed", "mixed_cuboid", "nplane"])
def test_regrouped_forward_equals_the_single_call(oracle, dev, case, monkeypatch):
    """the forward of a mixed-width meta with product-type levels runs as one call per feature width (8 / 4 / 2: wider lanes
    for the wide levels, bindings._lotd.REGROUP; the calls write disjoint columns through the meta's map_col) -- against the
    single call in pseudo levels of the global gcd (the reference's decomposition) and against the oracle: values and
    Jacobian, with max_level, without the Jacobian, and with a row-major Jacobian"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=6007, seed=21)
    assert m._groups is not None and len(m._groups) >= 2
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(_lotd, "REGROUP", on)
        res = []
        for kw in ({}, dict(max_level=m.n_levels // 2)):
            y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True, **kw)
            y0, _ = _lotd.lod_fwd(m, xt, pt, need_input_grad=False, **kw)
            assert torch.equal(y, y0)
            res += [y, j]
        outs[on] = res
    for a, b, nm in zip(outs[True], outs[False], ["y", "dy_dx", "y (max_level)", "dy_dx (max_level)"]):
        assert a.shape == b.shape
        assert_close(a, b.cpu().numpy(), rel=2e-6, name=f"regrouped vs single call: {nm}")
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    assert_close(outs[True][0], y_ref, name="y regrouped")
    assert_close(outs[True][1], j_ref, name="dy_dx regrouped")
    # the gradients that read this Jacobian
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, outs[True][1], need_input_grad=True, need_param_grad=True)
    assert_close(dx, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="dL_dx from the regrouped Jacobian")
    monkeypatch.setattr(m, "c_permute_dydx", False)            # row-major [N, E * D] Jacobian
    monkeypatch.setattr(_lotd, "REGROUP", True)
    y3, j3 = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert_close(j3.view(j_ref.shape), j_ref, name="dy_dx row-major regrouped")


@pytest.mark.parametrize("case", list(LOTD_CASES))
def test_bwd(oracle, dev, case, bin_mode):
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, seed=1)
    _, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    _, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    assert_close(dx, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="dL_dx")
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam", levels=m_ref)
    # strided dL_dy (e.g. a transposed view) must be honoured
    dx2, dp2 = _lotd.lod_bwd(m, gt.t().contiguous().t(), xt, pt, j.contiguous(), need_input_grad=True,
                             need_param_grad=True)
    assert_close(dx2, dx.cpu().numpy(), name="dL_dx strided")
    assert_close(dp2, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam strided", levels=m_ref)
    none_dx, none_dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=False, need_param_grad=False)
    assert none_dx is None and none_dp is None


@pytest.mark.parametrize("case", list(LOTD_CASES))
def test_bwd_bwd_input(oracle, dev, case, bin_mode):
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, seed=2)
    _, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    _, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    ddy, dp, dx = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, j, need_dLdinput_ddLdoutput=True,
                                          need_dLdinput_dparams=True, need_dLdinput_dinput=True)
    assert_close(ddy, oracle.lotd_bwd_bwd_ddLdy(m_ref, v, j_ref), name="dL_ddLdy")
    assert_close(dp, oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="d(dLdx)/dparam", levels=m_ref)
    assert_close(dx, oracle.lotd_bwd_bwd_dx(m_ref, v, g, x, p), name="d(dLdx)/dx")


@pytest.mark.parametrize("case", ["ngp_small", "ngp_smooth", "ngp_pair", "pair_f4"])
@pytest.mark.parametrize("coherent", [False, True])
def test_dense_quad_records_vs_pair_records(oracle, dev, case, coherent, bin_mode, hip_option):
    """dL/dparam of the pair path: the Dense levels' records in quad form (four entries of two neighbouring rows per 16-byte
    record, both weights as 24-bit fractions; default) against pair records only (option pair_quad = 0) and the fp64-accumulated
    oracle; random points and runs of points inside one cell (merged lanes emit singles, never quads); half gradients too"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=9001, seed=51)
    if coherent:
        base = x[::40].repeat(40, axis=0)[: x.shape[0]]
        x = np.clip(base + (np.linspace(0, 2e-3, x.shape[0], dtype=np.float32)[:, None] % 1e-4), 1e-6, 1 - 1e-6).astype(np.float32)
        xt = torch.from_numpy(x).to(dev)
    outs = {}
    for mode in ("1", "0"):
        hip_option("pair_quad", mode)
        outs[mode] = (_lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)[1],
                      _lotd.lod_bwd(m, gt, xt, pt, None, max_level=m.n_levels // 2, need_input_grad=False, need_param_grad=True)[1],
                      _lotd.lod_bwd(m, gt.half(), xt, pt.half(), None, need_input_grad=False, need_param_grad=True)[1])
    ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    assert_close(outs["1"][0], ref, name="dL/dparam, quad records", levels=m_ref)
    assert_close(outs["0"][0], ref, name="dL/dparam, pair records", levels=m_ref)
    assert_close(outs["1"][0], outs["0"][0].cpu().numpy(), rel=1e-6, name="quad vs pair records", levels=m_ref)
    assert_close(outs["1"][1], outs["0"][1].cpu().numpy(), rel=1e-6, name="quad vs pair records, max_level", levels=m_ref)
    assert_close(outs["1"][2].float(), outs["0"][2].float().cpu().numpy(), rel=1e-3, name="quad vs pair records, half", levels=m_ref)


@pytest.mark.parametrize("case", ["mixed", "mixed_cuboid", "mixed_smooth"])
@pytest.mark.parametrize("coherent", [False, True])
def test_vm_stage_a_three_threads_per_point(oracle, dev, case, coherent, hip_option):
    """dL/dparam and d(dL/dx)/dparam of metas with VM levels: stage A with three threads per point (six records each; default)
    against one thread per point (option vm_split = 0) -- the same records, stage B may add them in another order (fp64
    accumulators) -- and the oracle; also for runs of points inside one cell (the coherent-lane merge, per component)"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=7013, seed=41)
    if coherent:
        base = x[::50].repeat(50, axis=0)[: x.shape[0]]
        x = np.clip(base + (np.linspace(0, 3e-3, x.shape[0], dtype=np.float32)[:, None] % 2e-4), 1e-6, 1 - 1e-6).astype(np.float32)
        xt = torch.from_numpy(x).to(dev)
    outs = {}
    # "lines": three threads per point, the line tables' gradients accumulated in LDS, 12 plane records (round 4, default);
    # "1": three threads, 18 records; "0": one thread, 18 records
    # "direct": VM levels of <= 4 LDS-sized slices (every VM level of these small metas) accumulate in LDS without records
    for mode, (split, lines, direct) in (("direct", (1, 1, 1)), ("lines", (1, 1, 0)), ("1", (1, 0, 0)), ("0", (0, 0, 0))):
        hip_option("vm_split", split)
        hip_option("vm_lines_direct", lines)
        hip_option("vm_direct", direct)
        dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)[1]
        dp2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                      need_dLdinput_dinput=False)[1]
        dpm = _lotd.lod_bwd(m, gt, xt, pt, None, max_level=m.n_levels // 2, need_input_grad=False, need_param_grad=True)[1]
        outs[mode] = (dp, dp2, dpm)
    assert_close(outs["1"][0], outs["0"][0].cpu().numpy(), rel=1e-6, name="dL/dparam, 3 threads vs 1", levels=m_ref)
    assert_close(outs["1"][1], outs["0"][1].cpu().numpy(), rel=1e-6, name="d(dL/dx)/dparam, 3 threads vs 1", levels=m_ref)
    for k, nm in enumerate(("dL/dparam", "d(dL/dx)/dparam", "dL/dparam, max_level")):
        assert_close(outs["lines"][k], outs["1"][k].cpu().numpy(), rel=1e-6, name=f"{nm}, line tables in LDS vs records", levels=m_ref)
    assert_close(outs["1"][0], oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL/dparam", levels=m_ref)
    assert_close(outs["1"][1], oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="d(dL/dx)/dparam", levels=m_ref)
    for k, nm in enumerate(("dL/dparam", "d(dL/dx)/dparam", "dL/dparam, max_level")):
        assert_close(outs["direct"][k], outs["1"][k].cpu().numpy(), rel=1e-6, name=f"{nm}, VM levels without records vs records", levels=m_ref)
    assert_close(outs["direct"][0], oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL/dparam (VM direct)", levels=m_ref)
    assert_close(outs["direct"][1], oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="d(dL/dx)/dparam (VM direct)", levels=m_ref)
    assert_close(outs["lines"][0], oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL/dparam (lines in LDS)", levels=m_ref)
    assert_close(outs["lines"][1], oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="d(dL/dx)/dparam (lines in LDS)", levels=m_ref)


@pytest.mark.parametrize("case", ["ngp_small", "ngp_smooth", "ngp_pair", "pair_f4"])
@pytest.mark.parametrize("coherent", [False, True])
def test_second_order_dparam_pair_records_vs_corner_records(oracle, dev, case, coherent, bin_mode, hip_option):
    """d(dL/dx)/dparam of pair-path metas: pair records carrying the second-order weights (A_f = g_f C_m, wp' = wp + E_m / C_m;
    default) against the 12-byte corner records (option pair_second = 0) and the fp64-accumulated oracle -- random points and
    ray-like runs of points inside one cell (the coherent-lane merge sums the lower / upper halves of many records); a
    direction v along one axis makes C_m or E_m vanish for whole records (the guarded division)"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=6007, seed=31)
    if coherent:
        base = x[::40].repeat(40, axis=0)[: x.shape[0]]
        x = np.clip(base + np.linspace(0, 2e-4, x.shape[0], dtype=np.float32)[:, None] % 1e-3, 1e-6, 1 - 1e-6).astype(np.float32)
        xt = torch.from_numpy(x).to(dev)
    for vv in (v, v * np.array([1.0, 0.0, 0.0], np.float32), v * np.array([0.0, 0.0, 1.0], np.float32)):
        vvt = torch.from_numpy(np.ascontiguousarray(vv)).to(dev)
        outs = {}
        for mode in ("1", "0"):
            hip_option("pair_second", mode)
            outs[mode] = _lotd.lod_bwd_bwd_input(m, vvt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                                 need_dLdinput_dinput=False)[1]
        ref = oracle.lotd_bwd_bwd_dparam(m_ref, vv, g, x, p, accum_double=True)
        assert torch.isfinite(outs["1"]).all()
        assert_close(outs["1"], ref, name="d(dL/dx)/dparam, pair records", levels=m_ref)
        assert_close(outs["0"], ref, name="d(dL/dx)/dparam, corner records", levels=m_ref)


@pytest.mark.parametrize("case", ["ngp_small", "ngp_smooth", "ngp_pair", "pair_f4", "dense_f8", "hash_npow2", "hash_4d", "dense_2d"])
def test_hvp_level_parallel_vs_lane_serial(oracle, dev, case, hip_option):
    """d(dL/dx)/dx of Dense / Hash metas: one lane per (point, pseudo level) + a sum in level order (default, needs a
    scratch buffer) against one lane per point walking the levels (option hvp_levels = 0): the same order of the sum and
    -- except for the pair-lane kernel of 3-D 2-feature metas, which builds the Hessian from the lerp tree's differences --
    the same per-level arithmetic: identical for 2-feature pseudo levels, to the rounding of one extra association
    otherwise; with max_level too"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=9001, seed=17)
    outs = {}
    for mode in ("1", "0"):
        hip_option("hvp_levels", mode)
        a = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=False,
                                    need_dLdinput_dinput=True)[2]
        b = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, max_level=m.n_levels // 2, need_dLdinput_ddLdoutput=False,
                                    need_dLdinput_dparams=False, need_dLdinput_dinput=True)[2]
        outs[mode] = (a, b)
    pair_lane = m.n_dims_to_encode == 3 and m.n_feat_per_pseudo_lvl == 2      # lerp-tree Hessian: other association
    if m.n_feat_per_pseudo_lvl == 2 and not pair_lane:
        assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    else:
        rel = 2e-6 if pair_lane else 1e-6
        assert_close(outs["1"][0], outs["0"][0].cpu().numpy(), rel=rel, name="hvp level-parallel vs lane-serial")
        assert_close(outs["1"][1], outs["0"][1].cpu().numpy(), rel=rel, name="hvp level-parallel vs lane-serial, max_level")
    assert_close(outs["1"][0], oracle.lotd_bwd_bwd_dx(m_ref, v, g, x, p), name="d(dL/dx)/dx level-parallel")


@pytest.mark.parametrize("case", ["ngp_small", "hash_npow2", "dense_2d", "hash_4d", "dense_f8"])
def test_grid_index_bit_exact(oracle, dev, case):
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, seed=3)
    gi = _lotd.lod_get_grid_index(m, xt)
    assert gi.dtype == torch.int64
    assert_equal(gi, oracle.lotd_grid_index(m_ref, x), name="grid_inds")
    # reference smoke test (lotd/tests/unit_test_grid_inds.py:31-40): zeroing the indexed params zeroes y
    p2 = pt.clone()
    p2[gi.reshape(-1)] = 0
    y, _ = _lotd.lod_fwd(m, xt, p2)
    assert float(y.abs().max()) == 0.0


BINNED_CASES = ["ngp_small", "ngp_smooth", "hash_npow2", "dense_f8", "dense_2d", "hash_4d",
                "mixed", "mixed_cuboid", "mixed_smooth", "cp_only_2d4d", "cp_only_4d", "vecz_nplanemul",
                "nplane", "nplane_smooth", "cp_2d", "cp_4d", "nplane_4d"]


@pytest.mark.parametrize("case", BINNED_CASES)
def test_dparam_atomic_and_binned_paths_agree(oracle, dev, case):
    """the default (binned, fp64 LDS accumulation) and the hardware-atomic scatter must both match the oracle,
    including a point count that is not a multiple of the 512-point bin blocks and spans several of them"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=5003, seed=9)
    assert _lotd._dparam_workspace(m, 5003, dev)[1] > 0, "this meta must take the atomic-free path"
    ref1 = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    ref2 = oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True)
    for binned in (True, False):
        _lotd.USE_BINNED_DPARAM = binned
        try:
            _, dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
            _, dp2, _ = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False,
                                                need_dLdinput_dparams=True, need_dLdinput_dinput=False)
        finally:
            _lotd.USE_BINNED_DPARAM = True
        # hardware fp32 atomics add thousands of terms per entry of a small table in an arbitrary order (as the reference
        # does): their rounding noise is a few 1e-6 of the level's largest gradient and run-dependent, so that path gets 5e-5
        tol = 1e-5 if binned else 5e-5
        assert_close(dp, ref1, rel=tol, name=f"dL_dparam binned={binned}", levels=m_ref)
        assert_close(dp2, ref2, rel=tol, name=f"2nd dparam binned={binned}", levels=m_ref)
        # max_level restricts the scatter to the coarse levels
        _lotd.USE_BINNED_DPARAM = binned
        try:
            _, dp3 = _lotd.lod_bwd(m, gt, xt, pt, None, max_level=0, need_input_grad=False, need_param_grad=True)
        finally:
            _lotd.USE_BINNED_DPARAM = True
        assert_close(dp3, oracle.lotd_bwd_dparam(m_ref, g, x, p, max_level=0, accum_double=True), rel=tol, name="max_level=0", levels=m_ref)


@pytest.mark.parametrize("case", ["ngp_small", "mixed", "dense_2d", "nplane"])
def test_dparam_level_buckets(oracle, dev, case, bin_mode):
    """dL/dparam computed in level buckets (nr3d_lotd_bwd_dparam_levels; the data-parallel path reduces a finished
    bucket while the next one is accumulated): the buckets together are the one-call gradient (to fp32 rounding of the
    fp64 partial sums: how a hot table slice is split over workgroups follows the call's records), every bucket callback
    sees exactly its levels' slice, and levels outside the buckets stay zero"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=5003, seed=11)
    L = m.n_levels
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    for binned in (True, False):
        _lotd.USE_BINNED_DPARAM = binned
        try:
            dx0, dp0 = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
            cut = max(1, L // 2)
            buckets = [(cut, L - 1), (0, cut - 1)] if L > 1 else [(0, 0)]
            seen = []
            dx1, dp1 = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True, level_buckets=buckets,
                                     on_bucket=lambda k, sl: seen.append((k, sl.data_ptr(), sl.numel(), sl.clone())))
            _, dp2 = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True,
                                   level_buckets=[(L - 1, L - 1)])
        finally:
            _lotd.USE_BINNED_DPARAM = True
        assert_equal(dx1, dx0.cpu().numpy(), name="dL_dx")
        ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
        assert_close(dp1, ref, name=f"bucketed dL_dparam binned={binned}", levels=m_ref)
        assert_close(dp1, dp0.cpu().numpy(), rel=1e-6 if binned else 1e-5, name="bucketed vs one-call dL_dparam", levels=m_ref)
        assert [k for k, *_ in seen] == list(range(len(buckets)))
        for (k, ptr, numel, snap), (lo, hi) in zip(seen, buckets):
            a, b = m.level_offsets[lo], m.level_offsets[hi + 1]
            assert ptr == dp1.data_ptr() + 4 * a and numel == b - a
            assert torch.equal(snap, dp1[a:b]), "a bucket's slice is final when its callback runs (stream order)"
        a = m.level_offsets[L - 1]
        assert not dp2[:a].any(), "levels outside the buckets stay untouched"
        assert_close(dp2[a:], dp1[a:].cpu().numpy(), rel=1e-6 if binned else 1e-5, name="single-level bucket")
    with pytest.raises(RuntimeError, match="overlapping"):
        _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True, level_buckets=[(0, 0), (0, L - 1)])
    # an empty shard or max_level = -1 still announces every bucket (zeros): the ranks' collectives stay matched
    for kw in (dict(x=xt[:0], g=gt[:0], max_level=None), dict(x=xt, g=gt, max_level=-1)):
        seen = []
        _, dpz = _lotd.lod_bwd(m, kw["g"], kw["x"], pt, None, max_level=kw["max_level"], need_input_grad=False,
                               need_param_grad=True, level_buckets=buckets, on_bucket=lambda k, sl: seen.append((k, sl.numel())))
        assert [k for k, _ in seen] == list(range(len(buckets))) and sum(n_ for _, n_ in seen) == m.n_params
        assert not dpz.any()


def test_grid_index_rejects_other_types(oracle, dev):
    _lotd, m_ref, m, _, (xt, *_r) = _setup(oracle, dev, "mixed")
    with pytest.raises(RuntimeError, match="Only support Dense/Hash"):
        _lotd.lod_get_grid_index(m, xt)


@pytest.mark.parametrize("case", ["ngp_small", "mixed"])
def test_batched_and_max_level(oracle, dev, case, bin_mode):
    B = 3
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=2400, seed=4, n_batch=B)
    rng = np.random.default_rng(5)
    bi = rng.integers(-1, B, x.shape[0]).astype(np.int64)       # -1 => skipped point
    bo = (np.array([2, 0, 1]) * m.n_params).astype(np.int64)    # permuted batch placement
    bit, bot = torch.from_numpy(bi).to(dev), torch.from_numpy(bo).to(dev)
    assert _lotd._dparam_workspace(m, x.shape[0], dev, B)[1] > 0      # batched params also scatter atomic-free
    for kw_ref, kw in [
        (dict(batch_inds=bi), dict(batch_inds=bit)),
        (dict(batch_inds=bi, batch_offsets=bo), dict(batch_inds=bit, batch_offsets=bot)),
        (dict(batch_data_size=800), dict(batch_data_size=800)),
        (dict(max_level=2), dict(max_level=2)),
        (dict(batch_data_size=800, max_level=0), dict(batch_data_size=800, max_level=0)),
    ]:
        y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True, **kw_ref)
        y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True, **kw)
        assert_close(y, y_ref, name=f"y {kw_ref.keys()}")
        assert_close(j, j_ref, name=f"dy_dx {kw_ref.keys()}")
        if "batch_inds" in kw_ref:
            assert float(y[bit < 0].abs().max()) == 0.0
        _, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=False, need_param_grad=True, **kw)
        assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True, **kw_ref), name="dL_dparam", levels=m_ref)
        _lotd.USE_BINNED_DPARAM = False               # the hardware-atomic scatter on the same batched call
        try:
            _, dp_a = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=False, need_param_grad=True, **kw)
        finally:
            _lotd.USE_BINNED_DPARAM = True
        assert_close(dp_a, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True, **kw_ref), name="dL_dparam (atomic)", levels=m_ref)
        _, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, j, need_dLdinput_ddLdoutput=False,
                                              need_dLdinput_dparams=True, need_dLdinput_dinput=True, **kw)
        assert_close(dp2, oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True, **kw_ref), name="2nd dparam", levels=m_ref)
        assert_close(dx2, oracle.lotd_bwd_bwd_dx(m_ref, v, g, x, p, **kw_ref), name="2nd dx")
    # max_level <= -1: all-zero outputs of the documented shapes (lotd_torch_api.cu:294-297)
    y, j = _lotd.lod_fwd(m, xt, pt, max_level=-1, need_input_grad=True)
    assert tuple(y.shape) == (x.shape[0], m.n_encoded_dims) and float(y.abs().max()) == 0
    assert tuple(j.shape) == (x.shape[0], m.n_encoded_dims * m.n_dims_to_encode)


def test_generic_kernel_agrees_with_dense_hash_kernel(oracle, dev):
    """c_hash_only=False routes a Dense/Hash meta through the all-types kernel"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, "ngp_small", seed=6)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    m.c_hash_only = False
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert_close(y, y_ref, name="y generic")
    assert_close(j, j_ref, name="dy_dx generic")
    _, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=False, need_param_grad=True)
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dparam generic", levels=m_ref)


def test_half_table_copy_is_never_stale(oracle, dev, monkeypatch):
    """NATIVE_HALF off (A/B switch; the kernels read half tables themselves otherwise): half tables run on an fp32 copy made
    per call (bindings._lotd._p32; no cache since round 4).  An in-place update must be seen -- through the tensor itself and
    through a `.data` view, whose version counter stays 0 (what LoTDEncoding.inference_param passes) -- and so must a NEW
    tensor that the caching allocator places at the address of a freed one"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, "mixed", n=2001, seed=9)
    monkeypatch.setattr(_lotd, "NATIVE_HALF", False)
    ph = pt.half()
    assert not _lotd._native_half(m, ph, False)
    y1 = _lotd.lod_fwd(m, xt, ph)[0].float().clone()
    yd1 = _lotd.lod_fwd(m, xt, ph.data)[0].float().clone()                   # .data: version counter 0, and it stays 0
    assert torch.equal(yd1, y1)
    assert_close(y1, oracle.lotd_fwd(m_ref, x, ph.float().cpu().numpy())[0], rel=1e-3, name="y half (copy)")
    ph.mul_(2.0)                                                             # an optimizer step: the version changes
    y2 = _lotd.lod_fwd(m, xt, ph)[0].float()
    # values scale with the tables level by level (Dense: x 2, VM: x 4, CP: x 8): nothing may be left as it was
    assert ((y2 - y1).abs() > 1e-3 * y1.abs()).float().mean() > 0.9
    assert_close(y2, oracle.lotd_fwd(m_ref, x, ph.float().cpu().numpy())[0], rel=1e-3, name="y half after an in-place update")
    assert torch.equal(_lotd.lod_fwd(m, xt, ph.data)[0].float(), y2)         # the update is seen through .data as well
    addr = ph.data_ptr()
    del ph
    other = (pt * 0.5).half()                                                # likely the freed block again, version 0
    y3 = _lotd.lod_fwd(m, xt, other)[0].float()
    assert_close(y3, oracle.lotd_fwd(m_ref, x, other.float().cpu().numpy())[0], rel=1e-3, name=f"y of a new table (same address: {other.data_ptr() == addr})")


HALF_GENERAL_CASES = ["hash_npow2", "dense_f8", "dense_2d", "hash_4d", "mixed", "mixed_cuboid", "mixed_smooth", "vecz_nplanemul",
                      "nplane", "nplane_smooth", "cp_2d", "cp_4d", "nplane_4d"]


@pytest.mark.parametrize("case", HALF_GENERAL_CASES)
def test_half_tables_general_kernel_forward(oracle, dev, case):
    """half tables of metas the two-lane kernels do not serve (every level type, D = 2 / 3 / 4, wide pseudo levels): the
    general kernel reads the half entries itself (k_fwd<..., __half>) -- same fp32 arithmetic on the same values as the
    run on an fp32 copy of the table, one rounding of y to half: identical y and dy_dx; also with batched tables"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=3001, seed=12)
    ph = pt.half()
    outs = {}
    for native in (True, False):
        _lotd.NATIVE_HALF = native
        try:
            outs[native] = (_lotd.lod_fwd(m, xt, ph, need_input_grad=True), _lotd.lod_fwd(m, xt, ph, max_level=m.n_levels // 2))
        finally:
            _lotd.NATIVE_HALF = True
    (y, j), (ym, _) = outs[True]
    (y0, j0), (ym0, _) = outs[False]
    assert y.dtype == torch.float16 and j.dtype == torch.float32
    assert torch.equal(y, y0) and torch.equal(j, j0) and torch.equal(ym, ym0)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, ph.float().cpu().numpy(), need_dydx=True)
    # one rounding to half: 2^-11 relative, or half a subnormal step (2^-25) for the products of a 4-factor CP level
    assert np.allclose(y.float().cpu().numpy(), y_ref, rtol=1e-3, atol=6e-8), "y from half tables"
    assert_close(j.reshape(j_ref.shape), j_ref, name="dy_dx from half tables")
    # batched tables: three copies of the table, permuted placement, skipped points
    B = 3
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=1500, seed=13, n_batch=B)
    ph = pt.half()
    rng = np.random.default_rng(6)
    bit = torch.from_numpy(rng.integers(-1, B, x.shape[0]).astype(np.int64)).to(dev)
    bot = torch.from_numpy((np.array([2, 0, 1]) * m.n_params).astype(np.int64)).to(dev)
    for kw in (dict(batch_inds=bit), dict(batch_inds=bit, batch_offsets=bot), dict(batch_data_size=500)):
        yb, jb = _lotd.lod_fwd(m, xt, ph, need_input_grad=True, **kw)
        _lotd.NATIVE_HALF = False
        try:
            yb0, jb0 = _lotd.lod_fwd(m, xt, ph, need_input_grad=True, **kw)
        finally:
            _lotd.NATIVE_HALF = True
        assert yb.dtype == torch.float16 and torch.equal(yb, yb0) and torch.equal(jb, jb0), f"batched {list(kw)}"


@pytest.mark.parametrize("case", ["hash_npow2", "dense_2d", "mixed", "mixed_smooth", "vecz_nplanemul", "nplane", "nplane_smooth", "cp_4d",
                                  "ngp_smooth"])
@pytest.mark.parametrize("binned", [True, False])
def test_half_tables_general_kernels_gradients(oracle, dev, case, binned, monkeypatch):
    """dL/dparam (record path and hardware atomics) and the three second-order outputs from half tables: the kernels that read
    table entries (product-type factors in stage A / k_cp_direct / the atomic scatter; every level in the Hessian kernels)
    read the half entries themselves -- identical to the run on an fp32 copy of the table; batched tables too"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=2777, seed=21)
    monkeypatch.setattr(_lotd, "USE_BINNED_DPARAM", binned)
    ph, gh = pt.half(), gt.half()

    def run():
        _, j = _lotd.lod_fwd(m, xt, ph, need_input_grad=True)
        dx, dp = _lotd.lod_bwd(m, gh, xt, ph, j, need_input_grad=True, need_param_grad=True)
        ddy, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gh, xt, ph, j, need_dLdinput_ddLdoutput=True, need_dLdinput_dparams=True,
                                                need_dLdinput_dinput=True)
        return dx, dp, ddy, dp2, dx2
    native = run()
    _lotd.NATIVE_HALF = False
    try:
        copied = run()
    finally:
        _lotd.NATIVE_HALF = True
    atomics = not _lotd.USE_BINNED_DPARAM            # arrival-order fp32 sums: compared to tolerance, not bit for bit
    for k, (a, b) in enumerate(zip(native, copied)):
        assert a.dtype == b.dtype
        if atomics and k in (1, 3):
            assert_close(a.float(), b.float().cpu().numpy(), rel=1e-3, name=f"output {k}", levels=m_ref)
        else:
            assert torch.equal(a, b), f"output {k} differs between the half table and its fp32 copy"
    p_r, g_r = ph.float().cpu().numpy(), gh.float().cpu().numpy()
    assert_close(native[1].float(), oracle.lotd_bwd_dparam(m_ref, g_r, x, p_r, accum_double=True), rel=1e-3, name="dL_dparam", levels=m_ref)
    assert_close(native[4], oracle.lotd_bwd_bwd_dx(m_ref, v, g_r, x, p_r), rel=2e-5, name="d(dL/dx)/dx from half tables")
    if case in ("mixed", "hash_npow2"):
        B = 2
        _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=1200, seed=22, n_batch=B)
        ph, gh = pt.half(), gt.half()
        kw = dict(batch_data_size=600)
        outs = []
        for nat in (True, False):
            _lotd.NATIVE_HALF = nat
            try:
                _, j = _lotd.lod_fwd(m, xt, ph, need_input_grad=True, **kw)
                _, dp = _lotd.lod_bwd(m, gh, xt, ph, j, need_input_grad=False, need_param_grad=True, **kw)
                _, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gh, xt, ph, j, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                                      need_dLdinput_dinput=True, **kw)
            finally:
                _lotd.NATIVE_HALF = True
            outs.append((dp, dp2, dx2))
        for k, (a, b) in enumerate(zip(*outs)):
            if atomics and k < 2:
                assert_close(a.float(), b.float().cpu().numpy(), rel=1e-3, name=f"batched output {k}")
            else:
                assert torch.equal(a, b), f"batched output {k}"


@pytest.mark.parametrize("case", ["ngp_small", "ngp_pair", "pair_f4", "mixed"])
def test_half_params_and_inputs(oracle, dev, case):
    """fp16 storage, the reference's (float, half, float) type combination: params / y / dL_dy / dL_dparam half, x and
    dy_dx float, arithmetic in fp32.  Dense/Hash metas with 2-feature pseudo levels are served NATIVELY (the kernels read
    half table entries and gradients and write half results: no whole-table conversion), other metas through fp32
    copies; both are compared with the fp32 oracle on the rounded inputs -- y and dL_dparam to half precision, dy_dx /
    dL_dx (fp32 outputs of fp32 arithmetic on the rounded tables) to the fp32 tolerance -- and with each other."""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=5003, seed=7)
    ph, gh = pt.half(), gt.half()
    p_r, g_r = ph.float().cpu().numpy(), gh.float().cpu().numpy()
    native_expected = case != "mixed"
    assert bool(_lotd._native_half(m, ph, False)) == native_expected
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p_r, need_dydx=True)
    dp_ref = oracle.lotd_bwd_dparam(m_ref, g_r, x, p_r, accum_double=True)
    dx_ref = oracle.lotd_bwd_dx(m_ref, g_r, j_ref)
    outs = {}
    for native in (True, False):
        _lotd.NATIVE_HALF = native
        try:
            y, j = _lotd.lod_fwd(m, xt, ph, need_input_grad=True)
            y0, _ = _lotd.lod_fwd(m, xt, ph, need_input_grad=False)
            dx, dp = _lotd.lod_bwd(m, gh, xt, ph, j, need_input_grad=True, need_param_grad=True)
            _, dp_only = _lotd.lod_bwd(m, gh, xt, ph, None, need_input_grad=False, need_param_grad=True)     # no feature-major copy
        finally:
            _lotd.NATIVE_HALF = True
        assert y.dtype == torch.float16 and j.dtype == torch.float32 and dp.dtype == torch.float16 and dx.dtype == torch.float32
        assert torch.equal(y, y0)
        assert_close(y.float(), y_ref, rel=1e-3, name=f"y half native={native}")                   # one rounding to half: 2^-11
        assert_close(j.reshape(j_ref.shape), j_ref, name=f"dy_dx native={native}")
        assert_close(dx, dx_ref, name=f"dL_dx native={native}")
        assert_close(dp.float(), dp_ref, rel=1e-3, name=f"dL_dparam half native={native}", levels=m_ref)
        assert_close(dp_only.float(), dp_ref, rel=1e-3, name=f"dL_dparam half (row-major dL_dy) native={native}", levels=m_ref)
        outs[native] = (y, j, dx, dp)
    if native_expected:      # same fp32 arithmetic on the same rounded values, one rounding of the result: identical halves
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
        assert torch.equal(outs[True][2], outs[False][2])
        assert_close(outs[True][3].float(), outs[False][3].float().cpu().numpy(), rel=1e-3, name="native vs converted dL_dparam", levels=m_ref)
    else:                    # general kernels on the half table itself against the run on its fp32 copy: the same bits
        for a, b in zip(outs[True], outs[False]):
            assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="not supported"):
        _lotd.lod_fwd(m, xt.half(), pt)          # (half input, float params) is not a supported combination


@pytest.mark.parametrize("case,half", [("ngp_small", False), ("ngp_smooth", False), ("ngp_pair", False), ("ngp_pair", True), ("pair_f4", False)])
def test_bwd_direct_levels(oracle, dev, case, half, hip_option):
    """levels with <= 4 buckets bypass the records (k_pair_direct: LDS accumulation straight from x and dL_dy, default) --
    against the all-records route (option pair_direct = 0; same per-update arithmetic and fixed-point sums, the two differ only
    in how a replicated bucket's fp32 partial tables are split) and the oracle; plain and level-bucketed calls."""
    n = 70001
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=n, seed=13)
    if half:
        pt, gt = pt.half(), gt.half()
        p, g = pt.float().cpu().numpy(), gt.float().cpu().numpy()
    hip_option("pair_direct", "1")
    _, d1 = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    _, d1b = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    _, d1m = _lotd.lod_bwd(m, gt, xt, pt, None, max_level=1, need_input_grad=False, need_param_grad=True)
    hip_option("pair_direct", "0")
    _, d0 = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    _, d0m = _lotd.lod_bwd(m, gt, xt, pt, None, max_level=1, need_input_grad=False, need_param_grad=True)
    assert torch.equal(d1, d1b)                                        # reproducible
    ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    tol = 1e-3 if half else 1e-5
    assert_close(d1.float(), ref, rel=tol, name="dL_dparam direct levels", levels=m_ref)
    assert_close(d1.float(), d0.float().cpu().numpy(), rel=1e-3 if half else 1e-6, name="direct vs records", levels=m_ref)
    assert_close(d1m.float(), d0m.float().cpu().numpy(), rel=1e-3 if half else 1e-6, name="direct vs records, max_level=1", levels=m_ref)
    assert_close(d1m.float(), oracle.lotd_bwd_dparam(m_ref, g, x, p, max_level=1, accum_double=True), rel=tol,
                 name="dL_dparam direct levels, max_level=1", levels=m_ref)


def test_autograd_surface(oracle, dev):
    """LoTDFunction / FwdDydx / BwdDydx: loss_scale handling, clamping, prefix dims, second-order chain"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTD, generate_meta
    D, res, nf, types, T, smooth = LOTD_CASES["ngp_smooth"]
    enc = LoTD(D, res, nf, types, hashmap_size=T, use_smooth_step=smooth, dtype=torch.float)
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    x, p, g, v = lotd_inputs(m_ref.as_dict(), 512, 8)
    xt = torch.from_numpy(x).to(dev).view(8, 64, D).requires_grad_(True)
    pt = torch.from_numpy(p).to(dev).requires_grad_(True)
    gt = torch.from_numpy(g).to(dev).view(8, 64, -1)
    # first order
    y = enc(xt, pt)
    assert tuple(y.shape) == (8, 64, enc.out_features)
    y.backward(gt)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    assert_close(y.reshape(512, -1), y_ref, name="y")
    assert_close(xt.grad.reshape(512, D), oracle.lotd_bwd_dx(m_ref, g, j_ref), name="x.grad")
    assert_close(pt.grad, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="params.grad")
    # second order: nablas = dL_dy . dy_dx, loss = <nablas, v>
    xt2 = torch.from_numpy(x).to(dev).requires_grad_(True)
    pt2 = torch.from_numpy(p).to(dev).requires_grad_(True)
    gt2 = torch.from_numpy(g).to(dev).requires_grad_(True)
    y2, dy_dx = enc.forward_dydx(xt2, pt2)
    nablas = enc.backward_dydx(gt2, dy_dx, xt2, pt2)
    assert_close(nablas, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="nablas")
    vt = torch.from_numpy(v).to(dev)
    (nablas * vt).sum().backward()
    assert_close(gt2.grad, oracle.lotd_bwd_bwd_ddLdy(m_ref, v, j_ref), name="dL/d(dL_dy)")
    assert_close(pt2.grad, oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="2nd params.grad")
    # points outside [0,1] are clamped, not rejected
    y_out = enc(torch.tensor([[-0.5, 0.5, 1.5]], device=dev), pt.detach())
    y_cl = enc(torch.tensor([[1e-6, 0.5, 1 - 1e-6]], device=dev), pt.detach())
    assert torch.equal(y_out, y_cl)


def test_errors(oracle, dev):
    from nr3d_lib_amd.bindings import _lotd
    m = _lotd.LoDMeta(3, [8, 16], [2, 2], ["Dense", "Hash"], 1024)
    x = torch.rand(10, 3, device=dev)
    p = torch.zeros(m.n_params, device=dev)
    with pytest.raises(RuntimeError, match="integral multiple"):
        _lotd.lod_fwd(m, x, p[:-1])
    with pytest.raises(RuntimeError, match="divisor"):
        _lotd.lod_fwd(m, x, p, batch_data_size=3)
    with pytest.raises(RuntimeError, match="need `dy_dx`"):
        _lotd.lod_bwd(m, torch.zeros(10, 4, device=dev), x, p, None, need_input_grad=True)
    with pytest.raises(RuntimeError):
        _lotd.lod_fwd(m, x.cpu(), p.cpu())           # no CPU fallback
    with pytest.raises(RuntimeError, match="resolutions >= 3"):
        _lotd.LoDMeta(3, [2], [2], ["Dense"])
    with pytest.raises(RuntimeError, match="greatest common divisor"):
        _lotd.LoDMeta(3, [8], [3], ["Dense"])
    with pytest.raises(RuntimeError, match="hashmap_size"):
        _lotd.LoDMeta(3, [8], [2], ["Hash"])
    with pytest.raises(RuntimeError, match="3D"):
        _lotd.LoDMeta(2, [8], [2], ["VM"])


def test_lotd_encoding_module(oracle, dev):
    """LoTDEncoding: inputs in [-1, 1], halved nablas, max_level / window masking, gradients reach flattened_params"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding
    D, res, nf, types, T, smooth = LOTD_CASES["ngp_small"]
    cfg = dict(lod_res=res, lod_n_feats=nf, lod_types=types, hashmap_size=T)
    torch.manual_seed(1)
    enc = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, device=dev, param_init_cfg={"type": "uniform", "bound": 0.5})
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    rng = np.random.default_rng(4)
    x = rng.uniform(-0.999, 0.999, (2000, 3)).astype(np.float32)
    p = enc.flattened_params.detach().cpu().numpy()
    x01 = (x / np.float32(2.0) + np.float32(0.5)).astype(np.float32)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x01, p, need_dydx=True)
    xt = torch.from_numpy(x).to(dev)
    y = enc(xt)
    assert_close(y, y_ref, name="forward")
    y2, dy_dx = enc.forward_dydx(xt)
    g = rng.standard_normal(y_ref.shape).astype(np.float32)
    nablas = enc.backward_dydx(torch.from_numpy(g).to(dev), dy_dx, xt)
    assert_close(nablas, oracle.lotd_bwd_dx(m_ref, g, j_ref) / 2, name="nablas (halved)")
    y.backward(torch.from_numpy(g).to(dev))
    assert_close(enc.flattened_params.grad, oracle.lotd_bwd_dparam(m_ref, g, x01, p, accum_double=True), name="param grad")
    enc.max_level = 2
    y3 = enc(xt)
    assert_close(y3, oracle.lotd_fwd(m_ref, x01, p, max_level=2)[0], name="max_level")
    enc.max_level, enc.window = None, torch.linspace(0, 1, enc.out_features, device=dev)
    assert_close(enc(xt), y_ref * np.linspace(0, 1, enc.out_features, dtype=np.float32), name="window")


def test_lotd_batched_module(oracle, dev):
    """LoTDBatched: grower -> per-batch tables; batched inputs and flat inputs + bidx; gradients reach the grower"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDBatched
    D, res, nf, types, T, smooth = LOTD_CASES["mixed_smooth"]
    cfg = dict(lod_res=res, lod_n_feats=nf, lod_types=types, hashmap_size=T, use_smooth_step=smooth)
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    n_params, B, zdim = m_ref.as_dict()["n_params"], 3, 5
    torch.manual_seed(2)
    grower = torch.nn.Linear(zdim, n_params).to(dev)
    enc = LoTDBatched(3, lotd_cfg=cfg, grower=grower, device=dev)
    z = torch.randn(B, zdim, device=dev)
    enc.grow(z)
    p = enc.lod_params.detach().cpu().numpy()
    rng = np.random.default_rng(8)
    x = rng.uniform(-0.99, 0.99, (B, 700, 3)).astype(np.float32)
    x01 = (x / np.float32(2) + np.float32(0.5)).reshape(-1, 3)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x01, p, batch_data_size=700, need_dydx=True)
    xt = torch.from_numpy(x).to(dev)
    y = enc(xt)
    assert tuple(y.shape) == (B, 700, enc.out_features)
    assert_close(y.reshape(-1, enc.out_features), y_ref, name="batched input")
    bidx = rng.integers(0, B, 1500)
    xf = rng.uniform(-0.99, 0.99, (1500, 3)).astype(np.float32)
    yf = enc(torch.from_numpy(xf).to(dev), torch.from_numpy(bidx).to(dev))
    yf_ref, jf_ref = oracle.lotd_fwd(m_ref, xf / np.float32(2) + np.float32(0.5), p, batch_inds=bidx, need_dydx=True)
    assert_close(yf, yf_ref, name="flat input + bidx")
    g = rng.standard_normal(yf_ref.shape).astype(np.float32)
    yf.backward(torch.from_numpy(g).to(dev))
    dp_ref = oracle.lotd_bwd_dparam(m_ref, g, xf / np.float32(2) + np.float32(0.5), p, batch_inds=bidx, accum_double=True)
    # d loss / d grower.weight = (dL/dparams [B, n_params])^T z
    want_w = dp_ref.reshape(B, n_params).T.astype(np.float64) @ z.cpu().numpy().astype(np.float64)
    assert_close(grower.weight.grad, want_w.astype(np.float32), rel=1e-4, name="grower weight grad")
    y2, dy_dx = enc.forward_dydx(torch.from_numpy(xf).to(dev), torch.from_numpy(bidx).to(dev))
    nablas = enc.backward_dydx(torch.from_numpy(g).to(dev), dy_dx, torch.from_numpy(xf).to(dev), torch.from_numpy(bidx).to(dev))
    assert_close(nablas, oracle.lotd_bwd_dx(m_ref, g, jf_ref) / 2, name="nablas")
    enc.clear()
    assert not hasattr(enc, "lod_params")


@pytest.mark.parametrize("case", ["ngp_small", "mixed"])
def test_dparam_multi_pass_chunking(oracle, dev, hiplib, case, bin_mode):
    """the atomic-free scatter in several passes (2^10-point chunks over 5003 points: 4 full + 1 partial pass), with the
    feature-major dL_dy handed over by the dL/dx kernel (strided per pass) and with a plain row-major dL_dy"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=5003, seed=12)
    ref1 = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    ref2 = oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True)
    hiplib.nr3d_lotd_set_dparam_chunk_log2(10)
    try:
        _, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
        dx, dp = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)       # fused gT path
        _, dp_b = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)  # row-major dL_dy
        _, dp2, _ = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False,
                                            need_dLdinput_dparams=True, need_dLdinput_dinput=False)
    finally:
        hiplib.nr3d_lotd_set_dparam_chunk_log2(0)
    assert_close(dp, ref1, name="dL_dparam (fused, chunked)", levels=m_ref)
    assert_close(dp_b, ref1, name="dL_dparam (chunked)", levels=m_ref)
    assert_close(dp2, ref2, name="2nd-order dparam (chunked)", levels=m_ref)


@pytest.mark.parametrize("case", ["ngp_small", "mixed", "nplane"])
def test_dparam_coherent_points(oracle, dev, case, bin_mode):
    """samples along rays: long runs of consecutive points in the same coarse cell take stage A's wave-level merge
    (their updates are summed into the first lane of the run before they are binned)"""
    _lotd, m_ref, m, _, _ = _setup(oracle, dev, case, n=8)
    rng = np.random.default_rng(31)
    n_rays, n_per = 48, 160
    o = rng.uniform(0.05, 0.3, (n_rays, 1, 3)).astype(np.float32)
    d = rng.uniform(0.2, 0.65, (n_rays, 1, 3)).astype(np.float32)
    t = np.linspace(0, 1, n_per, dtype=np.float32).reshape(1, n_per, 1)
    x = (o + d * t).reshape(-1, 3).clip(1e-6, 1 - 1e-6).astype(np.float32)        # 7680 points, ray after ray
    x[5::97] = x[4::97][: len(x[5::97])]                                            # some exact duplicates too
    md = m_ref.as_dict()
    p = (rng.standard_normal(md["n_params"]) * 0.1).astype(np.float32)
    g = (rng.standard_normal((x.shape[0], md["n_encoded_dims"])) * 0.1).astype(np.float32)
    v = rng.standard_normal((x.shape[0], 3)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    _, dp = _lotd.lod_bwd(m, T(g), T(x), T(p), None, need_input_grad=False, need_param_grad=True)
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam (coherent)", levels=m_ref)
    _, dp2, _ = _lotd.lod_bwd_bwd_input(m, T(v), T(g), T(x), T(p), None, need_dLdinput_ddLdoutput=False,
                                        need_dLdinput_dparams=True, need_dLdinput_dinput=False)
    assert_close(dp2, oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True), name="2nd-order dparam (coherent)", levels=m_ref)


@pytest.mark.parametrize("n", [512, 1024, 4096])
def test_dparam_small_coherent_batch_constant_gradient(oracle, dev, n, bin_mode):
    """Fixed-point accumulators under the coherent-lane merge (round-2 advisor finding): for small batches the scale left
    single updates just below 2^51, and a run head carries the fp32 SUM of up to 64 merged lanes -- with a constant dL_dy
    and ~36 consecutive points per coarse cell the merged value left the binade of the mantissa rounding trick and came
    out wrong.  Zero-mean random gradients (test_dparam_coherent_points) never reach that; ones do."""
    _lotd, m_ref, m, _, _ = _setup(oracle, dev, "ngp_small", n=8)
    n_per = 128
    rng = np.random.default_rng(77)
    o = rng.uniform(0.1, 0.2, (n // n_per, 1, 3)).astype(np.float32)
    d = rng.uniform(0.05, 0.12, (n // n_per, 1, 3)).astype(np.float32)                # short rays: ~40 points per level-0 cell
    t = np.linspace(0, 1, n_per, dtype=np.float32).reshape(1, n_per, 1)
    x = (o + d * t).reshape(-1, 3).clip(1e-6, 1 - 1e-6).astype(np.float32)
    md = m_ref.as_dict()
    p = (rng.standard_normal(md["n_params"]) * 0.1).astype(np.float32)
    g = np.ones((n, md["n_encoded_dims"]), np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    _, dp = _lotd.lod_bwd(m, T(g), T(x), T(p), None, need_input_grad=False, need_param_grad=True)
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam (coherent, constant g)", levels=m_ref)
    # gradient mass: the interpolation weights of a point sum to 1 on every level
    tot = dp.double().cpu().numpy()
    for off, size, F in zip(md["level_offsets"], md["level_sizes"], md["level_n_feats"]):
        assert abs(tot[off:off + size * F].sum() - n * F) <= 1e-5 * n * F


@pytest.mark.parametrize("case", ["ngp_small", "mixed", "cp_2d", "hash_4d"])
def test_dparam_survives_points_outside_the_unit_cube(oracle, dev, bin_mode, case):
    """x outside [0, 1] (the Python layer clamps, lotd.py:68, but the C ABI takes what it is given): a pair whose two
    entries do not share a bucket of the level (or whose bucket lies past the table) is dropped by the record path instead
    of indexing its LDS histogram / accumulators out of bounds (round-2 advisor finding); since round 4 the generic stage A and
    the direct kernels (k_cp_direct, k_vm_direct) drop a point whose cell lies outside the level too (cases "mixed", "cp_2d", "hash_4d").  What such points add elsewhere
    is unspecified (the reference wraps / hashes them); the call must complete with finite values and leave the
    device in a state where the next, valid, call is exact."""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, seed=5)
    D = x.shape[1]
    x_bad = x.copy()
    x_bad[::7] = np.array([1.5, -0.25, 3.0, 0.5], np.float32)[:D]
    x_bad[3::11] = np.array([0.5, 40.0, 0.5, -7.0], np.float32)[:D]
    _, dp_bad = _lotd.lod_bwd(m, gt, torch.from_numpy(x_bad).to(dev), pt, None, need_input_grad=False, need_param_grad=True)
    torch.cuda.synchronize()
    assert torch.isfinite(dp_bad).all()
    _, dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam after a call with outside points", levels=m_ref)


@pytest.mark.parametrize("case", ["ngp_small", "mixed"])
def test_params_at_odd_alignment(oracle, dev, case):
    """params that start 4 bytes into an allocation: the 8 / 16-byte vector gathers must not be used"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, seed=14)
    big = torch.empty(pt.numel() + 1, device=dev)
    p_odd = big[1:]
    p_odd.copy_(pt)
    assert p_odd.data_ptr() % 8 == 4 and p_odd.is_contiguous()
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    y, j = _lotd.lod_fwd(m, xt, p_odd, need_input_grad=True)
    assert_close(y, y_ref, name="y")
    assert_close(j, j_ref, name="dy_dx")
    _, dp = _lotd.lod_bwd(m, gt, xt, p_odd, j, need_input_grad=False, need_param_grad=True)
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL_dparam", levels=m_ref)
    _, _, dx2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, p_odd, j, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=False,
                                        need_dLdinput_dinput=True)
    assert_close(dx2, oracle.lotd_bwd_bwd_dx(m_ref, v, g, x, p), name="2nd-order dx")


def _face_hugging_points(meta_dict, n, seed):
    """points whose locator value fma(x, R - 2, 0.5) sits ON a cell face or within a few ulp of it, in one random (level,
    dim) per point: x = the fp32 number nearest (k - 0.5) / (R - 2), moved by -2 ... +2 ulp.  The other coordinates are
    random.  Which cell such a point belongs to is decided by the last bit of one fma -- the kernels must make the
    reference's decision (lotd_cuda.h:959-1077: floorf(fma(x, scale, 0.5))), not a neighbouring one."""
    rng = np.random.default_rng(seed)
    D = meta_dict["n_dims_to_encode"]
    res = np.asarray(meta_dict["level_res_multidim"])[:, :D]
    x = rng.random((n, D)).astype(np.float32)
    lv = rng.integers(0, res.shape[0], n)
    dim = rng.integers(0, D, n)
    R = res[lv, dim].astype(np.int64)
    k = (rng.random(n) * (R - 2)).astype(np.int64) + 1                    # faces 1 .. R - 2 (inside the unit interval)
    xf = ((k.astype(np.float64) - 0.5) / np.maximum(R - 2, 1)).astype(np.float32)
    ulps = rng.integers(-2, 3, n)
    for _ in range(2):
        xf = np.where(ulps > 0, np.nextafter(xf, np.float32(2)), np.where(ulps < 0, np.nextafter(xf, np.float32(-1)), xf)).astype(np.float32)
        ulps = ulps - np.sign(ulps)
    x[np.arange(n), dim] = xf
    return np.clip(x, 1e-6, 1 - 1e-6).astype(np.float32)


@pytest.mark.parametrize("case", list(LOTD_CASES))
def test_points_on_cell_faces(oracle, dev, case):
    """round-3 review: every other LoTD parity input is resampled AWAY from cell faces (tests/util.py:lotd_inputs) and the
    full-size tests tolerate a few rows in a neighbouring cell, so the bit-for-bit cell decision was exercised only by the
    fuzzers.  Here EVERY point hugs a face of one level (0, 1 or 2 ulp either side): global parameter indices bit-exact
    (Dense / Hash metas, lod_get_grid_index), values and Jacobians against the oracle at the usual 1e-5 -- the Jacobian of a
    linear level jumps across a face, so one wrong cell shows as an O(1) error there -- for the two-lane, LDS-staged and
    generic kernels alike (n = 300 000 on the pair-type metas takes the staged route for the coarse levels)."""
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    from nr3d_lib_amd.bindings import _lotd
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    md = m_ref.as_dict()
    pair_type = D == 3 and all(t in ("Dense", "Hash") for t in types) and m.n_feat_per_pseudo_lvl == 2
    n = 300_000 if pair_type else 20_011
    x = _face_hugging_points(md, n, seed=77)
    # the generator must do what it says: most points within 3 ulp of a face in some (level, dim)
    v = x[:, None, :].astype(np.float64) * (np.asarray(md["level_res_multidim"])[None, :, :D] - 2) + 0.5
    assert (np.abs(v - np.round(v)).min((1, 2)) < 1e-4).mean() > 0.95
    rng = np.random.default_rng(78)
    p = (rng.standard_normal(md["n_params"]) * 0.1).astype(np.float32)
    xt, pt = torch.from_numpy(x).to(dev), torch.from_numpy(p).to(dev)
    if all(t in ("Dense", "Hash") for t in types):
        assert_equal(_lotd.lod_get_grid_index(m, xt), oracle.lotd_grid_index(m_ref, x), name="grid_inds on faces")
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True)
    y, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    assert_close(y, y_ref, name="y on faces")
    assert_close(j.reshape(n, -1, D), j_ref, name="dy_dx on faces")
    # gradient scatter: the entries touched are the cell's corners -- same decision, checked through dL/dparam
    g = (rng.standard_normal((n, m.n_encoded_dims)) * 0.1).astype(np.float32)
    dp = _lotd.lod_bwd(m, torch.from_numpy(g).to(dev), xt, pt, None, need_input_grad=False, need_param_grad=True)[1]
    assert_close(dp, oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True), name="dL/dparam on faces", levels=m_ref)


@pytest.mark.parametrize("case", ["mixed", "mixed_cuboid", "mixed_smooth"])
@pytest.mark.parametrize("points", ["uniform", "slab", "batched"])
def test_vm_levels_over_sorted_points(oracle, dev, case, points, hip_option):
    """VM levels over SORTED points (lotd_sorted.hip; option vm_sorted = 2 takes it whatever the table size, vm_direct = 0 hands it
    every VM level): a band's points are one range of the order of x_a; "slab" puts 40 000 points into two cell rows (one band
    with replicas, added in a fixed order), "batched" three table copies with permuted placement and skipped points.  Against the
    oracle and the record path, first and second order, twice"""
    from nr3d_lib_amd import _hip
    B = 3 if points == "batched" else 1
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=40013, seed=77, n_batch=B)
    kw_ref, kw = {}, {}
    if points == "slab":
        x[:, 0] = (0.43 + 0.02 * x[:, 0]).astype(np.float32)
        x[:, 1] = (0.61 + 0.02 * x[:, 1]).astype(np.float32)
        xt = torch.from_numpy(x).to(dev)
    if points == "batched":
        rng = np.random.default_rng(5)
        bi = rng.integers(-1, B, x.shape[0]).astype(np.int64)
        bo = (np.array([2, 0, 1]) * m.n_params).astype(np.int64)
        kw_ref, kw = dict(batch_inds=bi, batch_offsets=bo), dict(batch_inds=torch.from_numpy(bi).to(dev), batch_offsets=torch.from_numpy(bo).to(dev))
    ref1 = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True, **kw_ref)
    ref2 = oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True, **kw_ref)
    hip_option("vm_direct", 0)
    outs = {}
    for mode in (2, 0, 2):
        hip_option("vm_sorted", mode)
        _hip.prof_enable("lotd_direct")
        try:
            dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True, **kw)[1]
            dp2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                          need_dLdinput_dinput=False, **kw)[1]
            torch.cuda.synchronize()
        finally:
            _hip.prof_enable()
        assert_close(dp, ref1, name=f"dL/dparam vm_sorted={mode}", levels=m_ref)
        assert_close(dp2, ref2, name=f"d(dL/dx)/dparam vm_sorted={mode}", levels=m_ref)
        if mode in outs:
            assert torch.equal(dp, outs[mode][0]) and torch.equal(dp2, outs[mode][1]), "two runs over sorted points differ"
        outs[mode] = (dp, dp2)
    assert_close(outs[2][0], outs[0][0].cpu().numpy(), rel=1e-5, name="sorted vs records", levels=m_ref)


@pytest.mark.parametrize("case", ["mixed", "mixed_cuboid", "mixed_smooth", "cp_2d", "cp_only_2d4d"])
@pytest.mark.parametrize("scale", [1.0, 1e-6, 1e4])
def test_cp_and_vm_levels_without_records(oracle, dev, case, scale, hip_option):
    """CP levels (k_cp_direct) and small VM levels (k_vm_direct) accumulate dL/dparam and d(dL/dx)/dparam in LDS without records:
    64-bit fixed-point accumulators whose scale comes from the workgroup's own bound on its updates (round 4, default) against
    fp64 accumulators (option direct_fixed = 0), against the record path (options cp_direct = vm_direct = 0) and the oracle;
    gradients of very different magnitudes (the scale follows them), a point count that gives several replicas"""
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=40013, seed=61)
    g = (g * scale).astype(np.float32)
    gt = torch.from_numpy(g).to(dev)
    outs = {}
    for mode, (direct, fixed) in (("fixed", (1, 2)), ("fp64", (1, 0)), ("records", (0, 0))):     # 2: k_vm_direct in fixed point too
        hip_option("cp_direct", direct)
        hip_option("vm_direct", direct)
        hip_option("direct_fixed", fixed)
        dp = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)[1]
        dp2 = _lotd.lod_bwd_bwd_input(m, vt, gt, xt, pt, None, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                      need_dLdinput_dinput=False)[1]
        outs[mode] = (dp, dp2)
    ref1 = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    ref2 = oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True)
    for mode in outs:
        assert torch.isfinite(outs[mode][0]).all() and torch.isfinite(outs[mode][1]).all()
        assert_close(outs[mode][0], ref1, name=f"dL/dparam ({mode})", levels=m_ref)
        assert_close(outs[mode][1], ref2, name=f"d(dL/dx)/dparam ({mode})", levels=m_ref)
    assert_close(outs["fixed"][0], outs["fp64"][0].cpu().numpy(), rel=1e-6, name="fixed point vs fp64, first order", levels=m_ref)
    assert_close(outs["fixed"][1], outs["fp64"][1].cpu().numpy(), rel=1e-6, name="fixed point vs fp64, second order", levels=m_ref)
    # all-zero gradients: the bound is zero, the kernels keep their fp64 accumulators and must return zeros
    z = _lotd.lod_bwd(m, torch.zeros_like(gt), xt, pt, None, need_input_grad=False, need_param_grad=True)[1]
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize("smooth,half", [(False, False), (True, False), (False, True)])
def test_forward_lds_slabs_bit_identical(dev, hip_option, smooth, half):
    """the three forward routes of a 3-D Dense / Hash meta -- two lanes per (point, level) through L2 (fwd_lds_stage = 0), coarse
    Dense tables whole in LDS (1, default), and the round-5 experiment that also serves the next Dense levels from LDS slab by slab
    (2) -- give the same bits: which route a level takes depends on the batch size, so it must never show in the result.
    N = 2^19 + 77: above both staging thresholds, a ragged last workgroup; levels of 2 and 6 slabs (30^3, 42^3) and a cuboid one."""
    from nr3d_lib_amd.bindings import _lotd
    res = [16, 22, 30, 42, [36, 20, 50], 58, 111, 212]
    meta = _lotd.LoDMeta(3, res, [2] * 8, ["Dense"] * 6 + ["Hash"] * 2, 2 ** 16, smooth)
    n = (1 << 19) + 77
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    x[:64] = torch.tensor([1e-6, 0.5, 1 - 1e-6])[torch.randint(0, 3, (64, 3), generator=g)].to(dev)      # faces and corners of the box
    params = torch.empty(meta.n_params).uniform_(-1, 1, generator=g).to(dev)
    if half:
        params = params.half()
    out = {}
    for mode in (0, 1, 2):
        hip_option("fwd_lds_stage", mode)
        y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
        y0, _ = _lotd.lod_fwd(meta, x, params, need_input_grad=False)
        assert torch.equal(y, y0), "with and without the Jacobian"
        out[mode] = (y.clone(), j.clone())
    for mode in (1, 2):
        assert torch.equal(out[mode][0], out[0][0]), f"y, fwd_lds_stage = {mode}"
        assert torch.equal(out[mode][1], out[0][1]), f"dy_dx, fwd_lds_stage = {mode}"
    import ctypes
    from nr3d_lib_amd import _hip as H
    H.lib().nr3d_lotd_fwd_lds_levels.restype = ctypes.c_uint64
    by_slab = ctypes.c_uint64(0)
    hip_option("fwd_lds_stage", 2)
    mask = H.lib().nr3d_lotd_fwd_lds_levels(ctypes.byref(meta._cmeta()), ctypes.c_uint32(n), ctypes.byref(by_slab))
    assert mask == 0b011111 and by_slab.value == 0b011100, (bin(mask), bin(by_slab.value))    # 58^3 needs 15 slabs: left to the two-lane kernel


@pytest.mark.parametrize("case", ["ngp_pair", "ngp_small"])
def test_dparam_does_not_depend_on_the_order_of_the_points(oracle, dev, case):
    """the pair-record path on ORDERED inputs (round 6: k_pair_direct walks a coherent wave's updates in per-lane rotated order, the
    replicas of stage B and of the direct levels take interleaved point blocks): the same points along a Morton curve, sorted by x
    (rays), and with every point repeated 64 times in a row (whole waves in one cell) give the gradient of the random order --
    sums are exact in fixed point; what may differ is the fp32 rounding of lanes that stage A merges -- and that of the oracle"""
    from nr3d_lib_amd import _hip
    _lotd, m_ref, m, (x, p, g, v), (xt, pt, gt, vt) = _setup(oracle, dev, case, n=1 << 18, seed=21)
    ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)

    def run(order):
        xo, go = (xt if order is None else xt[order].contiguous()), (gt if order is None else gt[order].contiguous())
        return _lotd.lod_bwd(m, go, xo, pt, None, need_input_grad=False, need_param_grad=True)[1]
    base = run(None)
    assert_close(base, ref, name="random order vs oracle", levels=m_ref)
    morton = _hip.spatial_order(xt.contiguous(), 7).long()
    by_x = torch.argsort(xt[:, 0])
    for name, order in (("Morton", morton), ("sorted by x", by_x)):
        got = run(order)
        assert_close(got, base.cpu().numpy(), rel=2e-6, name=f"{name} vs random order", levels=m_ref)
    # whole waves in one cell: 4096 distinct points, each 64 times in a row
    rep = torch.arange(4096, device=dev).repeat_interleave(64)
    got = _lotd.lod_bwd(m, gt[rep].contiguous(), xt[rep].contiguous(), pt, None, need_input_grad=False, need_param_grad=True)[1]
    want = 64.0 * _lotd.lod_bwd(m, gt[:4096].contiguous(), xt[:4096].contiguous(), pt, None, need_input_grad=False, need_param_grad=True)[1]
    assert_close(got, want.cpu().numpy(), rel=2e-6, name="64 copies of every point in a row", levels=m_ref)


def test_loss_scale_changes_no_bit(dev, monkeypatch):
    """half tables: the reference's loss scale (x 128 on dL/dy, / 128 on the gradients; lotd.py:96-119) protects its half atomics from
    underflow; here the accumulation is exact and every gradient is rounded once, so running the protocol literally (two more passes)
    gives the same bits as skipping it (models/grid_encodings/lotd/lotd.py: APPLY_LOSS_SCALE, off by default)"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTD
    from nr3d_lib_amd.models.grid_encodings.lotd import lotd as lotd_mod
    D, res, nf, types, T, smooth = LOTD_CASES["ngp_small"]
    enc = LoTD(3, res, nf, types, hashmap_size=T, dtype=torch.half)
    assert enc.loss_scale == 128.0
    torch.manual_seed(11)
    x = torch.rand(50001, 3, device=dev)
    gy = (torch.randn(50001, enc.out_features, device=dev) * 0.05).half()
    outs = []
    for apply_scale in (False, True):
        monkeypatch.setattr(lotd_mod, "APPLY_LOSS_SCALE", apply_scale)
        grid = (torch.randn(enc.n_params, device=dev) * 0.1).half().requires_grad_(True) if not outs else outs[0][2].detach().clone().requires_grad_(True)
        xx = x.clone().requires_grad_(True)
        y = enc(xx, grid)
        y.backward(gy)
        outs.append((xx.grad.clone(), grid.grad.clone(), grid))
    assert outs[0][1].dtype == torch.float16 and float(outs[0][1].float().abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]), "dL/dx"
    a, b = outs[0][1].float(), outs[1][1].float()
    normal = (a.abs() >= 2.0 ** -14) & (b.abs() >= 2.0 ** -14) & (b.abs() * 128 < 65504)       # both protocols' half results in the normal range
    assert int(normal.sum()) > 1000 and torch.equal(a[normal], b[normal]), "dL/dgrid (normal range)"
    assert float((a - b).abs().max()) <= 2.0 ** -24, "dL/dgrid: subnormal results differ by at most one subnormal step (double rounding of the scaled protocol)"


@pytest.mark.parametrize("case,dims", [("ngp_small", (32, 16)), ("ngp_smooth", (64, 64, 1)), ("pair_f4", (32, 32, 16)),
                                       ("hash_npow2_f2", (48, 5)), ("ngp_pair", (32, 16))])
@pytest.mark.parametrize("ptype", ["float", "half"])
def test_encode_and_decode_in_one_kernel(oracle, dev, hip_option, case, dims, ptype):
    """round 6: nr3d_lotd_mlp_forward (encode -> decoder forward, only the asked columns leave) against the two calls it replaces.
    With the two-lane forward serving every level (fwd_lds_stage = 0) the features are the same bits and so is the decoder's output,
    on both decoder routes (f32 MFMA / bf16 x3); against the oracle's features + a float64 decoder to 1e-5 of the column scale."""
    from nr3d_lib_amd.bindings import _lotd, _mlp
    # (a hash table whose size is not a power of two, 2-feature levels: the % path of the gather)
    D, res, nf, types, T, smooth = (3, [9, 17, 33], [2, 2, 2], ["Dense", "Hash", "Hash"], 3001, False) if case == "hash_npow2_f2" else LOTD_CASES[case]
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    E = m.n_encoded_dims
    n = 70001                                               # not a multiple of 64: a ragged last round
    x, p, _, _ = lotd_inputs(m_ref.as_dict(), n, 5)
    rng = np.random.default_rng(9)
    widths = [E, *dims]
    ws = [torch.from_numpy((rng.standard_normal((b, a)) / np.sqrt(a)).astype(np.float32)).to(dev) for a, b in zip(widths[:-1], widths[1:])]
    bs = [torch.from_numpy((rng.standard_normal(b) * 0.1).astype(np.float32)).to(dev) for b in widths[1:]]
    desc = _mlp.MLPDesc(widths, _mlp.ACT_RELU, _mlp.ACT_NONE)
    assert _lotd.lod_mlp_fwd_ok(m, desc)
    packed = _mlp.pack(desc, ws, bs)
    xt = torch.from_numpy(x).to(dev)
    pt = torch.from_numpy(p).to(dev)
    pt = pt.half() if ptype == "half" else pt
    hip_option("fwd_lds_stage", 0)
    for x3 in (1, 0):
        hip_option("mlp_x3", x3)
        y, _ = _lotd.lod_fwd(m, xt, pt)
        want = _mlp.forward(desc, y.float(), packed)
        for cols in (None, 1):
            got = _lotd.lod_mlp_fwd(m, xt, pt, desc, packed, out_cols=cols)
            assert got.shape == (n, widths[-1] if cols is None else 1)
            assert_equal(got, want[:, :got.shape[1]], f"x3={x3} cols={cols}: fused vs two calls")
    # and against the oracle chain in float64
    feat = oracle.lotd_fwd(m_ref, x, pt.float().cpu().numpy())[0].astype(np.float64)
    h = feat
    for l, (w, b) in enumerate(zip(ws, bs)):
        h = h @ w.double().cpu().numpy().T + b.double().cpu().numpy()
        if l + 1 < len(ws):
            h = np.maximum(h, 0)
    got = _lotd.lod_mlp_fwd(m, xt, pt, desc, packed)
    assert_close(got, h, rel=2e-5 if ptype == "float" else 2e-3, name="fused encode + decode vs oracle + float64 decoder")


def test_encode_and_decode_range_and_module(dev):
    """outside the kernel's range the check says no and the module takes the two ops; inside, LoTD.forward_decoded == decoder(encoding)"""
    from nr3d_lib_amd.bindings import _lotd, _mlp
    from nr3d_lib_amd.models.blocks import MLP
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTD
    from nr3d_lib_amd.models.grid_encodings.lotd import lotd as lotd_mod
    D, res, nf, types, T, smooth = LOTD_CASES["mixed"]
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    assert not _lotd.lod_mlp_fwd_ok(m, _mlp.MLPDesc([m.n_encoded_dims, 32, 4]))          # VM / CP levels
    D, res, nf, types, T, smooth = LOTD_CASES["ngp_small"]
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    assert not _lotd.lod_mlp_fwd_ok(m, _mlp.MLPDesc([m.n_encoded_dims, 128, 4]))         # hidden width 128
    assert not _lotd.lod_mlp_fwd_ok(m, _mlp.MLPDesc([m.n_encoded_dims + 1, 32, 4]))      # not the encoder's width
    enc = LoTD(3, res, nf, types, hashmap_size=T, dtype=torch.float)
    torch.manual_seed(3)
    grid = (torch.randn(enc.n_params) * 0.1).to(dev)
    dec = MLP(enc.out_features, 16, D=1, W=32, dtype=torch.float, device=dev)
    x = torch.rand(3, 1111, 3, device=dev)
    with torch.no_grad():
        want = dec(enc(x, grid))
    assert lotd_mod.FUSE_DECODED is False              # measured slower on the full loop (round 6): the two ops are the default
    for fuse in (True, False):
        lotd_mod.FUSE_DECODED = fuse
        try:
            got = enc.forward_decoded(x, grid, dec, out_cols=2)
        finally:
            lotd_mod.FUSE_DECODED = False
        assert got.shape == (3, 1111, 2) and not got.requires_grad
        assert_close(got.reshape(-1, 2), want[..., :2].reshape(-1, 2).cpu().numpy(), rel=1e-5, name=f"forward_decoded fuse={fuse}")
