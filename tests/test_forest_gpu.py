"""GPU parity: forest LoTD kernels, the octree query and the forest marcher (through the C ABI / bindings) vs the CPU
oracle.  fp32 values within REL_TOL of max|ref|; indices, counts and the marcher's
t-sequence bit-exact."""
import numpy as np
import pytest
import torch

from test_forest_cpu import FORESTS, forest_scene
from util import assert_close, assert_equal

pytestmark = pytest.mark.gpu

FOREST_METAS = {
    # name: (lod_res, n_feats, types, hashmap_size, smoothstep)
    "dense_hash": ([4, 6, 9, 13], [2, 2, 2, 2], ["Dense", "Dense", "Hash", "Hash"], 2 ** 9, False),
    "mixed": ([4, 6, 5, 7, 6], [2, 4, 2, 2, 4], ["Dense", "VM", "NPlaneMul", "CP", "Hash"], 257, False),
    "mixed_smooth": ([5, 4, 6], [4, 4, 8], ["VM", "Dense", "Hash"], 509, True),
    "cuboid_f8": ([[4, 6, 5], [7, 5, 9]], [8, 8], ["Dense", "CP"], None, False),
}


# VM levels with two-feature pseudo levels (what lotd_sorted.hip serves): cuboid resolutions, several pseudo levels per level
SORTED_METAS = {
    "vm_cuboid": ([[5, 7, 6], [9, 8, 11], [4, 4, 4]], [2, 4, 2], ["VM", "VM", "Dense"], None, False),
    "vm_smooth": ([6, 10], [4, 2], ["VM", "VM"], None, True),
}


def _forest_meta(dev, fo):
    from nr3d_lib_amd.bindings._forest import ForestMeta
    m = ForestMeta()
    m.octree, m.exsum = torch.from_numpy(fo.octree).to(dev), torch.from_numpy(fo.exsum).to(dev)
    m.block_ks = torch.from_numpy(fo.block_ks).to(dev)
    m.n_trees, m.level, m.level_poffset = fo.n_trees, fo.level, fo.level_poffset
    m.world_origin, m.world_block_size = fo.world_origin.tolist(), fo.world_block_size.tolist()
    m.resolution = [1 << fo.level] * 3
    m.continuity_enabled = fo.continuity_enabled
    return m


def _inputs(m_ref, fo, n, seed):
    """points kept away from cell boundaries of every level (x*R + 0.5 not within 1e-3 of an integer)"""
    rng = np.random.default_rng(seed)
    x = rng.random((n, 3)).astype(np.float32).clip(1e-6, 1 - 1e-6)
    for _ in range(8):
        bad = np.zeros(n, bool)
        for res in m_ref.as_dict()["level_res_multidim"]:
            v = x.astype(np.float64) * np.array(res) + 0.5
            bad |= (np.abs(v - np.round(v)) < 1e-3).any(1)
        if not bad.any():
            break
        x[bad] = rng.random((int(bad.sum()), 3)).astype(np.float32).clip(1e-6, 1 - 1e-6)
    # a good share of the points right at the block faces, where the neighbour lookup happens
    edge = rng.random(n) < 0.3
    x[edge, rng.integers(0, 3, n)[edge]] = np.where(rng.random(int(edge.sum())) < 0.5, 0.004, 0.996).astype(np.float32)
    params = (rng.standard_normal(fo.n_trees * m_ref.n_params) * 0.1).astype(np.float32)
    g = (rng.standard_normal((n, m_ref.n_encoded_dims)) * 0.1).astype(np.float32)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    bi = rng.integers(0, fo.n_trees, n).astype(np.int64)
    return x, params, g, v, bi


def _setup(oracle, dev, forest, case, n=2500, seed=0, continuity=True):
    from nr3d_lib_amd.bindings import _lotd
    level, blocks = FORESTS[forest]
    fo = oracle.forest_from_blocks(blocks, level, continuity_enabled=continuity)
    res, nf, types, T, smooth = {**FOREST_METAS, **SORTED_METAS}[case]
    m_ref = oracle.lotd_create_meta(3, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(3, res, nf, types, T, smooth)
    arrs = _inputs(m_ref, fo, n, seed)
    return _lotd, fo, m_ref, (m, _forest_meta(dev, fo)), arrs, tuple(torch.from_numpy(a).to(dev) for a in arrs)


@pytest.mark.parametrize("forest", ["plus", "scatter", "single"])
def test_identify(oracle, dev, forest):
    from nr3d_lib_amd.bindings._forest import forest_identify
    level, blocks = FORESTS[forest]
    fo = oracle.forest_from_blocks(blocks, level)
    side = 1 << level
    ks = np.array([(x, y, z) for x in range(-1, side + 1) for y in range(-1, side + 1) for z in range(-1, side + 1)], np.int16)
    want = oracle.forest_identify(fo, ks)
    want = np.where(want < 0, -1, want - fo.level_poffset)
    assert_equal(forest_identify(_forest_meta(dev, fo), torch.from_numpy(ks).to(dev)), want.astype(np.int32))


@pytest.mark.parametrize("case", list(FOREST_METAS))
@pytest.mark.parametrize("binned", [True, False])
def test_half_tables_read_natively(oracle, dev, case, binned, monkeypatch):
    """half tables of a forest: the forest kernels (forward, both dL/dparam forms incl. the binned path's product-type
    factors, d(dL/dx)/dx) read the half entries themselves (ABI 3: param_dtype on the forest entry points) -- the same bits as
    the run on an fp32 copy of the tables (hardware-atomic dL/dparam: to tolerance, arrival-order sums)"""
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, "plus", case, continuity=True)
    monkeypatch.setattr(_lotd, "USE_BINNED_DPARAM", binned)
    ph, gh = pt.half(), gt.half()
    outs = []
    for native in (True, False):
        monkeypatch.setattr(_lotd, "NATIVE_HALF", native)
        y, j = _lotd.lod_fwd(metas, xt, ph, bit, need_input_grad=True)
        dx, dp = _lotd.lod_bwd(metas, gh, xt, ph, j, bit, need_input_grad=True, need_param_grad=True)
        ddy, dp2, dx2 = _lotd.lod_bwd_bwd_input(metas, vt, gh, xt, ph, j, bit, need_dLdinput_ddLdoutput=True,
                                                need_dLdinput_dparams=True, need_dLdinput_dinput=True)
        outs.append((y, j, dx, ddy, dx2, dp, dp2))
    for k, (a, b) in enumerate(zip(*outs)):
        assert a.dtype == b.dtype
        if k >= 5 and not binned:
            assert_close(a.float(), b.float().cpu().numpy(), rel=1e-3, name=f"output {k} (atomics)", levels=m_ref)
        else:
            assert torch.equal(a, b), f"output {k}: half tables vs their fp32 copy"
    assert outs[0][0].dtype == torch.float16
    y_ref, _ = oracle.lotd_forest_fwd(m_ref, fo, x, ph.float().cpu().numpy(), block_inds=bi, need_dydx=True)
    assert np.allclose(outs[0][0].float().cpu().numpy(), y_ref, rtol=1e-3, atol=6e-8)


@pytest.mark.parametrize("case", list(FOREST_METAS))
@pytest.mark.parametrize("forest,continuity", [("plus", True), ("plus", False), ("scatter", True), ("single", True)])
def test_fwd_bwd_and_second_order(oracle, dev, forest, continuity, case):
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, forest, case, continuity=continuity)
    y_ref, j_ref = oracle.lotd_forest_fwd(m_ref, fo, x, p, block_inds=bi, need_dydx=True)
    y, j = _lotd.lod_fwd(metas, xt, pt, bit, need_input_grad=True)
    assert_close(y, y_ref, name="y")
    assert_close(j.reshape(j_ref.shape), j_ref, name="dy_dx")
    y2, j2 = _lotd.lod_fwd(metas, xt, pt, bit, need_input_grad=False)
    assert j2 is None
    assert_close(y2, y_ref, name="y(no grad)")
    dx, dp = _lotd.lod_bwd(metas, gt, xt, pt, j, bit, need_input_grad=True, need_param_grad=True)
    assert_close(dx, oracle.lotd_bwd_dx(m_ref, g, j_ref), name="dL_dx")
    assert_close(dp, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, accum_double=True), name="dL_dparam", levels=m_ref)
    ddy, dp2, dx2 = _lotd.lod_bwd_bwd_input(metas, vt, gt, xt, pt, j, bit, need_dLdinput_ddLdoutput=True,
                                            need_dLdinput_dparams=True, need_dLdinput_dinput=True)
    assert_close(ddy, oracle.lotd_bwd_bwd_ddLdy(m_ref, v, j_ref), name="dL_ddLdy")
    assert_close(dp2, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, dL_ddLdx=v, accum_double=True),
                 name="d(dLdx)/dparam", levels=m_ref)
    assert_close(dx2, oracle.lotd_forest_bwd_bwd_dx(m_ref, fo, v, g, x, p, block_inds=bi), name="d(dLdx)/dx")


def test_block_modes_skips_max_level_and_errors(oracle, dev):
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, "scatter", "dense_hash", n=9 * 200, seed=5)
    T, n = fo.n_trees, 200
    # batched: points grouped per block, in block order
    y_ref, j_ref = oracle.lotd_forest_fwd(m_ref, fo, x, p, batch_data_size=n, need_dydx=True)
    y, j = _lotd.lod_fwd(metas, xt, pt, None, None, n, None, True)
    assert_close(y, y_ref, name="batched y")
    dp = _lotd.lod_bwd(metas, gt, xt, pt, j, None, None, n, None, False, True)[1]
    assert_close(dp, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, batch_data_size=n, accum_double=True), name="batched dparam", levels=m_ref)
    # block_offsets: tables stored in another order
    rng = np.random.default_rng(0)
    perm = rng.permutation(T)
    offs = np.empty(T, np.int64); offs[perm] = np.arange(T) * m_ref.n_params
    shuffled = np.concatenate([p[b * m_ref.n_params:(b + 1) * m_ref.n_params] for b in perm])
    st, ot = torch.from_numpy(shuffled).to(dev), torch.from_numpy(offs).to(dev)
    y_ref2, _ = oracle.lotd_forest_fwd(m_ref, fo, x, p, block_inds=bi)
    assert_close(_lotd.lod_fwd(metas, xt, st, bit, ot)[0], y_ref2, name="block_offsets y")
    dp_o = _lotd.lod_bwd(metas, gt, xt, st, None, bit, ot, None, None, False, True)[1]
    assert_close(dp_o, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, shuffled, block_inds=bi, block_offsets=offs, accum_double=True),
                 name="block_offsets dparam", levels=m_ref)
    # skipped points and max_level
    bi2 = bi.copy(); bi2[::4] = -1
    b2t = torch.from_numpy(bi2).to(dev)
    ys, js = _lotd.lod_fwd(metas, xt, pt, b2t, None, None, 1, True)
    yr, jr = oracle.lotd_forest_fwd(m_ref, fo, x, p, block_inds=bi2, max_level=1, need_dydx=True)
    assert_close(ys, yr, name="skips y"); assert_close(js.reshape(jr.shape), jr, name="skips dy_dx")
    assert float(ys[::4].abs().max()) == 0 and float(ys[:, 4:].abs().max()) == 0
    dps = _lotd.lod_bwd(metas, gt, xt, pt, None, b2t, None, None, 1, False, True)[1]
    assert_close(dps, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi2, max_level=1, accum_double=True), name="skips dparam", levels=m_ref)
    y0, j0 = _lotd.lod_fwd(metas, xt, pt, bit, None, None, -1, True)
    assert float(y0.abs().max()) == 0 and float(j0.abs().max()) == 0
    # errors: unsupported level type, 2-D meta, wrong block_offsets size, missing octree, forest grid index
    bad = _lotd.LoDMeta(3, [5], [2], ["CPfast"], None, False)
    with pytest.raises(RuntimeError, match="forest levels are"):
        _lotd.lod_fwd((bad, metas[1]), xt, torch.zeros(T * bad.n_params, device=dev), bit)
    m2 = _lotd.LoDMeta(2, [5], [2], ["Dense"], None, False)
    with pytest.raises(RuntimeError, match="n_dims_to_encode"):
        _lotd.lod_fwd((m2, metas[1]), xt[:, :2].contiguous(), torch.zeros(T * m2.n_params, device=dev), bit)
    with pytest.raises(RuntimeError, match="batch_offset"):
        _lotd.lod_fwd(metas, xt, pt, bit, ot[:-1].contiguous())
    with pytest.raises(RuntimeError, match="Not implemented for forest"):
        _lotd.lod_get_grid_index(metas, xt, bit)
    from nr3d_lib_amd.bindings._forest import ForestMeta
    with pytest.raises(RuntimeError, match="octree is not set"):
        _lotd.lod_fwd((metas[0], ForestMeta()), xt, pt, bit)


def test_forest_encoding_module_and_space(oracle, dev):
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDForestEncoding
    level, blocks = FORESTS["plus"]
    cfg = dict(lod_res=[4, 6, 9], lod_n_feats=[2, 2, 2], lod_types=["Dense", "VM", "Hash"], hashmap_size=257)
    enc = LoTDForestEncoding(3, lotd_cfg=cfg, dtype=torch.float, device=dev,
                             param_init_cfg=dict(type="uniform", bound=0.5))
    enc.populate(mode="from_corners", corners=blocks, level=level, world_origin=[-2., -2, -2], world_block_size=1.0)
    sp = enc.space
    fo = oracle.forest_from_blocks(blocks, level)
    assert_equal(sp.meta.octree, fo.octree); assert_equal(sp.meta.exsum, fo.exsum); assert_equal(sp.block_ks, fo.block_ks)
    assert sp.meta.level_poffset == fo.level_poffset and tuple(enc.forest_flattened_params.shape) == (fo.n_trees, enc.lod_meta.n_params)
    # world <-> block coordinates
    rng = np.random.default_rng(2)
    bi = rng.integers(0, fo.n_trees, 500)
    xb = (rng.random((500, 3)) * 1.9 - 0.95).astype(np.float32)
    world = sp.unnormalize_coords(torch.from_numpy(xb).to(dev), torch.from_numpy(bi).to(dev))
    xb2, bi2 = sp.normalize_coords(world)
    assert_equal(bi2, bi); assert_close(xb2, xb, rel=1e-5, name="normalize(unnormalize(x))")
    outside = torch.tensor([[5.0, 5.0, 5.0], [-2.5, -1.5, -1.5]], device=dev)
    assert sp.normalize_coords(outside)[1].tolist() == [-1, -1]
    # forward / backward / nablas against the oracle
    x = torch.from_numpy(xb).to(dev).requires_grad_(True)
    bit = torch.from_numpy(bi).to(dev)
    y = enc(x, bit)
    m_ref = oracle.lotd_create_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], 257)
    p = enc.forest_flattened_params.detach().cpu().numpy().ravel()
    x01 = np.clip(xb / np.float32(2.) + np.float32(0.5), 1e-6, 1 - 1e-6).astype(np.float32)
    y_ref, j_ref = oracle.lotd_forest_fwd(m_ref, fo, x01, p, block_inds=bi, need_dydx=True)
    assert_close(y, y_ref, name="module y")
    g = torch.from_numpy((rng.standard_normal(y_ref.shape) * 0.1).astype(np.float32)).to(dev)
    (y * g).sum().backward()
    gn = g.cpu().numpy()
    assert_close(enc.forest_flattened_params.grad.view(-1), oracle.lotd_forest_bwd_dparam(m_ref, fo, gn, x01, p, block_inds=bi, accum_double=True),
                 name="module dparam")
    assert_close(x.grad, oracle.lotd_bwd_dx(m_ref, gn, j_ref) / 2, name="module dx")
    # second order through forward_dydx / backward_dydx
    enc.zero_grad()
    y2, dydx = enc.forward_dydx(x.detach(), bit)
    gg = g.clone().requires_grad_(True)
    nablas = enc.backward_dydx(gg, dydx, x.detach(), bit)
    v = torch.from_numpy(rng.standard_normal((500, 3)).astype(np.float32)).to(dev)
    (nablas * v).sum().backward()
    vn = v.cpu().numpy() / 2
    assert_close(gg.grad, oracle.lotd_bwd_bwd_ddLdy(m_ref, vn, j_ref), name="module d nablas / d dL_dy")
    assert_close(enc.forest_flattened_params.grad.view(-1),
                 oracle.lotd_forest_bwd_dparam(m_ref, fo, gn, x01, p, block_inds=bi, dL_ddLdx=vn, accum_double=True), name="module d nablas / dparam")
    # batched input [n_trees, k, 3]; level views; state dict round trip
    xbat, _ = sp.sample_pts_uniform(num_pts_per_block=16)
    yb = enc(xbat)
    assert tuple(yb.shape) == (fo.n_trees, 16, enc.out_features)
    yb_ref, _ = oracle.lotd_forest_fwd(m_ref, fo, np.clip(xbat.cpu().numpy().reshape(-1, 3) / 2 + 0.5, 1e-6, 1 - 1e-6), p, batch_data_size=16)
    assert_close(yb.reshape(-1, enc.out_features), yb_ref, name="module batched y")
    assert tuple(enc.get_level_param(slice(None), 0, 'vol').shape) == (fo.n_trees, 4, 4, 4, 2)
    enc2 = LoTDForestEncoding(3, lotd_cfg=cfg, dtype=torch.float, device=dev)
    enc2.load_state_dict(enc.state_dict())
    assert_equal(enc2.space.block_ks, fo.block_ks)
    assert_close(enc2(x.detach(), bit), y_ref, name="reloaded module y")


@pytest.mark.parametrize("dt_gamma,max_steps", [(0.0, 64), (0.01, 64), (0.0, 5)])
def test_forest_marching_bit_exact(oracle, dev, dt_gamma, max_steps):
    from nr3d_lib_amd.bindings import _occ_grid
    fo, grid, o, d, near, far, (sb, se, sx, sp) = forest_scene(oracle, seed=1, n_rays=700)
    step = 0.04
    ref = oracle.forest_ray_marching(fo, o, d, near, far, sb, se, sx, sp, grid, step, 0.2, dt_gamma, max_steps, True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fm = _forest_meta(dev, fo)
    got = _occ_grid.forest_ray_marching(fm, t(o), t(d), t(near), t(far), t(sb), t(se), t(sx), t(sp), t(grid), step, 0.2,
                                        dt_gamma, max_steps, True)
    assert ref[1].shape[0] > 0
    for a, b, name in zip(got, ref, ("packed_info", "t_starts", "t_ends", "ridx", "blidx", "gidx")):
        assert_equal(a, b, name)
    got2 = _occ_grid.forest_ray_marching(fm, t(o), t(d), t(near), t(far), t(sb), t(se), t(sx), t(sp), t(grid), step, 0.2,
                                         dt_gamma, max_steps, False)
    assert got2[5] is None
    assert_equal(got2[1], ref[1], "t_starts (no gidx)")
    # no segments at all
    z = _occ_grid.forest_ray_marching(fm, t(o), t(d), t(near), t(far), t(sb[:0]), t(se[:0]), t(sx[:0]), t(np.zeros_like(sp)),
                                      t(grid), step, 0.2, dt_gamma, max_steps, True)
    assert z[1].shape == (0, 1) and int(z[0][:, 1].sum()) == 0


def test_space_ray_test_feeds_the_marcher(oracle, dev):
    """ForestBlockSpace.ray_test -> forest_ray_marching: the segments of the host-side slab test are the ones the
    numpy restatement finds, and marching them gives the oracle's samples"""
    from nr3d_lib_amd.bindings import _occ_grid
    from nr3d_lib_amd.models.spatial import ForestBlockSpace
    fo, grid, o, d, near, far, (sb, se, sx, sp) = forest_scene(oracle, seed=2, n_rays=400)
    space = ForestBlockSpace(device=dev)
    space.populate(mode="from_corners", corners=FORESTS["plus"][1], level=FORESTS["plus"][0], world_origin=[-2., -2, -2],
                   world_block_size=1.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rt = space.ray_test(t(o), t(d), near=t(near), far=t(far))
    hit = np.nonzero(sp[:, 1])[0]
    assert_equal(rt["rays_inds"], hit)
    assert_equal(rt["seg_pack_infos"][:, 1], sp[hit, 1])
    assert_equal(rt["seg_block_inds"], sb)
    assert_close(rt["seg_entries"], se, rel=1e-6, name="entries"); assert_close(rt["seg_exits"], sx, rel=1e-6, name="exits")
    got = _occ_grid.forest_ray_marching(space.meta, rt["rays_o"], rt["rays_d"], rt["near"], rt["far"], rt["seg_block_inds"].int(),
                                        rt["seg_entries"], rt["seg_exits"], rt["seg_pack_infos"].int(), t(grid), 0.04, 1e10, 0.0, 64, True)
    ref = oracle.forest_ray_marching(fo, o[hit], d[hit], near[hit], far[hit], rt["seg_block_inds"].cpu().numpy(),
                                     rt["seg_entries"].cpu().numpy(), rt["seg_exits"].cpu().numpy(),
                                     rt["seg_pack_infos"].cpu().numpy(), grid, 0.04, 1e10, 0.0, 64, True)
    for a, b, name in zip(got, ref, ("packed_info", "t_starts", "t_ends", "ridx", "blidx", "gidx")):
        assert_equal(a, b, name)


@pytest.mark.parametrize("case", ["dense_hash", "mixed", "mixed_smooth", "cuboid_f8"])
@pytest.mark.parametrize("forest", ["plus", "scatter"])
def test_dparam_binned_and_atomic_paths_agree(oracle, dev, forest, case):
    """forests take the sort + segmented-sum path (blocks = batch entries, every corner's updates binned into the tables of
    the block that owns it; Dense / Hash: 8 records per point and level, CP / NPlaneMul 24, VM 48); forcing the atomic
    scatter must give the same gradients, first and second order"""
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, forest, case, n=20000, seed=9)
    ref1 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, accum_double=True)
    ref2 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, dL_ddLdx=v, accum_double=True)
    from nr3d_lib_amd.bindings import _forest
    assert _forest._workspace(metas[0], metas[1], 20000, dev)[1] > 0            # the binned path applies
    _, j = _lotd.lod_fwd(metas, xt, pt, bit, need_input_grad=True)
    try:
        for binned in (True, False):
            _lotd.USE_BINNED_DPARAM = binned
            dp = _lotd.lod_bwd(metas, gt, xt, pt, None, bit, need_input_grad=False, need_param_grad=True)[1]
            tol = 1e-5 if binned else 5e-5          # fp32 atomics: arbitrary order, run-dependent rounding
            assert_close(dp, ref1, rel=tol, name=f"dL_dparam binned={binned}", levels=m_ref)
            dp2 = _lotd.lod_bwd_bwd_input(metas, vt, gt, xt, pt, j, bit, need_dLdinput_ddLdoutput=False,
                                          need_dLdinput_dparams=True, need_dLdinput_dinput=False)[1]
            assert_close(dp2, ref2, rel=tol, name=f"d(dLdx)/dparam binned={binned}", levels=m_ref)
    finally:
        _lotd.USE_BINNED_DPARAM = True
    # ray-like coherent points (the run-merging of stage A) inside one block, then crossing into its neighbour
    t = np.linspace(0.02, 0.98, 4096, dtype=np.float32)
    xs = np.stack([t, np.full_like(t, 0.37), np.full_like(t, 0.993)], 1)
    bs = np.full(4096, 1, np.int64)
    gs = g[:4096]
    dpc = _lotd.lod_bwd(metas, torch.from_numpy(gs).to(dev), torch.from_numpy(xs).to(dev), pt, None, torch.from_numpy(bs).to(dev),
                        need_input_grad=False, need_param_grad=True)[1]
    assert_close(dpc, oracle.lotd_forest_bwd_dparam(m_ref, fo, gs, xs, p, block_inds=bs, accum_double=True), name="coherent dparam", levels=m_ref)


@pytest.mark.parametrize("forest,continuity", [("plus", True), ("plus", False), ("scatter", True), ("single", True)])
@pytest.mark.parametrize("case", ["mixed", "vm_cuboid", "vm_smooth"])
def test_vm_levels_over_sorted_points(oracle, dev, forest, continuity, case, hip_option, monkeypatch):
    """VM levels of a forest over SORTED points (lotd_sorted.hip, option vm_sorted = 2: whatever the table size): interior cells
    from the band's own point range, boundary cells corner by corner from the face-distance list, no records -- against the
    oracle, against the record path, first and second order, twice (the sums have a fixed order)"""
    from nr3d_lib_amd import _hip
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, forest, case, n=20000, seed=31, continuity=continuity)
    ref1 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, accum_double=True)
    ref2 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, dL_ddLdx=v, accum_double=True)
    _, j = _lotd.lod_fwd(metas, xt, pt, bit, need_input_grad=True)
    outs = {}
    for mode in (2, 0, 2):
        hip_option("vm_sorted", mode)
        _hip.prof_enable("lotd_direct")
        try:
            dp = _lotd.lod_bwd(metas, gt, xt, pt, None, bit, need_input_grad=False, need_param_grad=True)[1]
            dp2 = _lotd.lod_bwd_bwd_input(metas, vt, gt, xt, pt, j, bit, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                          need_dLdinput_dinput=False)[1]
            ran = _hip.prof_read("lotd_direct")[1]
        finally:
            _hip.prof_enable()
        assert (ran > 0) == (mode == 2), f"vm_sorted={mode}: the sorted kernel ran {ran} times"
        assert_close(dp, ref1, rel=1e-5, name=f"dL_dparam vm_sorted={mode}", levels=m_ref)
        assert_close(dp2, ref2, rel=1e-5, name=f"d(dLdx)/dparam vm_sorted={mode}", levels=m_ref)
        if mode in outs:
            assert torch.equal(dp, outs[mode][0]) and torch.equal(dp2, outs[mode][1]), "two runs over sorted points differ"
        outs[mode] = (dp, dp2)
    # half tables read natively: the same bits as the run on their fp32 copy
    ph, gh = pt.half(), gt.half()
    hip_option("vm_sorted", 2)
    halves = []
    for native in (True, False):
        monkeypatch.setattr(_lotd, "NATIVE_HALF", native)
        halves.append(_lotd.lod_bwd(metas, gh, xt, ph, None, bit, need_input_grad=False, need_param_grad=True)[1])
    assert halves[0].dtype == halves[1].dtype and torch.equal(halves[0], halves[1])


@pytest.mark.parametrize("case", ["mixed", "vm_cuboid"])
def test_vm_levels_over_sorted_points_all_in_one_block(oracle, dev, case, hip_option):
    """every point in ONE block of the forest: the other blocks' bands have no points of their own (the empty-band shortcut of
    k_vm_sorted) but still receive the corners they own of that block's boundary cells"""
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, "plus", case, n=12000, seed=47)
    for blk in (0, fo.n_trees - 1):
        bi1 = np.full_like(bi, blk)
        bit1 = torch.from_numpy(bi1).to(dev)
        ref1 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi1, accum_double=True)
        ref2 = oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi1, dL_ddLdx=v, accum_double=True)
        hip_option("vm_sorted", 2)
        dp = _lotd.lod_bwd(metas, gt, xt, pt, None, bit1, need_input_grad=False, need_param_grad=True)[1]
        dp2 = _lotd.lod_bwd_bwd_input(metas, vt, gt, xt, pt, None, bit1, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                      need_dLdinput_dinput=False)[1]
        assert_close(dp, ref1, rel=1e-5, name=f"dL_dparam, all points in block {blk}", levels=m_ref)
        assert_close(dp2, ref2, rel=1e-5, name=f"d(dLdx)/dparam, all points in block {blk}", levels=m_ref)
        per_block = dp.view(fo.n_trees, -1).abs().amax(1)
        assert int((per_block > 0).sum()) > 1, "a block with neighbours must hand them the corners they own"


def test_forest_accel_end_to_end(oracle, dev):
    """ForestBlockSpace + OccGridAccelForest: per-block grids learnt from a world-space field (init, warm-up and steady
    steps, renderer samples), then world rays marched through the blocks they cross -- against the oracle's march on
    the same grids, and against the field itself"""
    from nr3d_lib_amd.models.accelerations.occgrid import OccGridEmaBatched
    from nr3d_lib_amd.models.accelerations.occgrid_accel import OccGridAccelForest
    from nr3d_lib_amd.models.spatial import ForestBlockSpace
    torch.manual_seed(0)
    level, blocks = FORESTS["plus"]
    space = ForestBlockSpace(device=dev)
    space.populate(mode="from_corners", corners=blocks, level=level, world_origin=[-2., -2, -2], world_block_size=1.0)
    accel = OccGridAccelForest(space, vox_size=1 / 16, occ_thre=0.3, ema_decay=0.9, n_steps_between_update=2, n_steps_warmup=4,
                               init_cfg=dict(mode="from_net", num_steps=2, num_pts_per_batch=2 ** 14),
                               update_from_net_cfg=dict(num_steps=1, num_pts_per_batch=2 ** 13))
    accel.populate()
    assert tuple(accel.get_occ_grid().shape) == (space.n_trees, 16, 16, 16) and isinstance(accel.occ, OccGridEmaBatched)
    centre = torch.tensor([-0.5, -0.5, -0.5], device=dev)            # centre of block (1,1,1), the hub of the plus

    def field_world(p):                                              # a spherical shell of radius 0.8 around the hub
        return torch.exp(-(((p - centre).norm(dim=-1) - 0.8) / 0.1) ** 2)

    def query(block_x, bidx):                                        # block coordinates in, field value out
        return field_world(space.unnormalize_coords(block_x, bidx))
    accel.train()
    assert accel.init(query)
    for it in range(1, 7):
        assert accel.step(it, query) == (it % 2 == 0)
    stats = accel.debug_stats()
    assert 0.02 < stats["frac_occupied"] < 0.6
    # the learnt grids agree with the field at voxel centres (well inside / well outside the shell)
    ctr = (torch.stack(torch.meshgrid(*[torch.arange(16, device=dev)] * 3, indexing="ij"), -1) + 0.5) / 8 - 1    # block coords
    for b in range(space.n_trees):
        f = field_world(space.unnormalize_coords(ctr.reshape(-1, 3), torch.full((4096,), b, device=dev))).view(16, 16, 16)
        g = accel.get_occ_grid()[b]
        assert float(g[f > 0.8].float().mean()) > 0.95 if bool((f > 0.8).any()) else True
        assert float(g[f < 0.01].float().mean()) < 0.02
    # world queries, renderer samples
    on_shell = centre + torch.tensor([[0.8, 0, 0], [0, 0.8, 0]], device=dev)       # in blocks (2,1,1) and (1,2,1)
    assert accel.query_world(on_shell).tolist() == [True, True]
    assert accel.query_world(torch.tensor([[-0.5, -0.5, -0.5], [5.0, 5, 5]], device=dev)).tolist() == [False, False]
    hub = torch.tensor([[-0.5, -0.5, -0.5]], device=dev)
    accel.collect_samples(hub, None, torch.tensor([7.0], device=dev), normalized=False)
    accel.step(8, query)
    assert accel.query_world(hub).tolist() == [True]
    p_occ, b_occ = accel.sample_pts_in_occupied(500)
    assert bool(accel.query_occupancy(p_occ, b_occ).all())
    # rays: segments from the space, march on the GPU == the oracle's march on the same grids
    n = 300
    o = (torch.rand(n, 3, device=dev) * 0.4 + torch.tensor([-3.2, -0.7, -0.7], device=dev))
    d = torch.nn.functional.normalize(centre + (torch.rand(n, 3, device=dev) - 0.5) * 1.5 - o, dim=1)
    rt = space.ray_test(o, d, near=0.05, far=8.0)
    assert rt["num_rays"] > 0
    ret = accel.ray_march(rt["rays_o"], rt["rays_d"], rt["near"], rt["far"], rt["seg_block_inds"], rt["seg_entries"],
                          rt["seg_exits"], rt["seg_pack_infos"], step_size=0.02, max_steps=128)
    fo = oracle.forest_from_blocks(blocks, level, world_origin=(-2.0, -2.0, -2.0), world_block_size=(1.0, 1.0, 1.0))
    c = lambda t: t.cpu().numpy()
    ref = oracle.forest_ray_marching(fo, c(rt["rays_o"]), c(rt["rays_d"]), c(rt["near"]), c(rt["far"]), c(rt["seg_block_inds"]),
                                     c(rt["seg_entries"]), c(rt["seg_exits"]), c(rt["seg_pack_infos"]), c(accel.get_occ_grid()),
                                     0.02, 1e10, 0.0, 128, False)
    hit = np.nonzero(ref[0][:, 1])[0]
    assert ret.num_hit_rays == len(hit) > 0
    assert_equal(ret.ridx_hit, hit, "ridx_hit")
    assert_equal(ret.pack_infos, ref[0][hit].astype(np.int64), "pack_infos")
    assert_equal(ret.depth_samples, ref[1][:, 0], "depth_samples")
    assert_equal(ret.deltas, (ref[2] - ref[1])[:, 0], "deltas")
    assert_equal(ret.blidx, ref[4].astype(np.int64), "blidx")
    assert int(ret.blidx_pack_infos[:, 1].sum()) == ret.samples.shape[0]
    # every sample sits where the field is alive (the shell), in world coordinates
    assert float(field_world(ret.samples + 0.5 * ret.deltas[:, None] * rt["rays_d"][ret.ridx]).median()) > 0.05
    coarse = accel.ray_march_simple_step_segment(rt["rays_o"], rt["rays_d"], rt["near"], rt["far"], rt["seg_block_inds"],
                                                 rt["seg_entries"], rt["seg_exits"], rt["seg_pack_infos"], step_mode="depth",
                                                 max_steps=64, min_step_size=0.05, dt_gamma=0.0)
    assert coarse["samples"].shape[0] == coarse["blidx"].shape[0] > 0


def test_forest_dparam_multi_pass_chunking(oracle, dev, hiplib):
    """the binned path in several point chunks (block_inds advance with the chunk; batched mode keeps its global point index)"""
    _lotd, fo, m_ref, metas, (x, p, g, v, bi), (xt, pt, gt, vt, bit) = _setup(oracle, dev, "scatter", "dense_hash", n=9 * 2000, seed=11)
    hiplib.nr3d_lotd_set_dparam_chunk_log2(12)                   # 4096-point chunks -> 5 passes
    try:
        dp = _lotd.lod_bwd(metas, gt, xt, pt, None, bit, need_input_grad=False, need_param_grad=True)[1]
        assert_close(dp, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, block_inds=bi, accum_double=True), name="chunked dparam", levels=m_ref)
        dpb = _lotd.lod_bwd(metas, gt, xt, pt, None, None, None, 2000, None, False, True)[1]
        assert_close(dpb, oracle.lotd_forest_bwd_dparam(m_ref, fo, g, x, p, batch_data_size=2000, accum_double=True), name="chunked batched dparam", levels=m_ref)
    finally:
        hiplib.nr3d_lotd_set_dparam_chunk_log2(0)
