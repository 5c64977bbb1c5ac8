"""GPU parity: occupancy-grid marcher vs the CPU oracle.  packed_info / ridx / gidx bit-exact; t_starts /
t_ends bit-exact too for AABB (same fp32 operation order, IEEE division, explicit FMA)."""
import numpy as np
import pytest
import torch

from util import assert_close, assert_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "16", "32", "64"], ids=lambda v: f"lanes_per_ray={v}")
def march_group(request, hip_option):
    """every test under each marcher: one lane per ray (k_march) and 16 / 64 lanes per ray (k_march_group, look-ahead along
    the t recurrence) -- the same bits are expected from all three"""
    hip_option("march_group", request.param)
    return request.param


def pinhole_rays(n_side, dist=4.0, seed=0, jitter=True):
    """rays from a camera at distance `dist` looking at the origin (as the reference's smoke test,
    occgrid_raymarch.py:281-295); near/far from the ray-AABB test against [-1,1]^3"""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(-0.45, 0.45, n_side), np.linspace(-0.45, 0.45, n_side), indexing="ij")
    d = np.stack([u.ravel(), v.ravel(), np.ones(u.size)], 1)
    if jitter:
        d[:, :2] += rng.normal(0, 1e-3, (u.size, 2))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    th = 0.7
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.4), -np.sin(0.4)], [0, np.sin(0.4), np.cos(0.4)]])
    d = (d @ R.T).astype(np.float32)
    o = np.tile((-dist * R[:, 2]).astype(np.float32), (u.size, 1))
    with np.errstate(divide="ignore"):
        t1, t2 = (-1 - o) / d, (1 - o) / d
    near = np.minimum(t1, t2).max(1)
    far = np.maximum(t1, t2).min(1)
    hit = far > np.maximum(near, 0)
    near = np.where(hit, np.maximum(near, 0), 0).astype(np.float32)
    far = np.where(hit, far, 0).astype(np.float32)     # missed rays: near == far == 0 -> no samples
    return o, d, near, far


def grids(res, seed):
    rng = np.random.default_rng(seed)
    rnd = rng.random(res) > 0.5
    ax = [np.linspace(-1, 1, r) for r in res]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    shell = (r > 0.55) & (r < 0.7)
    return {"random": rnd, "shell": shell, "empty": np.zeros(res, bool), "full": np.ones(res, bool)}


def run_both(oracle, dev, o, d, near, far, roi, grid, ctype, step, max_step, gamma, max_steps, **batch):
    from nr3d_lib_amd.bindings import _occ_grid
    ref = oracle.ray_marching(o, d, near, far, roi, grid, ctype, step, max_step, gamma, max_steps, True,
                              batch_inds=batch.get("batch_inds"), batch_data_size=batch.get("batch_data_size"))
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if grid.ndim == 4:
        got = _occ_grid.batched_ray_marching(t(o), t(d), t(near), t(far), t(batch.get("batch_inds")),
                                             batch.get("batch_data_size"), t(roi), t(grid),
                                             _occ_grid.ContractionType(ctype), step, max_step, gamma, max_steps, True)
    else:
        got = _occ_grid.ray_marching(t(o), t(d), t(near), t(far), t(roi), t(grid), _occ_grid.ContractionType(ctype),
                                     step, max_step, gamma, max_steps, True)
    return got, ref


ROI = np.array([-1, -1, -1, 1, 1, 1], np.float32)


@pytest.mark.parametrize("kind", ["random", "shell", "empty", "full"])
@pytest.mark.parametrize("gamma,max_step", [(0.0, 1e10), (0.01, 0.05)])
def test_aabb_bit_exact(oracle, dev, kind, gamma, max_step):
    res = (32, 24, 40)
    o, d, near, far = pinhole_rays(24, seed=1)
    grid = grids(res, 2)[kind]
    max_steps = 48 if kind == "full" else 128
    got, ref = run_both(oracle, dev, o, d, near, far, ROI, grid, 0, 2 * 3 ** 0.5 / 128, max_step, gamma, max_steps)
    names = ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]
    assert got[0].dtype == torch.int32 and got[3].dtype == torch.int32 and got[4].dtype == torch.int32
    assert got[1].shape[-1] == 1 and got[1].dim() == 2
    for g, r, n in zip(got, ref, names):
        assert_equal(g, r, name=f"{kind}/{n}")       # incl. the float t-values: same op order => same bits
    if kind == "empty":
        assert got[1].shape[0] == 0
    if kind == "full":
        assert int(got[0][:, 1].max()) == 48         # max_steps cap reached on the long rays


def test_c3_config_bit_exact(oracle, dev):
    """BASELINE config 3 shape: 128^3 grid, 4096 rays, <= 512 samples per ray"""
    res = (128, 128, 128)
    o, d, near, far = pinhole_rays(64, seed=7)
    rng = np.random.default_rng(7)
    grid = rng.random(res) > 0.5
    got, ref = run_both(oracle, dev, o, d, near, far, ROI, grid, 0, 2 * 3 ** 0.5 / 512, 1e10, 0.0, 512)
    for g, r, n in zip(got, ref, ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]):
        assert_equal(g, r, name=n)
    pi = got[0].cpu().numpy()
    assert pi[:, 1].max() <= 512 and (pi[1:, 0] == np.cumsum(pi[:-1, 1])).all()


@pytest.mark.parametrize("ctype", [1, 2])
def test_contractions(oracle, dev, ctype):
    """tanh / sphere contraction: transcendental (tanhf) / sqrt results may differ in the last bit between
    libm and the device, which can move a sample across a voxel face; require >= 99.9% identical rays"""
    res = (32, 32, 32)
    o, d, near, far = pinhole_rays(20, seed=3)
    far = near + 6.0
    grid = grids(res, 4)["random"]
    got, ref = run_both(oracle, dev, o, d, near, far, ROI * 0.5, grid, ctype, 0.02, 1e10, 0.0, 200)
    same = (got[0].cpu().numpy() == ref[0]).all(1).mean()
    assert same >= 0.999, f"only {same:.4f} of rays have identical packed_info"
    if same == 1.0:
        assert (got[4].cpu().numpy() == ref[4]).mean() >= 0.999
        assert_close(got[1], ref[1], name="t_starts")


def test_non_default_roi_and_degenerate_rays(oracle, dev):
    res = (16, 20, 12)
    o, d, near, far = pinhole_rays(16, seed=5)
    roi = np.array([-0.8, -0.6, -0.9, 0.7, 0.9, 0.5], np.float32)
    d[3] = np.array([0, 0, 1], np.float32)             # axis-aligned: inv_dir has +-inf components
    d[4] = np.array([1, 0, 0], np.float32)
    far[5] = near[5]                                    # zero-length
    near[6], far[6] = 2.0, 1.0                          # inverted
    grid = grids(res, 6)["random"]
    got, ref = run_both(oracle, dev, o, d, near, far, roi, grid, 0, 0.01, 1e10, 0.0, 64)
    for g, r, n in zip(got, ref, ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]):
        assert_equal(g, r, name=n)


def test_batched_bit_exact(oracle, dev):
    B, res = 3, (16, 16, 16)
    o, d, near, far = pinhole_rays(18, seed=9)      # 324 rays
    rng = np.random.default_rng(9)
    grid = rng.random((B,) + res) > 0.6
    roi = np.stack([ROI, ROI * 0.9, ROI * 1.1]).astype(np.float32)
    bi = rng.integers(-1, B, o.shape[0]).astype(np.int32)
    for kw in (dict(batch_inds=bi), dict(batch_data_size=108)):
        got, ref = run_both(oracle, dev, o, d, near, far, roi, grid, 0, 0.01, 1e10, 0.0, 96, **kw)
        for g, r, n in zip(got, ref, ["packed_info", "t_starts", "t_ends", "ridx", "bidx", "gidx"]):
            assert_equal(g, r, name=f"{list(kw)}/{n}")


def test_wrapper_record(oracle, dev):
    """occgrid_raymarch: the 9-field record, derived tensors and the no-hit case"""
    from nr3d_lib_amd.graphics.raymarch.occgrid_raymarch import occgrid_raymarch, occgrid_raymarch_batched
    res = (32, 32, 32)
    o, d, near, far = pinhole_rays(16, seed=11)
    grid = grids(res, 12)["shell"]
    t = lambda a: torch.from_numpy(a).to(dev)
    ret = occgrid_raymarch(t(grid), t(o), t(d), t(near), t(far), step_size=0.01, max_steps=256)
    ref = oracle.ray_marching(o, d, near, far, ROI, grid, 0, 0.01, 1e10, 0.0, 256, True)
    hit = np.nonzero(ref[0][:, 1])[0]
    assert ret.num_hit_rays == len(hit)
    assert_equal(ret.ridx_hit, hit, "ridx_hit")
    assert_equal(ret.pack_infos, ref[0][hit].astype(np.int64), "pack_infos")
    assert ret.pack_infos.dtype == torch.int64 and ret.ridx.dtype == torch.int64 and ret.gidx.dtype == torch.int64
    assert_equal(ret.depth_samples, ref[1][:, 0], "depth_samples")
    assert_equal(ret.deltas, ref[2][:, 0] - ref[1][:, 0], "deltas")
    samples_ref = o[ref[3]] + d[ref[3]] * ref[1]
    assert_close(ret.samples, samples_ref, name="samples")
    assert len(list(ret)) == 9 and ret["gidx"] is ret.gidx
    ret_p = occgrid_raymarch(t(grid), t(o), t(d), t(near), t(far), step_size=0.01, max_steps=256, perturb=True)
    assert torch.equal(ret_p.depth_samples, ret.depth_samples)      # reference quirk: unperturbed t_starts
    none = occgrid_raymarch(t(np.zeros(res, bool)), t(o), t(d), 0.0, 5.0)
    assert none.num_hit_rays == 0 and none.samples is None
    retb = occgrid_raymarch_batched(t(np.stack([grid, grid])), t(np.stack([o, o])), t(np.stack([d, d])), None,
                                    t(np.stack([near, near])), t(np.stack([far, far])), step_size=0.01, max_steps=256)
    assert retb.num_hit_rays == 2 * len(hit) and int(retb.bidx.max()) == 1


@pytest.mark.parametrize("side", [16, 200])
def test_fused_march_finish_against_chain(oracle, dev, side):
    """occgrid_raymarch: hit-ray compaction + per-sample epilogue inside the library (2 launches, readback shared with the
    marcher's) against the reference's op chain (nonzero, index, .long(), sub, index_select x2, addcmul): every field
    bit-identical (positions to one rounding) -- side 200 = 40 000 rays takes the three-launch scan, side 16 the single-workgroup one"""
    import nr3d_lib_amd.graphics.raymarch.occgrid_raymarch as orm
    o, d, near, far = pinhole_rays(side, seed=5)
    grid = grids((32, 32, 32), 9)["shell"]
    t = lambda a: torch.from_numpy(a).to(dev)
    recs = {}
    for fused in (True, False):
        orm.FUSED_FINISH = fused
        try:
            recs[fused] = orm.occgrid_raymarch(t(grid), t(o), t(d), t(near), t(far), step_size=0.02, max_steps=128)
        finally:
            orm.FUSED_FINISH = True
    a, b = recs[True], recs[False]
    assert a.num_hit_rays == b.num_hit_rays > 0
    for name in ("ridx_hit", "samples", "depth_samples", "deltas", "ridx", "pack_infos", "gidx"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.dtype == y.dtype and x.shape == y.shape, name
        if name == "samples":      # o + d * t: the kernel rounds once (fma, what the reference's CUDA addcmul contracts to), this
            assert (x - y).abs().max() <= 1e-6, name        # platform's ATen addcmul twice -- one ulp of |position| <= 4
        else:
            assert torch.equal(x, y), name
    ref = oracle.ray_marching(o, d, near, far, ROI, grid, 0, 0.02, 1e10, 0.0, 128, True)
    hit = np.nonzero(ref[0][:, 1])[0]
    assert_equal(a.ridx_hit, hit, "ridx_hit")
    assert_equal(a.pack_infos, ref[0][hit].astype(np.int64), "pack_infos")
    assert_equal(a.ridx, ref[3].astype(np.int64), "ridx")


def test_sample_cache_and_second_march_agree(oracle, dev, monkeypatch):
    """emit = compaction of the samples cached by the count pass (default) vs. a second march (no cache):
    identical outputs, and both equal to the oracle"""
    from nr3d_lib_amd.bindings import _occ_grid
    res = (64, 64, 64)                                   # power-of-two ROI and res: the multiply-only probe path
    o, d, near, far = pinhole_rays(32, seed=13)
    grid = grids(res, 14)["random"]
    args = (o, d, near, far, ROI, grid, 0, 0.01, 1e10, 0.0, 300)
    got_cached, ref = run_both(oracle, dev, *args)
    monkeypatch.setattr(_occ_grid, "SAMPLE_CACHE_MAX_BYTES", 0)
    got_twice, _ = run_both(oracle, dev, *args)
    for a, b, r, n in zip(got_cached, got_twice, ref, ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]):
        assert_equal(a, r, name=f"cached/{n}")
        assert_equal(b, r, name=f"two-march/{n}")


# ---------------------------------------------------------------------------------------------------------------------
# occupancy-value grid maintenance
# ---------------------------------------------------------------------------------------------------------------------
def test_update_occ_val_grid_bit_exact(oracle, dev):
    from nr3d_lib_amd.models.accelerations.occgrid import (binarize, update_occ_val_grid_, update_occ_val_grid_idx_)
    rng = np.random.default_rng(21)
    res = (24, 20, 28)
    grid0 = rng.uniform(-0.5, 2.0, res).astype(np.float32)
    n = 50000                                              # ~3.7 samples per voxel: duplicates, and untouched voxels
    gidx = np.stack([rng.integers(0, r, n) for r in res], 1)
    val = rng.uniform(-1.0, 3.0, n).astype(np.float32)     # mixed signs: both branches of the float atomic max
    t = lambda a: torch.from_numpy(a).to(dev)
    for ema in (1.0, 0.95):
        g = t(grid0.copy())
        update_occ_val_grid_idx_(g, t(gidx), t(val), ema_decay=ema)
        assert_equal(g, oracle.occ_update_grid(grid0, gidx, val, ema), name=f"gidx ema={ema}")
    # positions in [-1, 1]^3 incl. the faces and points outside (clamped), voxel rule of the reference in fp32
    pts = rng.uniform(-1.05, 1.05, (n, 3)).astype(np.float32)
    pts[:10] = 1.0; pts[10:20] = -1.0
    g = t(grid0.copy())
    update_occ_val_grid_(g, t(pts), t(val), ema_decay=0.9)
    want = oracle.occ_update_grid(grid0, oracle.occ_gidx_from_pts(pts, res), val, 0.9)
    assert_equal(g, want, name="pts")
    # voxel indices outside the grid are dropped (no out-of-bounds write), and a sample value of -0.0 counts as a sample
    gi2 = gidx[:1000].copy()
    gi2[::7, 0] = res[0]; gi2[3::7, 2] = -1
    keep = (gi2 >= 0).all(1) & (gi2 < np.array(res)).all(1)
    v2 = val[:1000].copy()
    v2[keep.nonzero()[0][:50]] = -0.0
    gneg = -np.abs(grid0) - 1.0                                  # every old value < -0.0: a -0.0 sample must win
    g3 = t(gneg.copy())
    update_occ_val_grid_idx_(g3, t(gi2), t(v2), ema_decay=0.5)
    assert_equal(g3, oracle.occ_update_grid(gneg, gi2[keep], v2[keep], 0.5), name="out-of-range gidx dropped, -0.0 kept")
    # no samples: nothing changes; binarize incl. the mean rule
    g2 = t(grid0.copy())
    update_occ_val_grid_idx_(g2, t(gidx[:0]), t(val[:0]), ema_decay=0.5)
    assert_equal(g2, grid0, name="empty update")
    assert_equal(binarize(g, 0.7), want > 0.7, name="binarize")
    thr = min(0.7, float(want.mean()) - 1e-5)
    assert_equal(binarize(g, 0.7, consider_mean=True), want > np.float32(thr), name="binarize(mean)")


def test_update_batched_occ_val_grid(oracle, dev):
    from nr3d_lib_amd.models.accelerations.occgrid import (update_batched_occ_val_grid_, update_batched_occ_val_grid_idx_)
    rng = np.random.default_rng(22)
    B, res = 3, (12, 10, 14)
    grid0 = rng.uniform(0, 1, (B,) + res).astype(np.float32)
    n = 4000
    t = lambda a: torch.from_numpy(a).to(dev)
    gidx = np.stack([rng.integers(0, r, n) for r in res], 1)
    bidx = rng.integers(0, B, n)
    val = rng.uniform(0, 2, n).astype(np.float32)
    g = t(grid0.copy())
    update_batched_occ_val_grid_idx_(g, t(bidx), t(gidx), t(val), ema_decay=0.8)
    assert_equal(g, oracle.occ_update_grid(grid0, gidx, val, 0.8, bidx=bidx), name="per-sample bidx")
    gb = np.stack([np.stack([rng.integers(0, r, n) for r in res], 1) for _ in range(B)])
    vb = rng.uniform(0, 2, (B, n)).astype(np.float32)
    g = t(grid0.copy())
    update_batched_occ_val_grid_idx_(g, None, t(gb), t(vb), ema_decay=0.8)
    assert_equal(g, oracle.occ_update_grid(grid0, gb, vb, 0.8), name="batched input")
    pb = rng.uniform(-1, 1, (B, n, 3)).astype(np.float32)
    g = t(grid0.copy())
    update_batched_occ_val_grid_(g, t(pb), None, t(vb), ema_decay=1.0)
    want = oracle.occ_update_grid(grid0, oracle.occ_gidx_from_pts(pb, res), vb, 1.0)
    assert_equal(g, want, name="batched pts")


def test_occ_grid_ema_loop_feeds_the_marcher(oracle, dev):
    """sample -> query -> update -> binarize -> march: the producer/consumer pair end to end"""
    from nr3d_lib_amd.models.accelerations.occgrid import binarize, sample_pts_in_voxels, update_occ_val_grid_
    res = (32, 32, 32)
    torch.manual_seed(0)
    occ_val = torch.zeros(res, device=dev)
    gidx_full = torch.stack(torch.meshgrid(*[torch.arange(r, device=dev) for r in res], indexing="ij"), -1).view(-1, 3)
    density = lambda p: torch.exp(-((p.norm(dim=-1) - 0.6) / 0.08) ** 2)          # a spherical shell
    for _ in range(4):
        pts, vidx = sample_pts_in_voxels(gidx_full, 2 ** 16, torch.tensor(res, device=dev))
        assert pts.shape[0] == vidx.shape[0] and float(pts.abs().max()) <= 1.0
        update_occ_val_grid_(occ_val, pts, density(pts), ema_decay=0.95)
    occ = binarize(occ_val, 0.3)
    frac = float(occ.float().mean())
    assert 0.02 < frac < 0.5
    o, d, near, far = pinhole_rays(16, seed=4)
    got, ref = run_both(oracle, dev, o, d, near, far, ROI, occ.cpu().numpy(), 0, 0.02, 1e10, 0.0, 128)
    for g, r, nme in zip(got, ref, ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]):
        assert_equal(g, r, name=nme)
    assert got[1].shape[0] > 0


def test_occ_grid_ema_module(oracle, dev):
    """OccGridEma: init from the field, warm-up and steady-state steps, renderer samples, queries, shrink / rescale,
    checkpoint round trip; the update step itself bit-exact against the numpy restatement"""
    from nr3d_lib_amd.models.accelerations.occgrid import OccGridEma, get_occ_val_fn
    torch.manual_seed(1)
    density = lambda p: torch.exp(-((p.norm(dim=-1) - 0.6) / 0.08) ** 2)          # a spherical shell of radius 0.6
    acc = OccGridEma(32, occ_thre=0.3, ema_decay=0.9, n_steps_between_update=4, n_steps_warmup=8, device=dev,
                     init_cfg=dict(mode="from_net", num_steps=2, num_pts=2 ** 16),
                     update_from_net_cfg=dict(num_steps=2, num_pts=2 ** 15))
    with pytest.raises(AssertionError):
        acc.step(4, density)                                                       # init() first
    assert acc.init(density) and not acc.init(density) and bool(acc.is_initialized)
    ctr = (torch.stack(torch.meshgrid(*[torch.arange(32, device=dev)] * 3, indexing="ij"), -1) + 0.5) / 16 - 1
    r = ctr.norm(dim=-1)
    inside = (r > 0.5) & (r < 0.7)
    assert float(acc.occ_grid[inside].float().mean()) > 0.9 and float(acc.occ_grid[(r < 0.3) | (r > 0.9)].float().mean()) == 0
    # the update rule on explicit samples == the oracle's scatter-max + decay
    pts = torch.rand(5000, 3, device=dev) * 2 - 1
    val = density(pts)
    before = acc.occ_val_grid.cpu().numpy().copy()
    acc.should_collect_samples = False
    acc._step_update_occ(pts, val)
    gidx = oracle.occ_gidx_from_pts(pts.cpu().numpy(), (32, 32, 32))
    assert_equal(acc.occ_val_grid, oracle.occ_update_grid(before, gidx, val.cpu().numpy(), 0.9), name="occ_val_grid")
    assert_equal(acc.occ_grid, acc.occ_val_grid.cpu().numpy() > 0.3, name="occ_grid")
    acc.should_collect_samples = True
    # step(): only every n_steps_between_update iterations; warm-up (uniform) and steady-state (1/2 + 1/4 + 1/4) sampling
    assert not acc.step(0, density) and not acc.step(3, density)
    assert acc.step(4, density) and acc.step(12, density)
    # samples from the renderer are merged at the next update (training mode only)
    hot = torch.tensor([[0.05, 0.05, 0.05]], device=dev)                            # empty region of the field
    acc.eval(); acc.collect_samples(hot, torch.tensor([5.0], device=dev))
    assert float(acc._occ_val_grid_pcl.max()) == 0
    acc.train(); acc.collect_samples(hot, torch.tensor([5.0], device=dev))
    assert float(acc._occ_val_grid_pcl.max()) == 5.0
    acc.step(16, density)
    assert bool(acc.query(hot)[0]) and float(acc._occ_val_grid_pcl.max()) == 0 and float(acc.occ_val_grid[16, 16, 16]) == 5.0
    assert bool(acc.query(torch.tensor([[0.6, 0.0, 0.0]], device=dev))[0]) and not bool(acc.query(torch.tensor([[0.95, 0.95, 0.95]], device=dev))[0])
    p_occ = acc.sample_pts_in_occupied(1000)
    assert bool(acc.query(p_occ).all())
    # shrink: the tight box around the shell; rescale keeps the shell where it is in world coordinates
    aabb = torch.tensor([[-1., -1, -1], [1, 1, 1]], device=dev)
    new = acc.try_shrink(aabb)
    assert float(new[0].max()) < -0.6 and float(new[1].min()) > 0.6 and float(new.abs().max()) <= 1.0
    acc.occ_val_grid[16, 16, 16] = 0
    acc.rescale_volume(aabb, torch.tensor([[-0.8, -0.8, -0.8], [0.8, 0.8, 0.8]], device=dev))
    ctr2 = ctr * 0.8
    r2 = ctr2.norm(dim=-1)
    assert float(acc.occ_grid[(r2 > 0.55) & (r2 < 0.65)].float().mean()) > 0.8 and float(acc.occ_grid[r2 < 0.35].float().mean()) == 0
    # checkpoints: the grid's resolution comes from the file
    other = OccGridEma(8, occ_thre=0.3, device=dev, init_cfg=dict(mode="constant", constant_value=0.0))
    other.load_state_dict(acc.state_dict())
    assert_equal(other.occ_grid, acc.occ_grid.cpu().numpy()); assert other.resolution.tolist() == [32, 32, 32] and bool(other.is_initialized)
    const = OccGridEma([8, 6, 4], occ_thre=0.3, device=dev, init_cfg=dict(mode="constant", constant_value=1.0))
    const.init()
    assert bool(const.occ_grid.all()) and tuple(const.occ_grid.shape) == (8, 6, 4)
    f = get_occ_val_fn("sdf", inv_s=10.0)
    assert abs(float(f(torch.zeros(1))) - 1.0) < 1e-6 and float(f(torch.tensor([1.0]))) < 0.03
    assert float(get_occ_val_fn("raw_sdf")(torch.tensor([0.25]))) == 0.75


def test_occ_grid_ema_batched_module(oracle, dev):
    """OccGridEmaBatched: per-entry fields, per-point and batched sample layouts, restricted updates, queries"""
    from nr3d_lib_amd.models.accelerations.occgrid import OccGridEmaBatched
    torch.manual_seed(2)
    radii = torch.tensor([0.3, 0.6, 0.85], device=dev)
    field = lambda pts, bidx: torch.exp(-((pts.norm(dim=-1) - radii[bidx]) / 0.08) ** 2)      # one shell radius per entry
    acc = OccGridEmaBatched(3, 24, occ_thre=0.3, ema_decay=0.9, n_steps_between_update=2, n_steps_warmup=4, device=dev,
                            init_cfg=dict(mode="net", num_steps=2, num_pts_per_batch=2 ** 15),
                            update_from_net_cfg=dict(num_steps=1, num_pts_per_batch=2 ** 14))
    assert acc.init(field) and tuple(acc.occ_grid.shape) == (3, 24, 24, 24)
    ctr = (torch.stack(torch.meshgrid(*[torch.arange(24, device=dev)] * 3, indexing="ij"), -1) + 0.5) / 12 - 1
    r = ctr.norm(dim=-1)
    for b in range(3):
        shell = (r - radii[b]).abs() < 0.03
        assert float(acc.occ_grid[b][shell].float().mean()) > 0.9 and float(acc.occ_grid[b][(r - radii[b]).abs() > 0.25].float().mean()) == 0
    # the update on explicit per-point samples == the oracle (entries as a leading grid dim)
    pts = torch.rand(4000, 3, device=dev) * 2 - 1
    bidx = torch.randint(0, 3, (4000,), device=dev)
    val = field(pts, bidx)
    before = acc.occ_val_grid.cpu().numpy().copy()
    acc.should_collect_samples = False
    acc._step_update_occ(pts, bidx, val)
    gidx = oracle.occ_gidx_from_pts(pts.cpu().numpy(), (24, 24, 24))
    assert_equal(acc.occ_val_grid, oracle.occ_update_grid(before, gidx, val.cpu().numpy(), 0.9, bidx=bidx.cpu().numpy()), name="per-point")
    # batched layout: [entries, n, 3] without bidx
    ptsb = torch.rand(3, 500, 3, device=dev) * 2 - 1
    valb = field(ptsb, torch.arange(3, device=dev).view(3, 1).expand(3, 500))
    before = acc.occ_val_grid.cpu().numpy().copy()
    acc._step_update_occ(ptsb, None, valb)
    gb = oracle.occ_gidx_from_pts(ptsb.reshape(-1, 3).cpu().numpy(), (24, 24, 24)).reshape(3, 500, 3)
    assert_equal(acc.occ_val_grid, oracle.occ_update_grid(before, gb, valb.cpu().numpy(), 0.9), name="batched")
    acc.should_collect_samples = True
    # steps; an update restricted to entry 2 leaves the others alone
    assert not acc.step(1, field) and acc.step(2, field) and acc.step(6, field)
    snap = acc.occ_val_grid.clone()
    acc._step(8, field, within_bi=torch.tensor([2], device=dev), num_steps=1, num_pts_per_batch=2 ** 12)
    assert torch.equal(snap[:2], acc.occ_val_grid[:2]) and not torch.equal(snap[2], acc.occ_val_grid[2])
    # renderer samples, queries (per point and batched), sampling inside occupied voxels
    acc.train()
    acc.collect_samples(torch.zeros(1, 3, device=dev), torch.tensor([1], device=dev), torch.tensor([3.0], device=dev))
    acc.step(10, field)
    assert acc.query(torch.zeros(1, 3, device=dev), torch.tensor([1], device=dev)).tolist() == [True]
    assert acc.query(torch.zeros(1, 3, device=dev), torch.tensor([0], device=dev)).tolist() == [False]
    qb = acc.query(torch.tensor([[[0.3, 0, 0]], [[0.6, 0, 0]], [[0.85, 0, 0]]], device=dev))
    assert qb.tolist() == [[True], [True], [True]]
    p, b = acc.sample_pts_in_occupied(600)
    assert bool(acc.query(p, b).all()) and set(b.unique().tolist()) == {0, 1, 2}
    p2, b2 = acc.sample_pts_in_occupied(100, within_bi=torch.tensor([2], device=dev))
    assert set(b2.unique().tolist()) == {0}                                  # local to within_bi
    other = OccGridEmaBatched(1, 4, occ_thre=0.3, device=dev, init_cfg=dict(mode="constant", constant_value=0.0))
    other.load_state_dict(acc.state_dict())
    assert other.num_batches == 3 and torch.equal(other.occ_grid, acc.occ_grid)


def test_batched_accel_end_to_end(oracle, dev):
    """BatchedBlockSpace.cur_batch__ray_test -> OccGridAccelBatched_{Getter,Ema}.cur_batch__ray_march (the batched HIP
    marcher) against the oracle marcher on the grids the accelerator holds: pack offsets / indices / depths bit-exact"""
    from nr3d_lib_amd.models.accelerations.occgrid_accel import OccGridAccelBatched_Ema, OccGridAccelBatched_Getter
    from nr3d_lib_amd.models.spatial import BatchedBlockSpace
    torch.manual_seed(3)
    B, N = 3, 200
    space = BatchedBlockSpace(aabb=[[-1, -1.5, -1], [1, 1.5, 1]], device=dev)
    radii = torch.tensor([0.45, 0.7, 0.9], device=dev)
    field = lambda x, bidx: (x.norm(dim=-1) < radii[bidx]).float()
    g = torch.Generator().manual_seed(4)
    rays_o = torch.tensor([0.0, 0.0, -4.0]).expand(B, N, 3).clone().to(dev)
    rays_d = (torch.randn(B, N, 3, generator=g) * 0.25 + torch.tensor([0, 0, 1.0])).to(dev)
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    rt = space.cur_batch__ray_test(rays_o, rays_d)
    assert rt["num_rays"] > 100

    getter = OccGridAccelBatched_Getter(space, resolution=[16, 24, 16], occ_thre=0.5, num_steps=3, num_pts_per_batch=2 ** 15,
                                        device=dev)
    getter.set_condition(B, val_query_fn_normalized_x_bi=field)
    ema = OccGridAccelBatched_Ema(space, num_batches=5, resolution=[16, 24, 16], occ_val_fn_cfg=dict(type="density"), occ_thre=0.5,
                                  init_cfg=dict(mode='from_net', num_steps=3, num_pts_per_batch=2 ** 15), device=dev)
    ema.set_condition(5, ins_inds_per_batch=torch.arange(5, device=dev))
    radii5 = torch.tensor([0.3, 0.45, 0.7, 0.9, 0.2], device=dev)
    field5 = lambda x, bidx: (x.norm(dim=-1) < radii5[bidx]).float()       # indexed by INSTANCE
    ema.init(field5)
    ema.set_condition(B, ins_inds_per_batch=torch.tensor([1, 2, 3], device=dev))      # the same three objects
    for name, acc in (("getter", getter), ("ema", ema)):
        grids_b = acc.occ_grid_per_batch
        assert tuple(grids_b.shape) == (B, 16, 24, 16) and grids_b.dtype == torch.bool and bool(grids_b.any())
        ret = acc.cur_batch__ray_march(rt["rays_o"], rt["rays_d"], rt["rays_bidx"], near=rt["near"], far=rt["far"],
                                       step_size=0.02, max_steps=128)
        roi = np.tile(np.array([-1, -1, -1, 1, 1, 1], np.float32), (B, 1))
        ref = oracle.ray_marching(rt["rays_o"].cpu().numpy(), rt["rays_d"].cpu().numpy(), rt["near"].cpu().numpy(),
                                  rt["far"].cpu().numpy(), roi, grids_b.cpu().numpy(), 0, 0.02, 1e10, 0.0, 128, True,
                                  batch_inds=rt["rays_bidx"].cpu().numpy().astype(np.int32))
        hit = np.nonzero(ref[0][:, 1])[0]
        assert ret.num_hit_rays == len(hit) > 0, name
        assert_equal(ret.ridx_hit, hit, f"{name}/ridx_hit")
        assert_equal(ret.pack_infos, ref[0][hit].astype(np.int64), f"{name}/pack_infos")
        assert_equal(ret.depth_samples, ref[1][:, 0], f"{name}/depth_samples")
        assert_equal(ret.bidx, ref[4].astype(np.int64), f"{name}/bidx")
        assert_equal(ret.gidx, ref[5].astype(np.int64), f"{name}/gidx")
        pts, bi = acc.cur_batch__sample_pts_in_occupied(256)
        assert bool(acc.cur_batch__query_occupancy(pts, bi).all())
    # the EMA variant keeps learning from the batch's samples
    ema.train()
    ema.cur_batch__collect_samples(torch.zeros(4, 3, device=dev), torch.tensor([0, 1, 2, 0], device=dev), torch.ones(4, device=dev))
    ema.cur_batch__step(16, field5)
    assert ema.debug_stats()["num_occupied"] > 0


def test_marcher_two_threads_two_streams(oracle, dev):
    """safe from several Python threads on different streams.  Two host threads march DIFFERENT ray
    sets (different sample counts) on their own streams, many times over, each checking every result against the oracle.
    The op's one readback (the sample count that sizes t_starts / ridx / ...) goes through a pinned staging buffer that was
    shared per (device, n) until round 3 -- a thread could then read the other's count and allocate too few samples
    (round-3 review and advisor finding; thread-local since round 4, nr3d_lib_amd/_hip.py: host_i64 / wait_i64, the pinned words the count kernels write into)."""
    import threading
    from nr3d_lib_amd.bindings import _occ_grid
    res = (48, 48, 48)
    grid = grids(res, 5)["random"]
    step = 2 * 3 ** 0.5 / 96
    sets = []
    for seed, side in ((11, 20), (12, 33)):
        o, d, near, far = pinhole_rays(side, seed=seed)
        ref = oracle.ray_marching(o, d, near, far, ROI, grid, 0, step, 1e10, 0.0, 96, True)
        sets.append((o, d, near, far, ref))
    assert sets[0][4][1].shape[0] != sets[1][4][1].shape[0]
    errors = []
    barrier = threading.Barrier(2)

    def work(k):
        try:
            o, d, near, far, ref = sets[k]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                args = (t(o), t(d), t(near), t(far), t(ROI), t(grid))
                stream.synchronize()
                barrier.wait()
                for it in range(60):
                    got = _occ_grid.ray_marching(*args, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, 96, True)
                    assert got[1].shape[0] == ref[1].shape[0], f"thread {k} iteration {it}: {got[1].shape[0]} samples, oracle {ref[1].shape[0]}"
                    if it % 10 == 0:
                        stream.synchronize()
                        for g, r, n in zip(got, ref, ["packed_info", "t_starts", "t_ends", "ridx", "gidx"]):
                            assert np.array_equal(g.cpu().numpy(), r), f"thread {k} iteration {it}: {n} differs"
                stream.synchronize()
        except Exception as e:                      # noqa: BLE001 -- reported in the main thread
            errors.append(e)
            try:
                barrier.abort()
            except Exception:                       # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("scene", ["c3_random", "shell_rgbless", "sparse_misses"])
def test_march_composite_in_one_call(oracle, dev, scene):
    """round 6: ray_marching_composite (count -> scan -> emit -> alpha + composite without the host in between, bound-sized buffers,
    one lazy readback) against the two-phase chain it replaces -- every march output bit-equal to the oracle's marcher, alpha / vw /
    mask / depth / rgb and all gradients bit-equal to tau_to_alpha + packed_composite forward / backward on the same samples"""
    from nr3d_lib_amd.bindings import _occ_grid, _pack_ops
    rng = np.random.default_rng(11)
    if scene == "c3_random":                                     # configs[2]: 128^3, 4096 rays, <= 512 samples per ray
        res, side, step, max_steps, with_rgb = (128, 128, 128), 64, 2 * 3 ** 0.5 / 512, 512, True
        grid = rng.random(res) > 0.5
    elif scene == "shell_rgbless":
        res, side, step, max_steps, with_rgb = (48, 40, 56), 40, 0.01, 300, False
        grid = grids(res, 3)["shell"]
    else:                                                        # most rays get no sample at all; a few cells only
        res, side, step, max_steps, with_rgb = (32, 32, 32), 33, 0.02, 64, True
        grid = rng.random(res) > 0.995
    o, d, near, far = pinhole_rays(side, seed=7)
    n = o.shape[0]
    ref = oracle.ray_marching(o, d, near, far, ROI, grid, 0, np.float32(step), 1e10, 0.0, max_steps, True)
    S = ref[1].shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ot, dt_, nt, ft, rt, gt = t(o), t(d), t(near), t(far), t(ROI), t(grid)
    sigma = t((10.0 * rng.random(S + 5) ** 2).astype(np.float32))               # more rows than samples: allowed
    rgb = t(rng.random((S + 5, 3), dtype=np.float32)) if with_rgb else None
    thre = 0.02
    mc = _occ_grid.ray_marching_composite(ot, dt_, nt, ft, rt, gt, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, max_steps, sigma,
                                          rgb, 1e-4, thre, True)
    gm, gd, gc = t(rng.standard_normal(n).astype(np.float32)), t(rng.standard_normal(n).astype(np.float32)), \
        (t(rng.standard_normal((n, 3)).astype(np.float32)) if with_rgb else None)
    mc.backward(gm, gd, gc, need_sigma=True)                     # enqueued BEFORE anything was read back
    out = mc.result()
    ga, gtt, gr, gs = mc.grads()
    assert (mc.totals()[0], out["n_hit"]) == (S, int((ref[0][:, 1] > 0).sum()))
    for name, r in zip(["packed_info", "t_starts", "t_ends", "ridx32", "gidx"], ref):
        assert_equal(out[name].view(r.shape), r, name=name)
    hit = np.nonzero(ref[0][:, 1])[0]
    assert_equal(out["ridx_hit"], hit, "ridx_hit")
    assert_equal(out["pack_infos"], ref[0][hit].astype(np.int64), "pack_infos")
    assert_equal(out["ridx"], ref[3].astype(np.int64), "ridx (int64)")
    # the two-phase chain on the same samples
    m = _occ_grid.ray_marching_finished(ot, dt_, nt, ft, rt, gt, _occ_grid.ContractionType.AABB, step, 1e10, 0.0, max_steps, True)
    for k in ("deltas", "samples"):
        assert_equal(out[k], m[k], k)
    alpha = _pack_ops.tau_to_alpha_forward(sigma[:S].contiguous(), m["deltas"])
    rgb_s = rgb[:S].contiguous() if with_rgb else None
    vw, mask, depth, col = _pack_ops.packed_composite_forward(alpha, m["t_starts"], rgb_s, m["pack_infos"], m["ridx_hit"], n, 1e-4, thre,
                                                              True, packs_tile=True)
    assert_equal(out["alpha"], alpha, "alpha"); assert_equal(out["vw"], vw, "vw")
    assert_equal(out["mask"], mask, "mask"); assert_equal(out["depth"], depth, "depth")
    if with_rgb:
        assert_equal(out["rgb"], col, "rgb")
    assert float(mask.abs().sum()) > 0 and int((out["mask"] == 0).sum()) >= n - len(hit)
    ga2, gt2, gr2 = _pack_ops.packed_composite_backward(alpha, vw, m["t_starts"], rgb_s, m["pack_infos"], m["ridx_hit"], 1e-4, thre, True,
                                                        mask, depth, gm, gd, gc, None, packs_tile=True)
    assert_equal(ga, ga2, "grad_alpha"); assert_equal(gtt, gt2, "grad_t")
    if with_rgb:
        assert_equal(gr, gr2, "grad_rgb")
    assert_equal(gs, _pack_ops.tau_to_alpha_backward(sigma[:S].contiguous(), m["deltas"], ga2), "grad_sigma")


def test_march_composite_handles_and_limits(dev):
    """two outstanding handles keep their own totals; sigma with fewer rows than samples raises at the readback; outside the one-call
    range the binding says so instead of allocating the bound"""
    from nr3d_lib_amd.bindings import _occ_grid
    rng = np.random.default_rng(2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d, near, far = pinhole_rays(24, seed=1)
    args = lambda g: (t(o), t(d), t(near), t(far), t(ROI), t(g), _occ_grid.ContractionType.AABB, 0.02, 1e10, 0.0, 128)
    g1, g2 = rng.random((32, 32, 32)) > 0.5, rng.random((32, 32, 32)) > 0.9
    sigma = torch.ones(24 * 24 * 128, device=dev)
    a = _occ_grid.ray_marching_composite(*args(g1), sigma)
    b = _occ_grid.ray_marching_composite(*args(g2), sigma)
    Sa, Sb = a.totals()[0], b.totals()[0]
    assert Sa > Sb > 0
    assert _occ_grid.ray_marching(*args(g1), True)[1].shape[0] == Sa and _occ_grid.ray_marching(*args(g2), True)[1].shape[0] == Sb
    short = _occ_grid.ray_marching_composite(*args(g1), sigma[:Sa - 1].contiguous())
    with pytest.raises(RuntimeError, match="rows"):
        short.totals()
    big = (t(np.repeat(o, 64, 0)), t(np.repeat(d, 64, 0)), t(np.repeat(near, 64)), t(np.repeat(far, 64))) + args(g1)[4:10] + (4096,)
    with pytest.raises(ValueError, match="two-phase"):
        _occ_grid.ray_marching_composite(*big, sigma)
