"""CPU: the forest restatement of the oracle (octree identify, LoTD forest, forest marcher) and the octree host logic.

The reference holds no vectors for the forest path and cannot be run here (CUDA + kaolin), so the forest oracle is
anchored on an identity with the single-block oracle, which IS pinned against the reference's goldens: a forest level
of resolution R evaluated in block b equals a plain Dense level of resolution R + 2 on the grid obtained by padding
b's table with the facing layer of its neighbours (zeros where there is none) -- same locator scale (R + 2) - 2 = R,
same corner order, so the equality is bit for bit, for values, Jacobians, parameter gradients and both second-order
terms."""
import numpy as np
import pytest
import torch

FORESTS = {
    # name: (level, block coordinates)
    "plus": (2, [(1, 1, 1), (2, 1, 1), (1, 2, 1), (0, 1, 1), (1, 1, 2), (2, 2, 1)]),
    "two": (1, [(0, 0, 0), (1, 0, 0)]),
    "scatter": (3, [(0, 0, 0), (7, 7, 7), (3, 4, 5), (4, 4, 5), (3, 5, 5), (3, 4, 4), (2, 4, 5), (0, 1, 0), (6, 7, 7)]),
    "single": (0, [(0, 0, 0)]),
}


def _padded(params_by_block, n_params, k, index_of, R, F):
    """(R+2)^3 x F grid around block k: interior = own table, shell = facing layer of the neighbours (or 0)"""
    P = np.zeros((R + 2, R + 2, R + 2, F), np.float32)
    owner = -np.ones((R + 2, R + 2, R + 2, 4), np.int64)         # (block, x, y, z) of every padded node
    for ix in range(R + 2):
        for iy in range(R + 2):
            for iz in range(R + 2):
                kk, l = list(k), [0, 0, 0]
                for d, i in enumerate((ix, iy, iz)):
                    if i == 0:
                        kk[d] -= 1; l[d] = R - 1
                    elif i == R + 1:
                        kk[d] += 1; l[d] = 0
                    else:
                        l[d] = i - 1
                b = index_of.get(tuple(kk))
                if b is not None:
                    P[ix, iy, iz] = params_by_block[b].reshape(R, R, R, F)[l[0], l[1], l[2]]
                    owner[ix, iy, iz] = (b, *l)
    return P, owner


@pytest.mark.parametrize("name", ["plus", "two", "single"])
@pytest.mark.parametrize("smooth", [False, True])
def test_forest_dense_level_equals_plain_level_on_padded_grid(oracle, name, smooth):
    level, blocks = FORESTS[name]
    fo = oracle.forest_from_blocks(blocks, level)
    R, F, N = 5, 2, 300
    rng = np.random.default_rng(3)
    meta = oracle.lotd_create_meta(3, [R], [F], ["dense"], use_smooth_step=smooth)
    meta_pad = oracle.lotd_create_meta(3, [R + 2], [F], ["dense"], use_smooth_step=smooth)
    npb = meta.n_params
    params = rng.standard_normal(fo.n_trees * npb).astype(np.float32)
    by_block = [params[b * npb:(b + 1) * npb] for b in range(fo.n_trees)]
    index_of = {tuple(k): i for i, k in enumerate(fo.block_ks.tolist())}
    x = rng.random((N, 3)).astype(np.float32)
    g = rng.standard_normal((N, F)).astype(np.float32)
    v = rng.standard_normal((N, 3)).astype(np.float32)
    for b in range(min(fo.n_trees, 3)):
        P, owner = _padded(by_block, npb, fo.block_ks[b].tolist(), index_of, R, F)
        bi = np.full(N, b, np.int64)
        y, j = oracle.lotd_forest_fwd(meta, fo, x, params, block_inds=bi, need_dydx=True)
        y2, j2 = oracle.lotd_fwd(meta_pad, x, P.ravel(), need_dydx=True)
        np.testing.assert_array_equal(y, y2)
        np.testing.assert_array_equal(j, j2)
        np.testing.assert_array_equal(oracle.lotd_forest_bwd_bwd_dx(meta, fo, v, g, x, params, block_inds=bi),
                                      oracle.lotd_bwd_bwd_dx(meta_pad, v, g, x, P.ravel()))
        for second in (None, v):
            gp = oracle.lotd_forest_bwd_dparam(meta, fo, g, x, params, block_inds=bi, dL_ddLdx=second)
            gp2 = (oracle.lotd_bwd_dparam(meta_pad, g, x, P.ravel()) if second is None
                   else oracle.lotd_bwd_bwd_dparam(meta_pad, second, g, x, P.ravel())).reshape(R + 2, R + 2, R + 2, F)
            want = np.zeros((fo.n_trees, R, R, R, F), np.float32)
            has = owner[..., 0] >= 0
            ob, ox, oy, oz = (owner[..., i][has] for i in range(4))
            want[ob, ox, oy, oz] = gp2[has]                     # injective: every padded node has one owner
            np.testing.assert_array_equal(gp.reshape(want.shape), want)
    # continuity off: the shell contributes nothing -> same as a forest of isolated blocks
    fo.continuity_enabled = False
    lone = oracle.forest_from_blocks([blocks[0]], level)
    b = index_of[tuple(blocks[0])]
    y_off, _ = oracle.lotd_forest_fwd(meta, fo, x, params, block_inds=np.full(N, b, np.int64))
    y_lone, _ = oracle.lotd_forest_fwd(meta, lone, x, by_block[b], block_inds=np.zeros(N, np.int64))
    np.testing.assert_array_equal(y_off, y_lone)


def test_forest_encoding_is_continuous_across_block_faces(oracle):
    level, blocks = FORESTS["plus"]
    fo = oracle.forest_from_blocks(blocks, level)
    meta = oracle.lotd_create_meta(3, [4, 6, 5, 7, 6], [2, 2, 2, 2, 2], ["dense", "vm", "nplanemul", "cp", "hash"], 257)
    rng = np.random.default_rng(0)
    params = rng.standard_normal(fo.n_trees * meta.n_params).astype(np.float32)
    index_of = {tuple(k): i for i, k in enumerate(fo.block_ks.tolist())}
    a, b = index_of[(1, 1, 1)], index_of[(2, 1, 1)]                  # neighbours along x
    yz = rng.random((64, 2)).astype(np.float32) * 0.5 + 0.25         # keep y, z away from the other faces
    eps = 1e-6
    left = np.concatenate([np.full((64, 1), 1 - eps, np.float32), yz], 1)
    right = np.concatenate([np.full((64, 1), eps, np.float32), yz], 1)
    ya, _ = oracle.lotd_forest_fwd(meta, fo, left, params, block_inds=np.full(64, a, np.int64))
    yb, _ = oracle.lotd_forest_fwd(meta, fo, right, params, block_inds=np.full(64, b, np.int64))
    dense_hash = [0, 1, 8, 9]                                         # the Dense and Hash levels' features
    # Dense is exactly continuous; the hash of a REMAPPED position differs between the two blocks' tables, like the
    # product types whose factors live in different blocks -- those are only bounded, so check Dense strictly
    assert np.abs(ya[:, :2] - yb[:, :2]).max() < 1e-4
    fo.continuity_enabled = False
    ya0, _ = oracle.lotd_forest_fwd(meta, fo, left, params, block_inds=np.full(64, a, np.int64))
    yb0, _ = oracle.lotd_forest_fwd(meta, fo, right, params, block_inds=np.full(64, b, np.int64))
    assert np.abs(ya0[:, :2] - yb0[:, :2]).max() > 1e-2 and dense_hash


def test_forest_skips_and_batching(oracle):
    level, blocks = FORESTS["scatter"]
    fo = oracle.forest_from_blocks(blocks, level)
    meta = oracle.lotd_create_meta(3, [4, 6], [2, 4], ["dense", "hash"], 97)
    rng = np.random.default_rng(1)
    T, n = fo.n_trees, 20
    params = rng.standard_normal(T * meta.n_params).astype(np.float32)
    x = rng.random((T * n, 3)).astype(np.float32)
    bi = np.repeat(np.arange(T), n).astype(np.int64)
    y, _ = oracle.lotd_forest_fwd(meta, fo, x, params, block_inds=bi)
    yb, _ = oracle.lotd_forest_fwd(meta, fo, x, params, batch_data_size=n)          # batched: blocks in order
    np.testing.assert_array_equal(y, yb)
    # block_offsets: the same tables stored in another order
    perm = rng.permutation(T)
    offs = np.empty(T, np.int64); offs[perm] = np.arange(T) * meta.n_params
    shuffled = np.concatenate([params[b * meta.n_params:(b + 1) * meta.n_params] for b in perm])
    yo, _ = oracle.lotd_forest_fwd(meta, fo, x, shuffled, block_inds=bi, block_offsets=offs)
    np.testing.assert_array_equal(y, yo)
    bi2 = bi.copy(); bi2[::3] = -1
    ys, js = oracle.lotd_forest_fwd(meta, fo, x, params, block_inds=bi2, need_dydx=True)
    assert (ys[::3] == 0).all() and (js[::3] == 0).all()
    np.testing.assert_array_equal(ys[1::3], y[1::3])
    yl, _ = oracle.lotd_forest_fwd(meta, fo, x, params, block_inds=bi, max_level=0)
    np.testing.assert_array_equal(yl[:, :2], y[:, :2]); assert (yl[:, 2:] == 0).all()
    bad = oracle.lotd_create_meta(3, [4], [2], ["cpfast"])
    with pytest.raises(RuntimeError):
        oracle.lotd_forest_fwd(bad, fo, x, np.zeros(T * bad.n_params, np.float32), block_inds=bi)


@pytest.mark.parametrize("name", list(FORESTS))
def test_identify_and_octree_builders(oracle, name):
    from nr3d_lib_amd.models.spatial.forest import _walk_octree, octree_from_corners
    level, blocks = FORESTS[name]
    fo = oracle.forest_from_blocks(blocks, level)
    index_of = {tuple(k): i for i, k in enumerate(fo.block_ks.tolist())}
    assert sorted(index_of) == sorted(blocks)
    side = 1 << level
    ks = np.array([(x, y, z) for x in range(-1, side + 1) for y in range(-1, side + 1) for z in range(-1, side + 1)], np.int16)
    got = oracle.forest_identify(fo, ks)
    want = np.array([index_of.get(tuple(k), -1 - fo.level_poffset) + fo.level_poffset for k in ks.tolist()])
    np.testing.assert_array_equal(got, want)                            # node index, -1 when there is no such block
    # the product's sort-based builder gives the same SPC layout as the oracle's level-by-level sets
    octree, exsum, points, pyramid = octree_from_corners(torch.tensor(blocks), level)
    np.testing.assert_array_equal(octree.numpy(), fo.octree)
    np.testing.assert_array_equal(exsum.numpy(), fo.exsum)
    assert int(pyramid[1, level]) == fo.level_poffset and int(pyramid[0, level]) == fo.n_trees
    np.testing.assert_array_equal(points[fo.level_poffset:].numpy(), fo.block_ks)
    walked, counts = _walk_octree(octree, level)
    np.testing.assert_array_equal(walked.numpy(), points.numpy())
    np.testing.assert_array_equal(counts, pyramid[0, :level + 1].numpy())


def _segments(fo, o, d, near, far):
    """per-ray (block, entry, exit) of the blocks a ray crosses, sorted by entry (numpy slab test)"""
    wo, wb = fo.world_origin.astype(np.float32), fo.world_block_size.astype(np.float32)
    bmin = fo.block_ks.astype(np.float32) * wb + wo
    sb, se, sx, sp = [], [], [], []
    for i in range(o.shape[0]):
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (bmin - o[i]) / d[i], (bmin + wb - o[i]) / d[i]
        tn = np.maximum(np.minimum(t0, t1).max(1), near[i]); tf = np.minimum(np.maximum(t0, t1).min(1), far[i])
        hit = np.nonzero((tf > tn) & (tf > 0))[0]
        hit = hit[np.argsort(tn[hit], kind="stable")]
        sp.append((len(sb), len(hit)))
        sb += hit.tolist(); se += tn[hit].tolist(); sx += tf[hit].tolist()
    return (np.array(sb, np.int32), np.array(se, np.float32), np.array(sx, np.float32), np.array(sp, np.int32).reshape(-1, 2))


def forest_scene(oracle, seed=0, n_rays=200, res=8):
    level, blocks = FORESTS["plus"]
    fo = oracle.forest_from_blocks(blocks, level, world_origin=(-2.0, -2.0, -2.0), world_block_size=(1.0, 1.0, 1.0))
    rng = np.random.default_rng(seed)
    grid = rng.random((fo.n_trees, res, res, res)) > 0.6
    o = (rng.random((n_rays, 3)) * 0.6 - 2.8).astype(np.float32)
    tgt = (rng.random((n_rays, 3)) * 2.0 - 1.5).astype(np.float32)
    d = tgt - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    near, far = np.full(n_rays, 0.05, np.float32), np.full(n_rays, 6.0, np.float32)
    return fo, grid, o, d, near, far, _segments(fo, o, d, near, far)


def test_forest_marcher_properties(oracle):
    fo, grid, o, d, near, far, (sb, se, sx, sp) = forest_scene(oracle)
    step = 0.04
    pi, ts, te, ridx, bl, gi = oracle.forest_ray_marching(fo, o, d, near, far, sb, se, sx, sp, grid, step, 1e10, 0.0, 64, True)
    assert pi[:, 1].sum() == ts.shape[0] > 0 and (pi[:, 1] <= 64).all()
    np.testing.assert_array_equal(ridx, np.repeat(np.arange(o.shape[0]), pi[:, 1]))
    assert grid.reshape(-1)[gi].all()                                   # every sample sits in an occupied voxel ...
    vol = grid[0].size
    np.testing.assert_array_equal(gi // vol, bl)                        # ... of the block it reports
    tm = (ts + te)[:, 0] * 0.5
    for i in np.nonzero(pi[:, 1])[0][:50]:
        b0, n = pi[i]
        segs = range(sp[i, 0], sp[i, 0] + sp[i, 1])
        for j in range(b0, b0 + n):                                     # mid-points lie inside a segment of that block
            assert any(sb[s] == bl[j] and se[s] - 1e-5 <= tm[j] <= sx[s] + 1e-5 for s in segs)
        assert (np.diff(ts[b0:b0 + n, 0]) > 0).all()
    # max_steps caps the per-ray count; no segments -> no samples
    pi2 = oracle.forest_ray_marching(fo, o, d, near, far, sb, se, sx, sp, grid, step, 1e10, 0.0, 3, False)[0]
    np.testing.assert_array_equal(pi2[:, 1], np.minimum(pi[:, 1], 3))
    empty = oracle.forest_ray_marching(fo, o, d, near, far, sb[:0], se[:0], sx[:0], np.zeros_like(sp), grid, step, 1e10, 0.0, 64, True)
    assert empty[0][:, 1].sum() == 0 and empty[1].shape == (0, 1)
