"""CPU: nr3d_lib_amd.models.grid_encodings.lotd.lotd_helpers against the reference's own helpers -- the parameter layout
of every level type (141 (level, op, dim) records, tests/golden/ref_python.json), the vertex coordinates and the N-linear
table sampler (tests/golden/ref_lotd_helpers.npz, made by make_golden_helpers.py)."""
import json
import os

import numpy as np
import pytest
import torch

from util import LOTD_CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _meta(case):
    from nr3d_lib_amd.bindings import _lotd
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    return _lotd.LoDMeta(D, res, nf, types, T, smooth)


def test_level_param_index_shape_matches_reference():
    from nr3d_lib_amd.models.grid_encodings.lotd.lotd_helpers import get_level_param, level_param_index_shape
    recs = json.load(open(os.path.join(GOLD, "ref_python.json")))["layout"]
    metas = {}
    for r in recs:
        m = metas.setdefault(r["case"], _meta(r["case"]))
        index, shape = level_param_index_shape(m, r["level"], r["op"], r["dim"])
        assert (index[0].start, index[0].stop, list(shape)) == (r["start"], r["stop"], r["shape"]), r
        p = torch.arange(m.n_params, dtype=torch.float32)
        v = get_level_param(p, m, r["level"], r["op"], r["dim"])
        assert tuple(v.shape) == tuple(r["shape"]) and v.data_ptr() == p.data_ptr() + 4 * r["start"]   # a view, not a copy
    assert len(recs) == 141 and len(metas) >= 4


def test_invalid_ops_and_batched_views():
    from nr3d_lib_amd.models.grid_encodings.lotd.lotd_helpers import get_level_param_batched, level_param_index_shape
    m = _meta("mixed")
    kinds = [int(t) for t in m.level_types]
    for l, k in enumerate(kinds):
        if k == 0:                                           # Dense: no factor tables
            with pytest.raises(RuntimeError, match="Invalid op"):
                level_param_index_shape(m, l, "line")
        if k == 3:                                           # CP: no planes
            with pytest.raises(RuntimeError, match="Invalid op"):
                level_param_index_shape(m, l, "plane", 0)
    with pytest.raises(RuntimeError, match="Invalid op"):
        level_param_index_shape(m, kinds.index(1), "volume")
    p = torch.arange(3 * m.n_params, dtype=torch.float32).view(3, m.n_params)
    l = kinds.index(1)                                       # a VM level
    index, shape = level_param_index_shape(m, l, "vec", 1)
    v = get_level_param_batched(p, m, [0, 2], l, "vec", 1)
    assert tuple(v.shape) == (2, *shape)
    assert torch.equal(v[1], p[2, index[0]].view(shape))
    assert tuple(get_level_param_batched(p, m, 1, l).shape) == (m.level_sizes[l], m.level_n_feats[l])
    assert tuple(get_level_param_batched(p, m, slice(None), l).shape) == (3, m.level_sizes[l], m.level_n_feats[l])


def test_param_vertices_and_interpolate_match_reference():
    from nr3d_lib_amd.models.grid_encodings.lotd.lotd_helpers import param_interpolate, param_vertices
    z = np.load(os.path.join(GOLD, "ref_lotd_helpers.npz"))
    for tag, res, dim in (("v1", 5, 1), ("v2", 6, 2), ("v3", 4, 3), ("vc", [4, 6, 5], 3)):
        for forest in (False, True):
            got = param_vertices(res, dim, is_forest=forest).numpy()
            want = z[f"{tag}_{int(forest)}"]
            assert got.shape == want.shape and np.abs(got - want).max() <= 1e-6, (tag, forest)
    for d in (1, 2, 3):
        param, x = torch.from_numpy(z[f"i{d}_param"]), torch.from_numpy(z[f"i{d}_x"])
        for forest in (False, True):
            got = param_interpolate(param, x, 6, forest).numpy()
            want = z[f"i{d}_y{int(forest)}"]
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (d, forest, np.abs(got - want).max())
    # the sampler reproduces the table at its own vertices (both conventions)
    g = torch.Generator().manual_seed(0)
    t = torch.randn(1, 5, 5, 5, 2, generator=g)
    for forest in (False, True):
        v = param_vertices(5, 3, is_forest=forest).unsqueeze(0)
        torch.testing.assert_close(param_interpolate(t, v, 5, forest), t, rtol=1e-5, atol=1e-5)


def test_rescale_volume_matches_reference():
    """LoTDEncoding.rescale_volume against the reference's method run on the same tables (tests/golden/
    ref_rescale_volume.npz, make_golden_rescale.py): Dense volume, VM lines + planes, NPlaneMul / NPlaneSum planes, CP lines"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding
    from nr3d_lib_amd.models.spatial import AABBSpace
    z = np.load(os.path.join(GOLD, "ref_rescale_volume.npz"))
    cfg = dict(lod_res=z["res"].tolist(), lod_n_feats=z["feats"].tolist(), lod_types=[str(t) for t in z["types"]])
    enc = LoTDEncoding(3, lotd_cfg=cfg, space=AABBSpace(aabb=torch.from_numpy(z["old_aabb"])), dtype=torch.float)
    with torch.no_grad():
        enc.flattened_params.copy_(torch.from_numpy(z["before"]))
    enc.rescale_volume(torch.from_numpy(z["new_aabb"]))
    got, want = enc.flattened_params.detach().numpy(), z["after"]
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), np.abs(got - want).max()
    # per level, so that a small level cannot hide behind a large one
    m = enc.lod_meta
    for l in range(m.n_levels):
        a, b = m.level_offsets[l], m.level_offsets[l + 1]
        assert np.abs(got[a:b] - want[a:b]).max() <= 2e-5 * np.abs(want[a:b]).max(), l
    hashed = LoTDEncoding(3, lotd_cfg=dict(lod_res=[8, 64], lod_n_feats=[2, 2], lod_types=["Dense", "Hash"], hashmap_size=1024),
                          space=AABBSpace(aabb=torch.from_numpy(z["old_aabb"])), dtype=torch.float)
    with pytest.raises(RuntimeError, match="does not support spatial operations"):
        hashed.rescale_volume(torch.from_numpy(z["new_aabb"]))


def test_vertex_convention_matches_the_encoder(oracle):
    """The helpers and the encoder agree on where a table's vertices are: a Dense level whose vertex values are a linear
    function of the vertex positions (param_vertices) encodes exactly that function (trilinear interpolation reproduces
    it) -- evaluated with the CPU oracle the HIP kernels are tested against -- and still does after rescale_volume has
    re-gridded the table onto a smaller box."""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding, param_vertices
    from nr3d_lib_amd.models.spatial import AABBSpace
    R = 12
    old = torch.tensor([[-1.0, -2.0, -0.5], [1.0, 2.0, 1.5]])
    enc = LoTDEncoding(3, lotd_cfg=dict(lod_res=[R], lod_n_feats=[2], lod_types=["Dense"]), space=AABBSpace(aabb=old),
                       dtype=torch.float)
    A = torch.tensor([[0.3, -0.2, 0.5], [-0.7, 0.1, 0.25]])
    b = torch.tensor([0.1, -0.4])
    field = lambda world: world @ A.t() + b
    m_ref = oracle.lotd_create_meta(3, [R], [2], ["Dense"])

    def encode(world, aabb):
        c, h = (aabb[1] + aabb[0]) / 2, (aabb[1] - aabb[0]) / 2
        u = ((world - c) / h) / 2 + 0.5                        # what LoTDEncoding.forward feeds the kernel
        y, _ = oracle.lotd_fwd(m_ref, u.numpy().astype(np.float32), enc.flattened_params.detach().numpy())
        return torch.from_numpy(y)

    def world_of(v, aabb):
        return v * (aabb[1] - aabb[0]) / 2 + (aabb[1] + aabb[0]) / 2
    enc.set_level_param(0, 'vol', value=field(world_of(param_vertices(R, 3), old)))
    g = torch.Generator().manual_seed(1)
    new = torch.tensor([[-0.6, -1.1, 0.0], [0.7, 1.5, 1.2]])
    pts = world_of(torch.rand(500, 3, generator=g) * 2 - 1, new)           # inside the new (hence the old) box
    torch.testing.assert_close(encode(pts, old), field(pts), rtol=0, atol=2e-6)
    enc.rescale_volume(new)
    torch.testing.assert_close(enc.get_level_param(0, 'vol').detach(), field(world_of(param_vertices(R, 3), new)), rtol=0, atol=5e-6)
    torch.testing.assert_close(encode(pts, new), field(pts), rtol=0, atol=5e-6)
