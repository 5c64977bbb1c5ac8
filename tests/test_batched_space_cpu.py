"""CPU: BatchedBlockSpace (ray test -> flat per-ray records with batch indices) and the stateless batched accelerator's
grid construction / point sampling / occupancy query (the marching itself needs the GPU: tests/test_occ_grid_gpu.py)."""
import torch


def test_ray_test_records():
    from nr3d_lib_amd.models.spatial import BatchedBlockSpace
    sp = BatchedBlockSpace(aabb=[[-1, -2, -1], [1, 2, 3]])
    assert sp.get_bounding_volume().tolist() == [0, 0, 1, 1, 2, 2] and sp.radius3d_original.tolist() == [1, 2, 2]
    g = torch.Generator().manual_seed(0)
    B, N = 3, 40
    o = torch.tensor([0.0, 0.0, -5.0]).expand(B, N, 3).clone()
    d = torch.randn(B, N, 3, generator=g) * 0.3 + torch.tensor([0, 0, 1.0])
    d[2] = torch.tensor([1.0, 0.0, 0.0])                          # batch entry 2: every ray misses the box
    tag = torch.arange(B * N).view(B, N)
    r = sp.cur_batch__ray_test(o, d, tag=tag)
    on, dn = sp.cur_batch__normalize_rays(o, d)
    # brute force: slab test per ray
    tmin, tmax = (-1 - on) / dn, (1 - on) / dn
    t_in, t_out = torch.minimum(tmin, tmax).amax(-1), torch.maximum(tmin, tmax).amin(-1)
    hit = (t_out > t_in) & (t_out > 0)
    assert r["num_rays"] == int(hit.sum()) and not bool(hit[2].any())
    assert bool((r["rays_inds"][1:] >= r["rays_inds"][:-1]).all()), "ray-major order"
    assert torch.equal(hit[r["rays_full_bidx"], r["rays_inds"]], torch.ones(r["num_rays"], dtype=torch.bool))
    assert torch.equal(r["tag"], tag[r["rays_full_bidx"], r["rays_inds"]])
    torch.testing.assert_close(r["near"], t_in[r["rays_full_bidx"], r["rays_inds"]])
    torch.testing.assert_close(r["rays_o"], on[r["rays_full_bidx"], r["rays_inds"]])
    # compact: only the batch entries some ray hits are kept and renumbered
    c = sp.cur_batch__ray_test(o, d, near=0.5, far=7.0, compact_batch=True, return_rays=False)
    assert c["full_bidx_map"].tolist() == [0, 1] and "rays_o" not in c
    assert torch.equal(c["full_bidx_map"][c["rays_bidx"]], c["rays_full_bidx"])
    assert float(c["near"].min()) >= 0.5 and float(c["far"].max()) <= 7.0
    x, bi = sp.cur_batch__sample_pts_uniform(B, 7)
    assert tuple(x.shape) == (B, 7, 3) and bi[:, 0].tolist() == [0, 1, 2] and float(x.abs().max()) <= 1
    w = sp.cur_batch__unnormalize_coords(x)
    torch.testing.assert_close(sp.cur_batch__normalize_coords(w), x)


def test_getter_accel_grids_sampling_and_query():
    from nr3d_lib_amd.models.accelerations.occgrid_accel import OccGridAccelBatched_Getter
    from nr3d_lib_amd.models.spatial import BatchedBlockSpace
    torch.manual_seed(0)
    acc = OccGridAccelBatched_Getter(BatchedBlockSpace(), resolution=8, occ_thre=0.5, num_steps=2, num_pts_per_batch=4096)
    radii = torch.tensor([0.5, 0.8])
    acc.set_condition(2, val_query_fn_normalized_x_bi=lambda x, bidx: (x.norm(dim=-1) < radii[bidx]).float())
    g = acc.occ_grid_per_batch
    assert tuple(g.shape) == (2, 8, 8, 8) and 0 < int(g[0].sum()) < int(g[1].sum()) < 512
    pts, bi = acc.cur_batch__sample_pts_in_occupied(300)
    assert tuple(pts.shape) == (300, 3) and set(bi.tolist()) == {0, 1}
    assert bool(acc.cur_batch__query_occupancy(pts, bi).all())
    corner = torch.full((2, 3), 0.99)
    assert not bool(acc.cur_batch__query_occupancy(corner, torch.tensor([0, 1])).any())
    acc.clean_condition()
    try:
        acc.cur_batch__query_occupancy(corner, torch.tensor([0, 1]))
        raise SystemExit("expected an assertion")
    except AssertionError as e:
        assert "set_condition" in str(e)


def test_get_space_and_dense_octree_helpers():
    import pytest
    from nr3d_lib_amd.models.spatial import (AABBSpace, BatchedBlockSpace, ForestBlockSpace, create_dense_grid,
                                            create_octree_dense, create_octree_root_only, get_space)
    assert isinstance(get_space("aabb"), AABBSpace) and isinstance(get_space(dict(type="Batched", bounding_size=3.0)), BatchedBlockSpace)
    assert isinstance(get_space("forest"), ForestBlockSpace) and get_space("none") is None and get_space(None) is None
    assert get_space(dict(type="batched", bounding_size=3.0)).radius3d.tolist() == [1.5, 1.5, 1.5]
    with pytest.raises(RuntimeError, match="Invalid space_type"):
        get_space("sphere")
    with pytest.raises(NotImplementedError):
        get_space("aabb_dynamic")
    g = create_dense_grid(2)
    assert tuple(g.shape) == (64, 3) and g.dtype == torch.int16 and g[-1].tolist() == [3, 3, 3] and g[1].tolist() == [0, 0, 1]
    # a full octree: every node has all eight children -- 1 + 8 + 64 bytes of 255 for three levels
    o = create_octree_dense(3)
    assert o.dtype == torch.uint8 and o.tolist() == [255] * 73
    assert create_octree_root_only().tolist() == [255]
