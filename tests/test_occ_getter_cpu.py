"""CPU: OccGridGetter (occupancy grid from a field in one go; getter.py of the reference) on an analytic field: voxels
entirely inside a ball must come out occupied, voxels entirely outside empty, whatever the random samples were."""
import pytest
import torch


def _expect(res, radius, centers=None):
    """(must_be_occupied, must_be_empty) masks: the voxel's nearest / farthest point to the centre vs the radius"""
    ax = [torch.arange(r) for r in res]
    g = torch.stack(torch.meshgrid(ax, indexing="ij"), -1).float()
    lo = g / torch.tensor(res) * 2 - 1
    hi = (g + 1) / torch.tensor(res) * 2 - 1
    c = torch.zeros(3) if centers is None else centers
    near = torch.maximum(torch.maximum(lo - c, c - hi), torch.zeros(())).norm(dim=-1)
    far = torch.maximum((lo - c).abs(), (hi - c).abs()).norm(dim=-1)
    return far < radius, near > radius


@pytest.mark.parametrize("method", ["occ_grid_from_net", "occ_grid_from_net_v2"])
def test_single_field(method):
    from nr3d_lib_amd.models.accelerations.occgrid import OccGridGetter
    torch.manual_seed(0)
    res = [12, 10, 8]
    g = OccGridGetter(res, occ_val_fn_cfg=dict(type="density"), occ_thre=0.5, num_steps=3, num_pts=4 * 960)
    ball = lambda x: (x.norm(dim=-1) < 0.6).float()
    grid = getattr(g, method)(ball)
    inside, outside = _expect(res, 0.6)
    assert grid.dtype == torch.bool and list(grid.shape) == res
    assert bool(grid[inside].all()) and not bool(grid[outside].any()) and int(inside.sum()) > 20 and int(outside.sum()) > 100
    # an SDF field through the logistic conversion: occupied near the surface only
    g2 = OccGridGetter(res, occ_val_fn_cfg=dict(type="sdf", inv_s=64.0), occ_thre=0.1, num_steps=2, num_pts=20 * 960)
    shell = getattr(g2, method)(lambda x: x.norm(dim=-1) - 0.6)
    assert bool(shell.any()) and not bool(shell[0, 0, 0]) and not bool(shell[6, 5, 4])       # far outside / deep inside


def test_batched_fields():
    from nr3d_lib_amd.models.accelerations.occgrid import OccGridGetter
    torch.manual_seed(1)
    res, B = [8, 8, 8], 3
    radii = torch.tensor([0.3, 0.6, 0.9])
    g = OccGridGetter(res, occ_thre=0.5, num_steps=3, num_pts_per_batch=6 * 512)
    v1 = g.occ_grid_from_net_batched_v1(B, lambda x: (x.norm(dim=-1) < radii.view(B, 1, 1)).float())
    v2 = g.occ_grid_from_net_batched_v2(B, lambda x, bidx: (x.norm(dim=-1) < radii[bidx]).float())
    for grid in (v1, v2):
        assert list(grid.shape) == [B, *res]
        for b in range(B):
            inside, outside = _expect(res, float(radii[b]))
            assert bool(grid[b][inside].all()) and not bool(grid[b][outside].any())
    with pytest.raises(RuntimeError, match="Invalid type of resolution"):
        OccGridGetter(resolution=1.5)
