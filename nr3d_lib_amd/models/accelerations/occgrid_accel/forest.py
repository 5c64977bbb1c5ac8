"""Occupancy-grid acceleration of a forest of blocks -- counterpart of ``OccGridAccelForest``
(nr3d_lib/models/accelerations/occgrid_accel/forest.py:36-223): one grid per block (an ``OccGridEmaBatched`` created once
the forest is populated), world-space rays marched through the blocks they cross."""
from typing import Dict, List, Tuple, Union

import torch
import torch.nn as nn

from nr3d_lib_amd.graphics.raymarch.occgrid_raymarch import occgrid_raymarch_forest
from nr3d_lib_amd.models.accelerations.occgrid import OccGridEmaBatched
from nr3d_lib_amd.models.spatial import ForestBlockSpace

__all__ = ['OccGridAccelForest']


class OccGridAccelForest(nn.Module):
    def __init__(self, space: ForestBlockSpace, resolution: Union[int, List[int], torch.Tensor] = None, vox_size: float = None,
                 dtype=torch.float, device=None, **occ_kwargs) -> None:
        super().__init__()
        assert isinstance(space, ForestBlockSpace), f"{self.__class__.__name__} expects space of ForestBlockSpace"
        assert (resolution is not None) != (vox_size is not None), "Please specify `vox_size` or `resolution` for OccGridAccel."
        self.space, self.dtype, self.occ = space, dtype, None
        if resolution is None:
            resolution = (self.space.world_block_size / vox_size).long()
        occ_kwargs.update(resolution=resolution)
        self.occ_kwargs = occ_kwargs
        self.training_granularity = 0.0

    device = property(lambda self: self.space.device)
    NUM_DIM = property(lambda self: self.occ.NUM_DIM)
    resolution = property(lambda self: self.occ.resolution)

    def populate(self):
        """after the space is populated: one grid per block"""
        self.occ = OccGridEmaBatched(num_batches=self.space.n_trees, **self.occ_kwargs, dtype=self.dtype, device=self.device)

    @torch.no_grad()
    def init(self, query_fn_block_x_blidx, logger=None):
        """``query_fn_block_x_blidx(block_x, bidx=blidx)``: field values at points normalised to their block's [-1,1]^3"""
        return self.occ.init(query_fn_block_x_blidx, logger=logger)

    @torch.no_grad()
    def step(self, cur_it: int, query_fn_block_x_blidx, logger=None):
        return self.occ.step(cur_it, query_fn_block_x_blidx, logger=logger)

    @torch.no_grad()
    def collect_samples(self, pts: torch.Tensor, blidx: torch.Tensor = None, val: torch.Tensor = ..., normalized=True):
        if self.training:
            if not normalized:
                pts, blidx = self.space.normalize_coords(pts, blidx)
            valid = (blidx >= 0).nonzero(as_tuple=True)
            self.occ.collect_samples(pts[valid], blidx[valid], val[valid])

    @torch.no_grad()
    def sample_pts_in_occupied(self, num_pts: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.occ.sample_pts_in_occupied(num_pts)

    @torch.no_grad()
    def query_occupancy(self, pts: torch.Tensor, blidx: torch.Tensor) -> torch.Tensor:
        return self.occ.query(pts, blidx)

    def get_occ_grid(self):
        return self.occ.occ_grid

    def ray_march(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, seg_block_inds: torch.Tensor = ...,
                  seg_entries: torch.Tensor = ..., seg_exits: torch.Tensor = ..., seg_pack_infos: torch.Tensor = ..., *,
                  perturb=False, step_size: float = 1e-3, max_step_size: float = 1e10, dt_gamma: float = 0.0, max_steps: int = 512):
        """world-space rays + their block segments (``space.ray_test``) -> RaymarchRetForest"""
        return occgrid_raymarch_forest(self.space.meta, self.get_occ_grid(), rays_o, rays_d, near, far, seg_block_inds,
                                       seg_entries, seg_exits, seg_pack_infos, perturb=perturb, step_size=step_size,
                                       max_step_size=max_step_size, dt_gamma=dt_gamma, max_steps=max_steps)

    def ray_march_simple_step_segment(self, rays_o, rays_d, near=None, far=None, seg_block_inds: torch.Tensor = ...,
                                      seg_entries: torch.Tensor = ..., seg_exits: torch.Tensor = ...,
                                      seg_pack_infos: torch.Tensor = ..., *, perturb=False, step_mode: str = 'depth', **step_kwargs):
        return self.space.ray_step_coarse(rays_o, rays_d, near, far, seg_block_inds, seg_entries, seg_exits, seg_pack_infos,
                                          step_mode=step_mode, perturb=perturb, **step_kwargs)

    def query_world(self, pts: torch.Tensor):
        """occupancy at world positions (False outside every block)"""
        x, blidx = self.space.normalize_coords(pts)
        valid = (blidx >= 0).nonzero(as_tuple=True)
        occupied = torch.zeros(blidx.shape, dtype=torch.bool, device=pts.device)
        if valid[0].numel() > 0:
            occupied[valid] = self.occ.query(x[valid], blidx[valid])
        return occupied

    @torch.no_grad()
    def debug_stats(self) -> Dict[str, float]:
        g = self.get_occ_grid()
        n = g.sum().item()
        return {'num_occupied': n, 'frac_occupied': n / g.numel()}
