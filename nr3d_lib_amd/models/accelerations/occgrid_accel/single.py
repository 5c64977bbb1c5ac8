"""Occupancy-grid acceleration of one AABB -- counterpart of ``OccGridAccel``
(nr3d_lib/models/accelerations/occgrid_accel/single.py:36-170): the object a field exposes as ``model.accel``, i.e. what
the ray-query drivers call (``ray_march``) and what the trainer ticks (``init`` / ``step`` / ``collect_samples``)."""
from typing import Dict, List, Union

import torch
import torch.nn as nn

from nr3d_lib_amd.graphics.raymarch.occgrid_raymarch import occgrid_raymarch
from nr3d_lib_amd.models.accelerations.occgrid import OccGridEma
from nr3d_lib_amd.models.spatial import AABBSpace

__all__ = ['OccGridAccel']


class OccGridAccel(nn.Module):
    def __init__(self, space: AABBSpace, resolution: Union[int, List[int], torch.Tensor] = None, vox_size: float = None,
                 dtype=torch.float, device=None, **occ_kwargs) -> None:
        super().__init__()
        assert isinstance(space, AABBSpace), f"{self.__class__.__name__} expects space of AABBSpace"
        assert (resolution is not None) != (vox_size is not None), "Please specify `vox_size` or `resolution` for OccGridAccel."
        self.space, self.dtype = space, dtype
        if resolution is None:
            resolution = (self.space.radius3d * 2 / vox_size).long()
        self.occ = OccGridEma(resolution=resolution, **occ_kwargs, dtype=dtype, device=self.device)
        self.training_granularity = 0.0

    device = property(lambda self: self.space.device)
    NUM_DIM = property(lambda self: self.occ.NUM_DIM)
    resolution = property(lambda self: self.occ.resolution)

    def get_occ_grid(self):
        return self.occ.occ_grid

    @torch.no_grad()
    def init(self, val_query_fn, logger=None):
        return self.occ.init(val_query_fn, logger=logger)

    @torch.no_grad()
    def step(self, cur_it: int, val_query_fn, logger=None):
        return self.occ.step(cur_it, val_query_fn, logger=logger)

    @torch.no_grad()
    def collect_samples(self, pts: torch.Tensor, val: torch.Tensor, normalized=True):
        if self.training:
            self.occ.collect_samples(pts if normalized else self.space.normalize_coords(pts), val)

    @torch.no_grad()
    def sample_pts_in_occupied(self, num_pts: int) -> torch.Tensor:
        return self.occ.sample_pts_in_occupied(num_pts)

    @torch.no_grad()
    def query_occupancy(self, pts: torch.Tensor) -> torch.Tensor:
        return self.occ.query(pts)

    def ray_march(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, *, perturb=False, normalized=True,
                  step_size: float = 1e-3, max_step_size: float = 1e10, dt_gamma: float = 0.0, max_steps: int = 512):
        if not normalized:
            rays_o, rays_d = self.space.normalize_rays(rays_o, rays_d)
        return occgrid_raymarch(self.get_occ_grid(), rays_o, rays_d, near, far, perturb=perturb, step_size=step_size,
                                max_step_size=max_step_size, dt_gamma=dt_gamma, max_steps=max_steps)

    @torch.no_grad()
    def rescale_volume(self, new_aabb: torch.Tensor):
        self.occ.rescale_volume(self.space.aabb.clone(), new_aabb)

    @torch.no_grad()
    def try_shrink(self) -> torch.Tensor:
        return self.occ.try_shrink(self.space.aabb.clone())

    @torch.no_grad()
    def num_occupied(self) -> int:
        return int(self.get_occ_grid().sum().item())

    @torch.no_grad()
    def frac_occupied(self) -> float:
        g = self.get_occ_grid()
        return g.sum().item() / g.numel()

    @torch.no_grad()
    def debug_stats(self) -> Dict[str, float]:
        return {'num_occupied': self.num_occupied(), 'frac_occupied': self.frac_occupied()}
