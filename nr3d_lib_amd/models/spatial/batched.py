"""``BatchedBlockSpace``: the valid region of a BATCH of objects that share one axis-aligned box (each batch entry is its
own table set / occupancy grid; the box is the same for all of them).

Counterpart of the reference's nr3d_lib/models/spatial/batched.py (BatchedBlockSpace :20-146): coordinate and ray
normalisation into the box's [-1, 1] cube, uniform point sampling per batch entry, and the ray test that turns
[B, N_rays, 3] rays into the flat per-ray records (with batch indices) the batched ray marcher takes.
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from .forest import ray_box_intersection

__all__ = ['BatchedBlockSpace']


class BatchedBlockSpace(nn.Module):
    def __init__(self, bounding_size: float = None, aabb=None, dtype=torch.float, device=None) -> None:
        super().__init__()
        self.dtype = dtype
        if aabb is None:                       # `bounding_size` only matters without an explicit box
            half = (2.0 if bounding_size is None else bounding_size) / 2.
            aabb = [[-half] * 3, [half] * 3]
        aabb = aabb.to(dtype=dtype, device=device) if isinstance(aabb, torch.Tensor) else \
            torch.tensor(np.asarray(aabb, dtype=np.float64), dtype=dtype, device=device)
        self.register_buffer('aabb', aabb, persistent=True)

    device = property(lambda self: self.aabb.device)
    center = property(lambda self: (self.aabb[1] + self.aabb[0]) / 2.)
    radius3d = property(lambda self: (self.aabb[1] - self.aabb[0]) / 2.)
    radius3d_original = property(lambda self: self.radius3d)          # this space never shrinks

    def get_bounding_volume(self) -> torch.Tensor:
        return torch.cat([self.center, self.radius3d], dim=-1)

    def set_condition(self, *args, **kwargs):
        pass

    def clean_condition(self):
        pass

    # ---- operations on the current batch ------------------------------------------------------------------------
    def cur_batch__unnormalize_coords(self, coords: torch.Tensor, bidx: torch.LongTensor = None):
        return coords * self.radius3d + self.center

    def cur_batch__normalize_coords(self, world_coords: torch.Tensor, bidx: torch.LongTensor = None):
        return (world_coords - self.center) / self.radius3d

    def cur_batch__normalize_rays(self, rays_o: torch.Tensor, rays_d: torch.Tensor):
        """rays such that ``o + d * depth`` is in [-1, 1] coordinates at the original depths (|d| changes)"""
        return (rays_o - self.center) / self.radius3d, rays_d / self.radius3d

    def cur_batch__sample_pts_uniform(self, batch_size: int, num_pts_per_batch: int) -> Tuple[torch.Tensor, torch.LongTensor]:
        x = torch.empty([batch_size, num_pts_per_batch, 3], dtype=self.dtype, device=self.device).uniform_(-1, 1)
        bidx = torch.arange(batch_size, dtype=torch.long, device=self.device).unsqueeze(-1).expand(batch_size, num_pts_per_batch)
        return x, bidx

    def cur_batch__ray_test(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, return_rays=True,
                            normalized=False, compact_batch=False, **extra_ray_data):
        """rays_o / rays_d [B, N_rays, 3], near / far [B, N_rays] | float | None  ->  the rays that cross the box as flat
        records sorted by ray index: rays_inds, rays_bidx (index into the batch entries kept: all of them, or with
        ``compact_batch`` only those some ray hits -- ``full_bidx_map`` maps back), rays_full_bidx, near, far, the selected
        ``extra_ray_data`` and (``return_rays``) rays_o / rays_d."""
        assert rays_o.dim() == rays_d.dim() == 3
        if not normalized:
            rays_o, rays_d = self.cur_batch__normalize_rays(rays_o, rays_d)
        with torch.no_grad():
            B = rays_o.shape[0]
            t_in, t_out, _ = ray_box_intersection(rays_o, rays_d, aabb_min=-1., aabb_max=1.)
            if near is not None:
                t_in = torch.maximum(t_in, torch.as_tensor(near, dtype=t_in.dtype, device=t_in.device))
            if far is not None:
                t_out = torch.minimum(t_out, torch.as_tensor(far, dtype=t_out.dtype, device=t_out.device))
            hit = (t_out > t_in) & (t_out > (0 if near is None else near))
            if far is not None:
                hit = hit & (t_in < far)
        if compact_batch:
            full_bidx_map = hit.any(dim=-1).nonzero().long()[..., 0]
            ridx, bidx = hit[full_bidx_map].t().nonzero(as_tuple=True)
        else:
            full_bidx_map = torch.arange(B, device=rays_o.device, dtype=torch.long)
            ridx, bidx = hit.t().nonzero(as_tuple=True)         # ray-major: consecutive ray indices
        full_bidx = full_bidx_map[bidx]
        sel = (full_bidx, ridx)
        ret = dict(num_rays=ridx.numel(), rays_inds=ridx, rays_bidx=bidx, full_bidx_map=full_bidx_map,
                   rays_full_bidx=full_bidx, near=t_in[sel], far=t_out[sel])
        ret.update({k: v[sel] if isinstance(v, torch.Tensor) else v for k, v in extra_ray_data.items()})
        if return_rays:
            ret.update(rays_o=rays_o[sel], rays_d=rays_d[sel])
        return ret
