"""Single axis-aligned block -- counterpart of ``AABBSpace`` (nr3d_lib/models/spatial/aabb.py:21-140): the box a field
lives in, with the world <-> [-1, 1]^3 maps the encoder, the occupancy grid and the marcher agree on."""
import numpy as np
import torch
import torch.nn as nn

__all__ = ['AABBSpace']


class AABBSpace(nn.Module):
    def __init__(self, bounding_size: float = None, aabb=None, dtype=torch.float, device=None) -> None:
        super().__init__()
        self.dtype = dtype
        if aabb is None:                       # `bounding_size` only matters without an explicit box
            hbs = (2.0 if bounding_size is None else bounding_size) / 2.
            aabb = [[-hbs] * 3, [hbs] * 3]
        aabb = aabb.to(dtype=dtype, device=device) if isinstance(aabb, torch.Tensor) else \
            torch.tensor(np.asarray(aabb, dtype=np.float64), dtype=dtype, device=device)
        self.register_buffer('aabb', aabb.view(2, 3), persistent=True)
        # the coordinate scale at construction: survives rescale_volume, keeps `sdf` / `nablas` comparable across shrinks
        self.register_buffer('radius3d_original', self.radius3d.data.clone(), persistent=True)

    device = property(lambda self: self.aabb.device)
    center = property(lambda self: (self.aabb[1] + self.aabb[0]) / 2.)
    radius3d = property(lambda self: (self.aabb[1] - self.aabb[0]) / 2.)

    def get_bounding_volume(self) -> torch.Tensor:
        return torch.cat([self.center, self.radius3d], dim=-1)

    def unnormalize_coords(self, coords: torch.Tensor):
        return coords * self.radius3d + self.center

    def normalize_coords(self, world_coords: torch.Tensor):
        return (world_coords - self.center) / self.radius3d

    def normalize_rays(self, rays_o: torch.Tensor, rays_d: torch.Tensor):
        """such that new_o + new_d * depth is in [-1, 1] for the same depth (the norm of rays_d changes)"""
        return (rays_o - self.center) / self.radius3d, rays_d / self.radius3d

    def sample_pts_uniform(self, num_pts: int):
        return torch.empty([num_pts, 3], dtype=self.dtype, device=self.device).uniform_(-1, 1)

    def contains(self, pts: torch.Tensor):
        return torch.logical_and(pts >= self.aabb[0], pts < self.aabb[1]).all(-1)

    def ray_test(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, return_rays=True, normalized=False,
                 **extra_ray_data):
        """slab test against the box -> dict(num_rays, rays_inds, near, far[, rays_o, rays_d] + the extras of the hit rays)
        (aabb.py:85-100); the returned rays are the normalised ones"""
        if not normalized:
            rays_o, rays_d = self.normalize_rays(rays_o, rays_d)
        with torch.no_grad():
            t0, t1 = (-1. - rays_o) / rays_d, (1. - rays_o) / rays_d
            near_ = torch.minimum(t0, t1).max(dim=-1).values
            far_ = torch.maximum(t0, t1).min(dim=-1).values
            if near is not None:
                near_ = torch.maximum(near_, torch.as_tensor(near, dtype=near_.dtype, device=near_.device))
            if far is not None:
                far_ = torch.minimum(far_, torch.as_tensor(far, dtype=far_.dtype, device=far_.device))
            mask = (far_ > near_) & (far_ > (0 if near is None else near))
            if far is not None:
                mask = mask & (near_ < far)
            ridx = mask.nonzero().long()[..., 0]
        ret = dict(num_rays=ridx.shape[0], rays_inds=ridx, near=near_[ridx], far=far_[ridx])
        ret.update({k: v[ridx] if isinstance(v, torch.Tensor) else v for k, v in extra_ray_data.items()})
        if return_rays:
            ret.update(rays_o=rays_o[ridx], rays_d=rays_d[ridx])
        return ret

    def rescale_volume(self, new_aabb: torch.Tensor, reset_scale=False):
        self.aabb = new_aabb.to(self.aabb).view(2, 3)

    def extra_repr(self) -> str:
        return f"aabb={self.aabb}"
