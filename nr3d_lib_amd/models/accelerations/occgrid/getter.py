"""``OccGridGetter``: an occupancy grid computed from a field in one go (no EMA state) -- a producer of the boolean grid
the ray marcher consumes, e.g. before rendering from a loaded checkpoint.

Counterpart of the reference's nr3d_lib/models/accelerations/occgrid/getter.py (OccGridGetter :18-156).  Plain PyTorch on
any device; the field query is the caller's function (typically the LoTD encoder + decoder).  Per step random points are
drawn in the voxels (``num_pts`` in total, at least one per voxel), converted to occupancy values and thresholded
(``binarize``); a voxel is occupied once any step found it so.  ``*_v2``: later steps only sample voxels that are still
empty.  The reference's v2 reduces the samples of a voxel with a max and scatters with torch_scatter.scatter_max; every
voxel index occurs once there, so a plain indexed store is the same operation.  Its ``occ_grid_from_net_batched_v1``
reshapes without the batch dimension (fails for B > 1); here the batch dimension is kept.
"""
from math import prod
from typing import List, Union

import numpy as np
import torch
import torch.nn as nn

from .ema_single import get_occ_val_fn
from .utils import binarize, resolution_tensor

__all__ = ['OccGridGetter']


class OccGridGetter(nn.Module):
    NUM_DIM: int = 3

    def __init__(self, resolution: Union[int, List[int], torch.Tensor] = 128, occ_val_fn_cfg=dict(type='density'),
                 occ_val_fn=None, occ_thre: float = 0.01, occ_thre_consider_mean=False, num_steps=4,
                 num_pts_per_batch: int = 2 ** 18, num_pts: int = None, dtype=torch.float, device=None) -> None:
        super().__init__()
        self.dtype = dtype
        resolution = resolution_tensor(resolution, self.NUM_DIM, device)
        self.register_buffer("resolution", resolution, persistent=False)
        axes = [torch.arange(r, device=device) for r in resolution.tolist()]
        self.register_buffer("gidx_full", torch.stack(torch.meshgrid(axes, indexing='ij'), dim=-1).view(-1, self.NUM_DIM),
                             persistent=False)
        self.occ_thre = occ_thre
        self.occ_val_fn = get_occ_val_fn(**occ_val_fn_cfg) if occ_val_fn is None else occ_val_fn
        self.occ_thre_consider_mean = occ_thre_consider_mean
        self.num_steps, self.num_pts, self.num_pts_per_batch = num_steps, num_pts, num_pts_per_batch

    device = property(lambda self: self.resolution.device)

    def _points(self, voxels: torch.Tensor, n_per_vox: int, lead=()):
        """uniform points in the given voxels, in the [-1, 1] coordinates of the grid: [*lead, V, n_per_vox, 3]"""
        jitter = torch.rand([*lead, voxels.shape[0], n_per_vox, self.NUM_DIM], device=self.device, dtype=self.dtype)
        return ((voxels.unsqueeze(-2) + jitter) / self.resolution) * 2 - 1

    def _occupied(self, occ_val: torch.Tensor) -> torch.Tensor:
        return binarize(occ_val, self.occ_thre, self.occ_thre_consider_mean)

    @torch.no_grad()
    def occ_grid_from_net(self, val_query_fn, progress=False) -> torch.Tensor:
        """every step samples all voxels"""
        res, V = self.resolution.tolist(), self.gidx_full.shape[0]
        grid = torch.zeros(res, dtype=torch.bool, device=self.device)
        for _ in range(self.num_steps):
            n = int(self.num_pts // V) + 1
            val = val_query_fn(self._points(self.gidx_full, n))
            grid |= self._occupied(self.occ_val_fn(val.flatten()).view(V, n)).any(dim=-1).view(res)
        return grid

    @torch.no_grad()
    def occ_grid_from_net_v2(self, val_query_fn, progress=False, verbose=False) -> torch.Tensor:
        """every step samples the voxels that are still empty"""
        res = self.resolution.tolist()
        grid = torch.zeros(res, dtype=torch.bool, device=self.device)
        for _ in range(self.num_steps):
            empty = (~grid).nonzero().long()
            if empty.shape[0] == 0:
                break
            n = int(self.num_pts // empty.shape[0]) + 1
            val = val_query_fn(self._points(empty, n))
            best = self.occ_val_fn(val.flatten()).view(empty.shape[0], n).max(-1).values
            full = torch.zeros(prod(res), dtype=best.dtype, device=self.device)
            full[(empty * empty.new_tensor([res[1] * res[2], res[2], 1])).sum(-1)] = best
            grid |= self._occupied(full).view(res)
        return grid

    @torch.no_grad()
    def occ_grid_from_net_batched_v1(self, B: int, val_query_fn_batched, progress=False) -> torch.Tensor:
        """a batch of fields queried together, ``val_query_fn_batched([B, V, n, 3]) -> [B, V, n]``"""
        res, V = self.resolution.tolist(), self.gidx_full.shape[0]
        grid = torch.zeros([B, *res], dtype=torch.bool, device=self.device)
        for _ in range(self.num_steps):
            n = int(self.num_pts_per_batch // V) + 1
            val = val_query_fn_batched(self._points(self.gidx_full, n, lead=(B,)))
            grid |= self._occupied(self.occ_val_fn(val.flatten()).view(B, V, n)).any(dim=-1).view(grid.shape)
        return grid

    @torch.no_grad()
    def occ_grid_from_net_batched_v2(self, B: int, val_query_fn_normalized_x_bi, progress=False) -> torch.Tensor:
        """batched, still-empty voxels only: ``val_query_fn_normalized_x_bi(pts [V', n, 3], bidx=[V', n]) -> [V', n]``"""
        res = self.resolution.tolist()
        grid = torch.zeros([B, *res], dtype=torch.bool, device=self.device)
        for _ in range(self.num_steps):
            empty = (~grid).nonzero().long()
            if empty.shape[0] == 0:
                break
            bidx, vox = empty[..., 0], empty[..., 1:]
            n = int(self.num_pts_per_batch // empty.shape[0]) + 1
            val = val_query_fn_normalized_x_bi(self._points(vox, n), bidx=bidx.view(-1, 1).expand(-1, n))
            best = self.occ_val_fn(val.flatten()).view(empty.shape[0], n).max(-1).values
            full = torch.zeros(prod(grid.shape), dtype=best.dtype, device=self.device)
            full[(empty * empty.new_tensor([prod(res), res[1] * res[2], res[2], 1])).sum(-1)] = best
            grid |= self._occupied(full).view(grid.shape)
        return grid
