"""Occupancy-grid acceleration for a batch of objects in one shared box (``BatchedBlockSpace``): per batch entry one
grid, marched by the batched HIP ray marcher (``occgrid_raymarch_batched`` -> nr3d_batched_ray_marching_*).

Counterpart of the reference's nr3d_lib/models/accelerations/occgrid_accel/batched.py: the common per-batch operations
(``OccGridAccelBatched_Base`` :31-83), the persistent EMA grids selected per batch by instance index
(``OccGridAccelBatched_Ema`` :85-170) and the stateless variant that recomputes the grids from the conditioned network
whenever a batch is set (``OccGridAccelBatched_Getter`` :234-293).  The debug visualisations are not provided.
"""
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from nr3d_lib_amd.graphics.raymarch.occgrid_raymarch import RaymarchRetBatched, occgrid_raymarch_batched
from nr3d_lib_amd.models.accelerations.occgrid import OccGridEmaBatched, OccGridGetter, err_msg_empty_occ, sample_pts_in_voxels
from nr3d_lib_amd.models.spatial.batched import BatchedBlockSpace

__all__ = ['OccGridAccelBatched_Base', 'OccGridAccelBatched_Ema', 'OccGridAccelBatched_Getter']


class OccGridAccelBatched_Base(nn.Module):
    """operations on the grids of the CURRENT batch (``occ_grid_per_batch`` [B, Rx, Ry, Rz], set by set_condition)"""
    occ_grid_per_batch: torch.Tensor = None

    def _grids(self, what: str) -> torch.Tensor:
        assert self.occ_grid_per_batch is not None, f"Please call set_condition() first before {what}()"
        return self.occ_grid_per_batch

    @torch.no_grad()
    def cur_batch__sample_pts_in_occupied(self, num_pts: int) -> Tuple[torch.Tensor, torch.LongTensor]:
        """uniform points in the occupied voxels of the current batch -> (pts in [-1, 1], local batch index)"""
        occupied = self._grids("sample_pts_in_occupied").nonzero().long()
        assert occupied.numel() > 0, err_msg_empty_occ
        pts, vidx = sample_pts_in_voxels(occupied[..., 1:], num_pts, resolution=self.resolution, dtype=self.dtype)
        return pts, occupied[..., 0][vidx]

    @torch.no_grad()
    def cur_batch__query_occupancy(self, pts: torch.Tensor, bidx: torch.LongTensor) -> torch.BoolTensor:
        """pts in [-1, 1] with their local batch index -> occupied?"""
        grids = self._grids("query_occupancy")
        res = self.resolution
        gidx = ((pts / 2. + 0.5) * res).long().clamp(res.new_tensor([0]), res - 1)
        return grids[(bidx,) + tuple(gidx.movedim(-1, 0))]

    def cur_batch__ray_march(self, rays_o: torch.Tensor, rays_d: torch.Tensor, rays_bidx: torch.Tensor = None, *, near=None,
                             far=None, perturb=False, step_size: float = 1e-3, max_step_size: float = 1e10,
                             dt_gamma: float = 0.0, max_steps: int = 512) -> RaymarchRetBatched:
        return occgrid_raymarch_batched(self._grids("ray_march"), rays_o, rays_d, rays_bidx, near, far, perturb=perturb,
                                        step_size=step_size, max_step_size=max_step_size, dt_gamma=dt_gamma,
                                        max_steps=max_steps)


class OccGridAccelBatched_Ema(OccGridAccelBatched_Base):
    """one persistent EMA grid per instance; a batch selects its instances' grids"""

    def __init__(self, space: BatchedBlockSpace, num_batches: int, resolution: Union[int, List[int], torch.Tensor] = None,
                 dtype=torch.float, device=None, **occ_kwargs) -> None:
        super().__init__()
        self.training_granularity = 0.0
        assert isinstance(space, BatchedBlockSpace), f"{self.__class__.__name__} expects space of BatchedBlockSpace"
        self.space, self.dtype, self.num_batches = space, dtype, num_batches
        self.occ = OccGridEmaBatched(**occ_kwargs, num_batches=num_batches, resolution=resolution, dtype=dtype, device=device)
        self.clean_condition()

    device = property(lambda self: self.space.device)
    NUM_DIM = property(lambda self: self.occ.NUM_DIM)
    resolution = property(lambda self: self.occ.resolution)

    def get_occ_grid(self):
        return self.occ.occ_grid

    def set_condition(self, batch_size: int, *, ins_inds_per_batch: torch.LongTensor = None, val_query_fn_normalized_x_bi=None):
        assert ins_inds_per_batch is not None, f"`ins_inds_per_batch` is required for {type(self)}"
        self.batch_size, self.ins_inds_per_batch = batch_size, ins_inds_per_batch
        self.occ_grid_per_batch = self.get_occ_grid()[ins_inds_per_batch].contiguous()

    def clean_condition(self):
        self.batch_size = self.ins_inds_per_batch = self.occ_grid_per_batch = None

    @torch.no_grad()
    def init(self, val_query_fn_normalized_x_bi, logger=None):
        assert self.batch_size is not None and self.batch_size == self.num_batches, \
            "Before init(), should set_condition() on all batches (i.e. all latents)."
        self.occ.init(val_query_fn_normalized_x_bi, logger=logger)      # bidx = instance index here

    @torch.no_grad()
    def cur_batch__step(self, cur_it: int, val_query_fn_normalized_x_bi, logger=None):
        assert self.ins_inds_per_batch is not None, "Please call set_condition() first before step()"
        self.occ.step(cur_it, val_query_fn_normalized_x_bi, within_bi=self.ins_inds_per_batch, logger=logger)

    @torch.no_grad()
    def cur_batch__collect_samples(self, pts: torch.Tensor, bidx: torch.LongTensor, val: torch.Tensor, normalized=True):
        """forward-hook style: the field's samples of this batch (local ``bidx``) feed the instances' grids"""
        if self.training:
            assert self.ins_inds_per_batch is not None, "Please call set_condition() first before collect_samples()"
            if not normalized:
                pts = self.space.cur_batch__normalize_coords(pts, bidx)
            self.occ.collect_samples(pts, self.ins_inds_per_batch[bidx], val)

    @torch.no_grad()
    def debug_stats(self):
        grid = self.get_occ_grid()
        return dict(frac_occupied=float(grid.float().mean()), num_occupied=int(grid.sum()))


class OccGridAccelBatched_Getter(OccGridAccelBatched_Base):
    """no state: the grids of a batch are computed from the conditioned network when the batch is set"""

    def __init__(self, space: BatchedBlockSpace, resolution: Union[int, List[int], torch.Tensor] = None, dtype=torch.float,
                 device=None, **occ_kwargs) -> None:
        super().__init__()
        self.training_granularity = 0.0
        assert isinstance(space, BatchedBlockSpace), f"{self.__class__.__name__} expects space of BatchedBlockSpace"
        self.space, self.dtype = space, dtype
        self.occ_getter = OccGridGetter(**occ_kwargs, resolution=resolution, dtype=dtype, device=device)
        self.clean_condition()

    device = property(lambda self: self.space.device)
    resolution = property(lambda self: self.occ_getter.resolution)     # (`reslution` in the reference: a typo there)

    def set_condition(self, batch_size: int, *, ins_inds_per_batch: torch.LongTensor = None, val_query_fn_normalized_x_bi=None):
        assert val_query_fn_normalized_x_bi is not None, f"`val_query_fn_normalized_x_bi` is required for {type(self)}"
        self.batch_size = batch_size
        self.occ_grid_per_batch = self.occ_getter.occ_grid_from_net_batched_v2(batch_size, val_query_fn_normalized_x_bi)

    def clean_condition(self):
        self.batch_size = self.occ_grid_per_batch = None

    def init(self, val_query_fn_normalized_x_bi, logger=None):
        pass

    def cur_batch__step(self, cur_it: int, val_query_fn_normalized_x_bi, logger=None):
        pass

    def cur_batch__collect_samples(self, pts, bidx, val, normalized=True):
        pass

    def debug_stats(self):
        return {}
