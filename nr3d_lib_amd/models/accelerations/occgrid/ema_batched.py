"""A stack of occupancy grids, one per batch entry (object / block), maintained like ``OccGridEma`` -- counterpart of
``OccGridEmaBatched`` (nr3d_lib/models/accelerations/occgrid/ema_batched.py:17-309).  The field is queried as
``val_query_fn(pts, bidx=bidx)`` with points normalised to the entry's own [-1, 1]^3; sampling, the renderer's samples
and the update run over (entry, voxel) pairs; the scatter is the batched HIP pair (utils.update_batched_*)."""
from copy import deepcopy
from typing import List, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .ema_single import _voxel_index_table, get_occ_val_fn
from .utils import (binarize, resolution_tensor, sample_pts_in_voxels, update_batched_occ_val_grid_,
                    update_batched_occ_val_grid_idx_)

__all__ = ['OccGridEmaBatched']

_EMPTY = "Occupancy grid becomes empty during training. Your model/algorithm/training setting might be incorrect. Please check."


class OccGridEmaBatched(nn.Module):
    NUM_DIM: int = 3

    def __init__(self, num_batches: int, resolution: Union[int, List[int], torch.Tensor] = 128,
                 occ_val_fn_cfg=dict(type='density'), occ_val_fn=None, occ_thre: float = 0.01, occ_thre_consider_mean=False,
                 ema_decay: float = 0.95, n_steps_between_update: int = 16, n_steps_warmup: int = 256, init_cfg=dict(),
                 update_from_net_cfg=dict(), update_from_samples_cfg=dict(), dtype=torch.float, device=None, group=None) -> None:
        super().__init__()
        self.num_batches, self.dtype = num_batches, dtype
        resolution = resolution_tensor(resolution, self.NUM_DIM, device)
        shape = [num_batches, *resolution.tolist()]
        self.register_buffer('is_initialized', torch.tensor([False], dtype=torch.bool), persistent=True)
        self.register_buffer("resolution", resolution, persistent=False)
        self.register_buffer("occ_grid", torch.zeros(shape, dtype=torch.bool, device=device), persistent=True)
        self.register_buffer("occ_val_grid", torch.zeros(shape, dtype=dtype, device=device), persistent=True)
        self.register_buffer("gidx_full", _voxel_index_table(resolution), persistent=False)
        self._register_load_state_dict_pre_hook(self._before_load_state_dict)
        self.ema_decay, self.init_cfg = ema_decay, init_cfg
        self.update_from_net_cfg, self.update_from_samples_cfg = update_from_net_cfg, update_from_samples_cfg
        self.should_collect_samples: bool = update_from_samples_cfg is not None
        self.occ_thre, self.occ_thre_consider_mean = occ_thre, occ_thre_consider_mean
        self.occ_val_fn = get_occ_val_fn(**occ_val_fn_cfg) if occ_val_fn is None else occ_val_fn
        self.n_steps_between_update, self.n_steps_warmup = n_steps_between_update, n_steps_warmup
        self.group = group
        if self.should_collect_samples:
            self.register_buffer('_occ_val_grid_pcl', torch.zeros(shape, dtype=dtype, device=device), persistent=False)

    @property
    def device(self) -> torch.device:
        return self.resolution.device

    def _before_load_state_dict(self, state_dict, prefix, *unused):
        occ_grid = state_dict[prefix + 'occ_grid']
        if list(occ_grid.shape) != list(self.occ_grid.shape):
            self.occ_grid = torch.zeros_like(occ_grid, device=self.device)
            self.occ_val_grid = torch.zeros(occ_grid.shape, dtype=self.dtype, device=self.device)
            self.num_batches = occ_grid.shape[0]
            self.resolution = torch.tensor(list(occ_grid.shape[1:]), dtype=torch.int32, device=self.device)
            self.gidx_full = _voxel_index_table(self.resolution)
            if self.should_collect_samples:
                self._occ_val_grid_pcl = torch.zeros(occ_grid.shape, dtype=self.dtype, device=self.device)

    def _rebinarize(self):
        self.occ_grid = binarize(self.occ_val_grid, self.occ_thre, self.occ_thre_consider_mean)

    def _in(self, idx: torch.Tensor, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """n points inside the (entry, voxel) rows of idx [m, 4] -> (pts, bidx)"""
        pts, vidx = sample_pts_in_voxels(idx[:, 1:], n, self.resolution, dtype=self.dtype)
        return pts, idx[vidx, 0]

    # ---- initialisation --------------------------------------------------------------------------------------------------
    def init(self, val_query_fn_normalized_x_bi=None, logger=None) -> bool:
        if bool(self.is_initialized):
            return False
        cfg = deepcopy(self.init_cfg)
        mode = cfg.pop('mode')
        if mode == 'constant':
            self._init_from_constant(**cfg)
        elif mode in ('from_net', 'net'):
            self._init_from_net(val_query_fn_normalized_x_bi, **cfg)
        else:
            raise RuntimeError(f"Invalid init_mode={mode}")
        self.is_initialized.fill_(True)
        return True

    @torch.no_grad()
    def _init_from_constant(self, constant_value: float):
        self.occ_val_grid.fill_(constant_value)
        self._rebinarize()

    @torch.no_grad()
    def _init_from_net(self, val_query_fn_normalized_x_bi, *, num_steps=4, num_pts_per_batch: int = 2 ** 18, num_pts: int = None):
        num_pts = num_pts_per_batch * self.num_batches if num_pts is None else num_pts
        for _ in range(num_steps):
            idx_empty = self.occ_grid.logical_not().nonzero().long()
            if idx_empty.shape[0] > 0:
                pts, bidx = self._in(idx_empty, num_pts)
                val = val_query_fn_normalized_x_bi(pts, bidx=bidx)
                update_batched_occ_val_grid_(self.occ_val_grid, pts, bidx, self.occ_val_fn(val), ema_decay=1.0, group=self.group)
                self._rebinarize()

    # ---- per-iteration update --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, cur_it: int, val_query_fn_normalized_x_bi, within_bi: torch.Tensor = None, logger=None) -> bool:
        assert bool(self.is_initialized), f"{type(self)} should init() first before step(cur_it={cur_it})"
        if cur_it > 0 and cur_it % self.n_steps_between_update == 0:
            self._step(cur_it, val_query_fn_normalized_x_bi, within_bi=within_bi, **self.update_from_net_cfg)
            return True
        return False

    @torch.no_grad()
    def _step(self, cur_it: int, val_query_fn_normalized_x_bi, *, within_bi: torch.Tensor = None, num_steps=4,
              num_pts_per_batch: int = 2 ** 18, num_pts: int = None):
        """``within_bi``: restrict the update to these entries (indices are local to it while sampling)"""
        num_pts = num_pts_per_batch * self.num_batches if num_pts is None else num_pts
        n_entries = len(within_bi) if within_bi is not None else self.num_batches
        warm = cur_it < self.n_steps_warmup
        if not warm:
            occ = self.occ_grid[within_bi].contiguous() if within_bi is not None else self.occ_grid
            nonempty, empty = occ.nonzero().long(), occ.logical_not().nonzero().long()
            assert nonempty.numel() > 0, _EMPTY
        pts_all, bidx_all, val_all = [], [], []
        for _ in range(num_steps):
            n_uniform = num_pts if warm else num_pts // 2
            pts, _ = sample_pts_in_voxels(self.gidx_full, n_uniform, self.resolution, dtype=self.dtype)
            parts = [(pts, torch.randint(n_entries, size=(len(pts),), dtype=torch.long, device=self.device))]
            if not warm:
                if empty.numel() > 0:
                    parts.append(self._in(empty, num_pts // 4))
                parts.append(self._in(nonempty, num_pts // 4))
            pts = torch.cat([p for p, _ in parts], 0)
            bidx = torch.cat([b for _, b in parts], 0)
            pts_all.append(pts); bidx_all.append(bidx)
            val_all.append(val_query_fn_normalized_x_bi(pts, bidx=bidx))
        bidx = torch.cat(bidx_all, 0)
        if within_bi is not None:
            bidx = within_bi[bidx]
        self._step_update_occ(torch.cat(pts_all, 0), bidx, torch.cat(val_all, 0))

    @torch.no_grad()
    def _step_update_occ(self, pts: torch.Tensor, bidx: torch.Tensor = None, val: torch.Tensor = ...):
        res = self.resolution
        if bidx is None:                                  # batched layout [num_batches, n, 3]
            bidx = torch.arange(self.num_batches, device=self.device).view(-1, 1).expand(-1, pts.flatten(1, -2).shape[1]).reshape(-1)
        pts, bidx, occ_val = pts.reshape(-1, self.NUM_DIM), bidx.flatten(), self.occ_val_fn(val).flatten()
        gidx = ((pts / 2. + 0.5) * res).long().clamp(res.new_tensor([0]), res - 1)
        if self.should_collect_samples:
            idx_pcl = self._occ_val_grid_pcl.nonzero().long()
            if idx_pcl.numel() > 0:
                bidx = torch.cat([bidx, idx_pcl[:, 0]], 0)
                gidx = torch.cat([gidx, idx_pcl[:, 1:]], 0)
                occ_val = torch.cat([occ_val, self._occ_val_grid_pcl[tuple(idx_pcl.t())]], 0)
            self._occ_val_grid_pcl.zero_()
        update_batched_occ_val_grid_idx_(self.occ_val_grid, bidx, gidx, occ_val, ema_decay=self.ema_decay, group=self.group)
        self._rebinarize()

    # ---- samples handed over by the renderer -------------------------------------------------------------------------------
    @torch.no_grad()
    def collect_samples(self, pts: torch.Tensor, bidx: torch.Tensor = None, val: torch.Tensor = ...):
        if self.training and self.should_collect_samples:
            self._collect_samples(pts.flatten(0, -2), bidx.flatten(), val.flatten(), **self.update_from_samples_cfg)

    @torch.no_grad()
    def _collect_samples(self, pts: torch.Tensor, bidx: torch.Tensor = None, val: torch.Tensor = ...):
        update_batched_occ_val_grid_(self._occ_val_grid_pcl, pts, bidx, self.occ_val_fn(val), ema_decay=1.0)

    # ---- queries -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_pts_in_occupied(self, num_pts: int, within_bi: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (pts, entry index; local to ``within_bi`` when given)"""
        nonempty = (self.occ_grid[within_bi] if within_bi is not None else self.occ_grid).nonzero().long()
        assert nonempty.numel() > 0, _EMPTY
        return self._in(nonempty, num_pts)

    @torch.no_grad()
    def query(self, pts: torch.Tensor, bidx: torch.Tensor = None) -> torch.Tensor:
        """pts [N, 3] with bidx [N], or batched pts [num_batches, ..., 3]"""
        res = self.resolution
        gidx = ((pts / 2. + 0.5) * res).long().clamp(res.new_tensor([0]), res - 1)
        if bidx is None:
            bidx = torch.arange(self.num_batches, device=self.device).view(-1, *[1] * (pts.dim() - 2)).expand(pts.shape[:-1])
        return self.occ_grid[(bidx,) + tuple(gidx.movedim(-1, 0))]

    def extra_repr(self) -> str:
        return "occ_grid=[" + ','.join(str(s) for s in self.occ_grid.shape) + "]"
