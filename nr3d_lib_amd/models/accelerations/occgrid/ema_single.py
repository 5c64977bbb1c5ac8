"""Occupancy grid maintained as an exponential moving maximum of a field's values -- counterpart of ``OccGridEma``
(nr3d_lib/models/accelerations/occgrid/ema_single.py:17-268): the owner of the ``occ_grid`` the marcher walks.

Every ``n_steps_between_update`` training iterations ``step`` queries the field at fresh samples -- uniformly over all
voxels during warm-up, afterwards 1/2 uniform + 1/4 inside empty + 1/4 inside occupied voxels -- merges the samples the
renderer handed over through ``collect_samples`` since the last update, and applies
``grid <- max(ema_decay * grid, per-voxel max of the samples)`` on the touched voxels, then re-binarises.  The scatter
is the HIP pair ``nr3d_occ_scatter_max`` / ``nr3d_occ_apply_max`` (utils.py); with ``group=`` every rank samples and
queries 1/world of the points and the per-voxel maxima are all-reduced(MAX) before the decay is applied (the reference
has no multi-GPU path).  Coordinates are normalised to [-1, 1]^3."""
import functools
from copy import deepcopy
from typing import List, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import binarize, resolution_tensor, sample_pts_in_voxels, update_occ_val_grid_, update_occ_val_grid_idx_

__all__ = ['OccGridEma', 'get_occ_val_fn', 'sdf_to_occ_val', 'normalized_logistic_density']


def normalized_logistic_density(x: torch.Tensor, inv_s) -> torch.Tensor:
    """logistic density scaled to peak 1 (nr3d_lib/maths/common.py:122-133)"""
    return (1. / torch.cosh((inv_s * x / 2.).clamp_(-20, 20))) ** 2


def sdf_to_occ_val(sdf: torch.Tensor, *, inv_s: float = None, inv_s_anneal_cfg: dict = None):
    if inv_s_anneal_cfg is not None:                        # utils.py:63-68: the schedule's value at the configured iteration
        from nr3d_lib_amd.models.annealers import get_anneal_val
        inv_s = get_anneal_val(**inv_s_anneal_cfg)
    else:
        assert inv_s is not None, "Need config `inv_s`"
    return normalized_logistic_density(sdf, inv_s)


def get_occ_val_fn(type: str = 'sdf', **kwargs):
    """field value -> occupancy value (utils.py:70-80)"""
    if type == 'sdf':
        return functools.partial(sdf_to_occ_val, **kwargs)
    if type == 'raw_sdf':
        return lambda sdf: 1.0 - torch.abs(sdf)
    if type in ('occ', 'density'):
        return nn.Identity()
    raise RuntimeError(f"Invalid type={type}")


def _voxel_index_table(resolution: torch.Tensor) -> torch.Tensor:
    axes = [torch.arange(int(r), device=resolution.device) for r in resolution.tolist()]
    return torch.stack(torch.meshgrid(axes, indexing='ij'), dim=-1).view(-1, len(axes))


class OccGridEma(nn.Module):
    NUM_DIM: int = 3

    def __init__(self, resolution: Union[int, List[int], torch.Tensor] = 128, occ_val_fn_cfg=dict(type='density'),
                 occ_val_fn=None, occ_thre: float = 0.01, occ_thre_consider_mean=False, ema_decay: float = 0.95,
                 n_steps_between_update: int = 16, n_steps_warmup: int = 256, init_cfg=dict(), update_from_net_cfg=dict(),
                 update_from_samples_cfg=dict(), dtype=torch.float, device=None, group=None) -> None:
        super().__init__()
        self.dtype = dtype
        resolution = resolution_tensor(resolution, self.NUM_DIM, device)
        shape = resolution.tolist()
        self.register_buffer('is_initialized', torch.tensor([False], dtype=torch.bool), persistent=True)
        self.register_buffer("resolution", resolution, persistent=False)
        self.register_buffer("occ_grid", torch.zeros(shape, dtype=torch.bool, device=device), persistent=True)
        self.register_buffer("occ_val_grid", torch.zeros(shape, dtype=self.dtype, device=device), persistent=True)
        self.register_buffer("gidx_full", _voxel_index_table(resolution), persistent=False)
        self._register_load_state_dict_pre_hook(self._before_load_state_dict)
        self.ema_decay, self.init_cfg = ema_decay, init_cfg
        self.update_from_net_cfg, self.update_from_samples_cfg = update_from_net_cfg, update_from_samples_cfg
        self.should_collect_samples: bool = update_from_samples_cfg is not None
        self.occ_thre, self.occ_thre_consider_mean = occ_thre, occ_thre_consider_mean
        self.occ_val_fn = get_occ_val_fn(**occ_val_fn_cfg) if occ_val_fn is None else occ_val_fn
        self.n_steps_between_update, self.n_steps_warmup = n_steps_between_update, n_steps_warmup
        self.group = group                 # process group whose ranks share this grid (None: single process)
        if self.should_collect_samples:
            # per-voxel maxima of the samples seen by the renderer since the last update
            self.register_buffer('_occ_val_grid_pcl', torch.zeros(shape, dtype=self.dtype, device=device), persistent=False)

    @property
    def device(self) -> torch.device:
        return self.resolution.device

    def _before_load_state_dict(self, state_dict, prefix, *unused):
        if prefix + 'val_grid' in state_dict:                       # older checkpoints
            state_dict[prefix + 'occ_val_grid'] = state_dict.pop(prefix + 'val_grid')
        occ_grid = state_dict[prefix + 'occ_grid']
        if list(occ_grid.shape) != list(self.occ_grid.shape):       # the checkpoint decides the resolution
            self.occ_grid = torch.zeros_like(occ_grid, device=self.device)
            self.occ_val_grid = torch.zeros(occ_grid.shape, dtype=self.dtype, device=self.device)
            self.resolution = torch.tensor(list(occ_grid.shape), dtype=torch.int32, device=self.device)
            self.gidx_full = _voxel_index_table(self.resolution)
            if self.should_collect_samples:
                self._occ_val_grid_pcl = torch.zeros(occ_grid.shape, dtype=self.dtype, device=self.device)

    def _rebinarize(self):
        self.occ_grid = binarize(self.occ_val_grid, self.occ_thre, self.occ_thre_consider_mean)

    def _sample(self, gidx, n):
        return sample_pts_in_voxels(gidx, n, resolution=self.resolution, dtype=self.dtype)[0]

    def _share(self, n: int) -> int:
        """this rank's part of an n-point sampling budget"""
        if self.group is None or not torch.distributed.is_initialized():
            return n
        pg = None if self.group is True else self.group
        return -(-n // torch.distributed.get_world_size(pg))

    # ---- initialisation --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def init(self, val_query_fn=None, logger=None) -> bool:
        if bool(self.is_initialized):
            return False
        cfg = deepcopy(self.init_cfg)
        mode = cfg.pop('mode')
        if mode == 'constant':
            self._init_from_constant(**cfg)
        elif mode == 'from_net':
            self._init_from_net(val_query_fn, **cfg)
        else:
            raise RuntimeError(f"Invalid init_mode={mode}")
        self.is_initialized.fill_(True)
        return True

    @torch.no_grad()
    def _init_from_constant(self, constant_value: float):
        self.occ_val_grid.fill_(constant_value)
        self._rebinarize()

    @torch.no_grad()
    def _init_from_net(self, val_query_fn, *, num_steps=4, num_pts: int = 2 ** 18):
        for _ in range(num_steps):
            gidx_empty = self.occ_grid.logical_not().nonzero().long()       # first round: every voxel
            if gidx_empty.shape[0] > 0:
                pts = self._sample(gidx_empty, self._share(num_pts))
                update_occ_val_grid_(self.occ_val_grid, pts, self.occ_val_fn(val_query_fn(pts)), ema_decay=1.0, group=self.group)
                self._rebinarize()

    # ---- per-iteration update --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, cur_it: int, val_query_fn, logger=None) -> bool:
        assert bool(self.is_initialized), f"{type(self)} should init() first before step(cur_it={cur_it})"
        if cur_it > 0 and cur_it % self.n_steps_between_update == 0:
            self._step(cur_it, val_query_fn, **self.update_from_net_cfg)
            return True
        return False

    @torch.no_grad()
    def _step(self, cur_it: int, val_query_fn, *, num_steps=4, num_pts: int = 2 ** 18):
        pts_all, val_all = [], []
        if cur_it < self.n_steps_warmup:
            budget = [(self.gidx_full, self._share(num_pts))]
        else:
            nonempty = self.occ_grid.nonzero().long()
            empty = self.occ_grid.logical_not().nonzero().long()
            assert nonempty.numel() > 0, "Occupancy grid becomes empty during training. Your model/algorithm/training " \
                                         "settings might be incorrect. Please check configs and tensorboard."
            budget = [(self.gidx_full, self._share(num_pts // 2))]
            if empty.numel() > 0:
                budget.append((empty, self._share(num_pts // 4)))
            budget.append((nonempty, self._share(num_pts // 4)))
        for _ in range(num_steps):
            pts = torch.cat([self._sample(g, n) for g, n in budget], dim=0)
            pts_all.append(pts)
            val_all.append(val_query_fn(pts))
        self._step_update_occ(torch.cat(pts_all, 0), torch.cat(val_all, 0))

    @torch.no_grad()
    def _step_update_occ(self, pts: torch.Tensor, val: torch.Tensor):
        pts, occ_val = pts.flatten(0, -2), self.occ_val_fn(val.flatten())
        res = self.resolution
        gidx = ((pts / 2. + 0.5) * res).long().clamp(res.new_tensor([0]), res - 1)
        if self.should_collect_samples:
            idx_pcl = self._occ_val_grid_pcl.nonzero().long()
            if idx_pcl.numel() > 0:
                gidx = torch.cat([gidx, idx_pcl], dim=0)
                occ_val = torch.cat([occ_val, self._occ_val_grid_pcl[tuple(idx_pcl.t())]], dim=0)
            self._occ_val_grid_pcl.zero_()
        update_occ_val_grid_idx_(self.occ_val_grid, gidx, occ_val, ema_decay=self.ema_decay, group=self.group)
        self._rebinarize()

    # ---- samples handed over by the renderer -------------------------------------------------------------------------------
    @torch.no_grad()
    def collect_samples(self, pts: torch.Tensor, val: torch.Tensor = None):
        """to be called like a forward hook with the points a render pass evaluated and the field's values there"""
        if self.training and self.should_collect_samples:
            self._collect_samples(pts, val, **self.update_from_samples_cfg)

    @torch.no_grad()
    def _collect_samples(self, pts: torch.Tensor, val: torch.Tensor):
        update_occ_val_grid_(self._occ_val_grid_pcl, pts, self.occ_val_fn(val), ema_decay=1.0)

    # ---- queries -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_pts_in_occupied(self, num_pts: int) -> torch.Tensor:
        nonempty = self.occ_grid.nonzero().long()
        assert nonempty.numel() > 0, "Occupancy grid becomes empty during training. Your model/algorithm/training " \
                                     "settings might be incorrect. Please check configs and tensorboard."
        return self._sample(nonempty, num_pts)

    @torch.no_grad()
    def query(self, pts: torch.Tensor) -> torch.Tensor:
        res = self.resolution
        gidx = ((pts / 2. + 0.5) * res).long().clamp(res.new_tensor([0]), res - 1)
        return self.occ_grid[tuple(gidx.movedim(-1, 0))]

    # ---- volume shrinking -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def try_shrink(self, old_aabb: torch.Tensor) -> torch.Tensor:
        """the tight box (one voxel of margin) around the occupied voxels, in the coordinates of ``old_aabb`` [2, 3]"""
        origin, scale = (old_aabb[1] + old_aabb[0]) / 2., (old_aabb[1] - old_aabb[0]) / 2.
        idx = self.occ_grid.nonzero()
        lo, hi = idx.min(dim=0).values, idx.max(dim=0).values
        box = torch.stack([lo - 1, hi + 1], 0).clamp_(hi.new_tensor([0]), self.resolution - 1)
        return ((box / self.resolution) * 2 - 1) * scale + origin

    @torch.no_grad()
    def rescale_volume(self, old_aabb: torch.Tensor, new_aabb: torch.Tensor):
        """resample the value grid from the old box onto the voxel corners of the new one (trilinear), re-binarise"""
        new_aabb = new_aabb.view(2, self.NUM_DIM)
        origin, scale = (old_aabb[1] + old_aabb[0]) / 2., (old_aabb[1] - old_aabb[0]) / 2.
        new_origin, new_scale = (new_aabb[1] + new_aabb[0]) / 2., (new_aabb[1] - new_aabb[0]) / 2.
        v = (self.gidx_full / self.resolution) * 2. - 1.                 # the new grid's vertices, normalised to the new box
        v = ((v * new_scale + new_origin) - origin) / scale             # ... expressed in the old box
        shape = self.resolution.tolist()
        self.occ_val_grid = F.grid_sample(
            self.occ_val_grid.view([1, 1, *shape]), v.view(1, *shape, self.NUM_DIM).flip(-1),    # grid_sample wants (z, y, x)
            align_corners=True, padding_mode='zeros').squeeze(0).squeeze(0).contiguous()
        self._rebinarize()

    def extra_repr(self) -> str:
        return "occ_grid=[" + ','.join(str(s) for s in self.occ_grid.shape) + "]"
