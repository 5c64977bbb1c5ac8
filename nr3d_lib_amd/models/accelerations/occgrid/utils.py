"""Occupancy-grid maintenance: the producer of the marcher's ``grid_binary`` input.

Counterpart of nr3d_lib/models/accelerations/occgrid/utils.py:17-125 (same names, arguments, in-place semantics).
``update_*_`` = EMA-decayed running maximum of an occupancy value per voxel, over the voxels that received samples;
the reference builds it on ``torch_scatter.scatter_max(..., out=ema_decay * grid)`` -- here two HIP kernels
(``nr3d_occ_scatter_max`` / ``nr3d_occ_apply_max``) with an optional ``all_reduce(MAX)`` in between, so ranks that hold
different shards of the samples end up with the same grid.  ``binarize`` / ``sample_pts_in_voxels`` are small tensor
expressions and stay in torch."""
import ctypes as C
from typing import Optional, Tuple

import torch

from nr3d_lib_amd import _hip as H

__all__ = ['err_msg_empty_occ', 'sample_pts_in_voxels', 'binarize', 'update_occ_val_grid_idx_', 'update_occ_val_grid_',
           'update_batched_occ_val_grid_idx_', 'update_batched_occ_val_grid_']

err_msg_empty_occ = ("Occupancy grid becomes empty during training. Your model/algorithm/training settings might be "
                     "incorrect. Please check configs and tensorboard.")



def resolution_tensor(resolution, num_dim: int, device=None) -> torch.Tensor:
    """int32 [num_dim] voxel counts from what the accelerator constructors accept: one int (a cube), a sequence / array of
    ``num_dim`` ints, or a tensor"""
    if torch.is_tensor(resolution):
        res = resolution.to(dtype=torch.int32, device=device).reshape(-1)
    else:
        try:
            vals = [int(v) for v in resolution]
        except TypeError:
            if not isinstance(resolution, (int,)) or isinstance(resolution, bool):
                raise RuntimeError(f"Invalid type of resolution={type(resolution)}")
            vals = [int(resolution)] * num_dim
        res = torch.tensor(vals, dtype=torch.int32, device=device)
    if res.numel() != num_dim:
        raise RuntimeError(f"resolution: expected {num_dim} entries, got {res.numel()}")
    return res

def sample_pts_in_voxels(gidx: torch.Tensor, num_pts: int, resolution: torch.Tensor, device=None,
                         dtype=torch.float) -> Tuple[torch.Tensor, torch.Tensor]:
    """uniform points in [-1, 1] inside the voxels ``gidx`` [N, num_dim]; returns (pts, voxel index of each point).
    Few points per voxel: draw voxels with replacement; otherwise the same count in every voxel (utils.py:17-38)."""
    assert gidx.dim() == 2, "Only support gidx with shape [N,num_dim]"
    device = device or gidx.device
    n_vox, n_dim = gidx.shape
    res = resolution.float()
    if num_pts / n_vox < 2.0:
        vidx = torch.randint(n_vox, [num_pts], device=device)
        jitter = torch.rand([num_pts, n_dim], device=device, dtype=dtype)
        pts = (gidx[vidx] + jitter) / res * 2 - 1
    else:
        per = int(num_pts // n_vox) + 1
        jitter = torch.rand([n_vox, per, n_dim], device=device, dtype=dtype)
        pts = ((gidx[:, None, :] + jitter) / res).view(-1, n_dim) * 2 - 1
        vidx = torch.arange(n_vox, device=device, dtype=torch.long).repeat_interleave(per)
    return pts, vidx


def binarize(occ_val: torch.Tensor, occ_threshold: float, consider_mean=False, eps=1e-5) -> torch.Tensor:
    """occupied = value above the threshold (or above min(threshold, mean - eps)), utils.py:75-78"""
    thr = (occ_val.mean() - eps).clamp_max_(occ_threshold) if consider_mean else occ_threshold
    return occ_val > thr


def _scatter_max_apply(grid: torch.Tensor, n_batches: int, res, *, gidx=None, pts=None, bidx=None, per_batch=0,
                       occ_val=None, ema_decay=1.0, group=None):
    H.require_gpu(grid, occ_val)
    if grid.dtype != torch.float32 or not grid.is_contiguous():
        raise RuntimeError("occ_val_grid: expected a contiguous float32 tensor (updated in place)")
    val = occ_val.detach().flatten().to(torch.float32).contiguous()
    n = val.numel()
    if gidx is not None:
        gidx = gidx.reshape(-1, 3).to(torch.int64).contiguous()
        if gidx.shape[0] != n:
            raise RuntimeError(f"gidx: expected {n} rows, got {gidx.shape[0]}")
    else:
        pts = pts.detach().reshape(-1, 3).to(torch.float32).contiguous()
        if pts.shape[0] != n:
            raise RuntimeError(f"pts: expected {n} rows, got {pts.shape[0]}")
    if bidx is not None:
        bidx = bidx.flatten().to(torch.int64).contiguous()
    cres = (C.c_int32 * 3)(*[int(r) for r in res])
    with torch.cuda.device(grid.device):
        st = H.stream_of(grid)
        vmax = torch.empty(grid.numel(), dtype=torch.float32, device=grid.device)
        H.check(H.lib().nr3d_occ_scatter_max(C.c_uint64(n), H.ptr(gidx), H.ptr(pts), H.ptr(bidx), C.c_uint64(per_batch),
                                             H.ptr(val), cres, H.u32(n_batches), H.ptr(vmax), st))
        if group is not None and group is not False and torch.distributed.is_initialized():
            pg = None if group is True else group
            if torch.distributed.get_world_size(pg) > 1:      # ranks hold different samples of the same update
                torch.distributed.all_reduce(vmax, op=torch.distributed.ReduceOp.MAX, group=pg)
        H.check(H.lib().nr3d_occ_apply_max(C.c_uint64(grid.numel()), H.f32(ema_decay), H.ptr(vmax), H.ptr(grid), st))


def update_occ_val_grid_idx_(occ_val_grid: torch.Tensor, gidx: torch.Tensor, occ_val: torch.Tensor,
                             ema_decay: float = 1.0, group=None):
    """in place: grid[v] = max(ema_decay * grid[v], max of occ_val over the samples with voxel index v) (utils.py:84-93).
    ``group``: a process group (or True for the default one) whose ranks hold different samples of the same update."""
    _scatter_max_apply(occ_val_grid, 1, occ_val_grid.shape, gidx=gidx, occ_val=occ_val, ema_decay=ema_decay, group=group)


def update_occ_val_grid_(occ_val_grid: torch.Tensor, pts: torch.Tensor, occ_val: torch.Tensor, ema_decay: float = 1.0,
                         group=None):
    """same with sample positions in [-1, 1]^3 (utils.py:95-100)"""
    _scatter_max_apply(occ_val_grid, 1, occ_val_grid.shape, pts=pts, occ_val=occ_val, ema_decay=ema_decay, group=group)


def update_batched_occ_val_grid_idx_(occ_val_grid: torch.Tensor, bidx: Optional[torch.Tensor] = None, gidx: torch.Tensor = ...,
                                     occ_val: torch.Tensor = ..., ema_decay: float = 1.0, group=None):
    """grid [B, Rx, Ry, Rz]; either per-sample ``bidx`` or batched ``occ_val`` / ``gidx`` [B, num_pts(, 3)]
    (utils.py:103-118)"""
    B = occ_val_grid.shape[0]
    per = 0 if bidx is not None else occ_val.flatten(1, -1).shape[1]
    _scatter_max_apply(occ_val_grid, B, occ_val_grid.shape[1:], gidx=gidx, bidx=bidx, per_batch=per, occ_val=occ_val,
                       ema_decay=ema_decay, group=group)


def update_batched_occ_val_grid_(occ_val_grid: torch.Tensor, pts: torch.Tensor, bidx: Optional[torch.Tensor] = None,
                                 occ_val: torch.Tensor = ..., ema_decay: float = 1.0, group=None):
    """utils.py:120-125"""
    B = occ_val_grid.shape[0]
    per = 0 if bidx is not None else occ_val.flatten(1, -1).shape[1]
    _scatter_max_apply(occ_val_grid, B, occ_val_grid.shape[1:], pts=pts, bidx=bidx, per_batch=per, occ_val=occ_val,
                       ema_decay=ema_decay, group=group)
