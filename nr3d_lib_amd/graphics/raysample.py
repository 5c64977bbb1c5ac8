"""Depth samplers on rays -- counterpart of the parts of nr3d_lib/graphics/raysample.py the hot-path drivers use:
``packed_sample_cdf`` (:38-62, on the HIP ``packed_invert_cdf``) and the three ``batch_sample_step_*`` ladders
(:285-383).  Same names, arguments and return conventions."""
from typing import Tuple

import torch

from nr3d_lib_amd.graphics.pack_ops import packed_invert_cdf

__all__ = ['packed_sample_cdf', 'batch_sample_step_linear', 'batch_sample_step_wrt_depth',
           'batch_sample_step_wrt_sqrt_depth']


def _per_ray(near, far, prefix_shape):
    if prefix_shape is None:
        prefix_shape = [1] if list(near.shape) == [1] else near.squeeze().shape
    shp = tuple(prefix_shape)
    return near.squeeze().expand(shp).unsqueeze(-1), far.squeeze().expand(shp).unsqueeze(-1), shp


def _steps(num_samples, shp, perturb, like, first=0):
    """{first, first+1, ...} (+ U(0,1) per sample when perturbing)"""
    idx = torch.arange(first, first + num_samples, device=like.device)
    if perturb:
        idx = idx + torch.rand((*shp, num_samples), dtype=like.dtype, device=like.device)
    return idx


def _with_deltas(t, last):
    """interval lengths: differences, the last one given by `last` (a tensor) or repeated (None)"""
    d = torch.zeros_like(t)
    d[..., :-1] = torch.diff(t, dim=-1)
    d[..., -1] = d[..., -2] if last is None else last
    return d


def batch_sample_step_linear(near, far, num_samples: int, prefix_shape=None, perturb=False, return_dt=False):
    """uniform in depth; perturb: one uniform sample per stratum (raysample.py:285-311)"""
    near, far, shp = _per_ray(near, far, prefix_shape)
    dt = (far - near) / (num_samples if perturb else num_samples - 1)
    t = torch.addcmul(near, _steps(num_samples, shp, perturb, near).to(near.dtype), dt)
    if not return_dt:
        return t
    return t, (_with_deltas(t, (far[..., 0] - near[..., 0]) / num_samples) if perturb else dt.expand((*shp, num_samples)))


@torch.no_grad()
def _geometric(near, far, num_samples, shp, perturb):
    n = num_samples if perturb else num_samples - 1
    logk = torch.log(far / near) / n                          # far / near must be finite: callers clamp it
    first = -num_samples if perturb else 1 - num_samples
    return far * torch.exp(logk * _steps(num_samples, logk.shape[:-1], perturb, near, first=first))


@torch.no_grad()
def batch_sample_step_wrt_depth(near, far, num_samples: int, prefix_shape=None, perturb=False, return_dt=False):
    """step proportional to depth (geometric ladder with ratio clamped to num_samples, raysample.py:342-362)"""
    near, far, shp = _per_ray(near, far, prefix_shape)
    ratio = (far / near).clamp_max(num_samples)
    near_ = far / ratio
    t = _geometric(near_, far, num_samples, shp, perturb)
    t = (t - near_) * ((far - near) / (far - near_)) + near
    return (t, _with_deltas(t, None)) if return_dt else t


@torch.no_grad()
def batch_sample_step_wrt_sqrt_depth(near, far, num_samples: int, prefix_shape=None, perturb=False, return_dt=False):
    """step proportional to sqrt(depth): t = (k i + c)^2 / 4 (raysample.py:364-383)"""
    near, far, shp = _per_ray(near, far, prefix_shape)
    c = (4 * near).sqrt()
    k = ((4 * far).sqrt() - c) / (num_samples if perturb else num_samples - 1)
    t = 0.25 * (k * _steps(num_samples, shp, perturb, near) + c).square()
    return (t, _with_deltas(t, None)) if return_dt else t


def packed_sample_cdf(bins: torch.Tensor, cdfs: torch.Tensor, pack_infos: torch.Tensor, num_to_sample: int,
                      perturb=False) -> Tuple[torch.Tensor, torch.Tensor]:
    """inverse-CDF sampling per pack: bins / cdfs packed [num_pts] (cdf with a leading zero per pack) ->
    (t_samples, global bin index) both [num_packs, num_to_sample] (raysample.py:38-62)"""
    n_packs = pack_infos.shape[0]
    if perturb:
        u = batch_sample_step_linear(bins.new_zeros(n_packs), bins.new_ones(n_packs), num_to_sample, perturb=True)
    else:
        u = torch.linspace(0., 1., num_to_sample + 2, device=bins.device, dtype=bins.dtype)[1:-1].expand(n_packs, num_to_sample)
    return packed_invert_cdf(bins, cdfs.to(bins.dtype), u.contiguous(), pack_infos)
