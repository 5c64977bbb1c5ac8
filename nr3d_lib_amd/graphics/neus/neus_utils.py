"""NeuS opacity from SDF samples -- counterpart of nr3d_lib/graphics/neus/neus_utils.py:52-111,164-189 (the functions the
upsampling driver uses).  Alpha of the interval between two consecutive samples = relative drop of the logistic CDF
``sigmoid(sdf * inv_s)``; packed variants take the difference inside each pack (last sample: appended value or 0)."""
import torch

from nr3d_lib_amd.graphics.pack_ops import packed_diff

__all__ = ['neus_cdf', 'neus_ray_cdf_to_alpha', 'neus_ray_sdf_to_alpha', 'neus_packed_cdf_to_alpha',
           'neus_packed_sdf_to_alpha', 'neus_packed_sdf_to_upsample_alpha']


def neus_cdf(x, inv_s):
    return torch.sigmoid(x * inv_s)


def neus_ray_cdf_to_alpha(cdf: torch.Tensor, append_cdf_1=False):
    """[..., n] cdf -> [..., n-1] alpha (or [..., n] with a virtual last cdf of 1)"""
    if append_cdf_1:
        drop, ref = -cdf.diff(append=cdf.new_ones((*cdf.shape[:-1], 1))), cdf
    else:
        drop, ref = -cdf.diff(), cdf[..., :-1]
    return (drop / (ref + 1e-5)).clamp_min(0)


def neus_ray_sdf_to_alpha(sdf: torch.Tensor, inv_s, append_cdf_1=False):
    return neus_ray_cdf_to_alpha(neus_cdf(sdf, inv_s), append_cdf_1=append_cdf_1)


def neus_packed_cdf_to_alpha(cdf: torch.Tensor, pack_infos: torch.Tensor, append_cdf_1=False,
                             pack_cdf_appends: torch.Tensor = None):
    if append_cdf_1:
        pack_cdf_appends = cdf.new_ones(pack_infos.shape[0])
    drop = -packed_diff(cdf, pack_infos, pack_appends=pack_cdf_appends)
    return (drop / (cdf + 1e-5)).clamp_min(0)


def neus_packed_sdf_to_alpha(sdf: torch.Tensor, inv_s, pack_infos: torch.Tensor, append_cdf_1=False,
                             pack_sdf_appends: torch.Tensor = None):
    appends = None if pack_sdf_appends is None else neus_cdf(pack_sdf_appends, inv_s)
    return neus_packed_cdf_to_alpha(neus_cdf(sdf, inv_s), pack_infos, append_cdf_1=append_cdf_1, pack_cdf_appends=appends)


@torch.no_grad()
def neus_packed_sdf_to_upsample_alpha(sdf: torch.Tensor, depth_samples: torch.Tensor, inv_s, pack_infos: torch.Tensor):
    """the NeuS paper's up-sampling opacity: sdf at both interval ends re-estimated from the mid-point value and the
    smaller (more negative) of the current / previous slopes, clamped to [-10, 0] (neus_utils.py:164-189)"""
    d_sdf = packed_diff(sdf, pack_infos)                      # trailing zero in every pack
    deltas = packed_diff(depth_samples, pack_infos)
    slope = d_sdf / (deltas + 1e-5)
    prev = slope.roll(1).index_fill_(0, pack_infos[:, 0], 0)  # previous interval's slope, 0 at pack starts
    slope = torch.minimum(prev, slope).clamp_(-10, 0)
    mid = (sdf + d_sdf * 0.5).to(depth_samples.dtype)
    half = slope * deltas * 0.5
    cdf_prev, cdf_next = torch.sigmoid((mid - half) * inv_s), torch.sigmoid((mid + half) * inv_s)
    return ((cdf_prev - cdf_next) / (cdf_prev + 1e-5)).clamp_min_(0)
