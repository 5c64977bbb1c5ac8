"""Volume-rendering helpers on packed and batched rays -- counterpart of nr3d_lib/graphics/nerf/nerf_utils.py:23-160
(same names and argument meaning).  The packed variants run on the HIP pack_ops."""
import torch

from nr3d_lib_amd.graphics.pack_ops import (packed_alpha_to_vw, packed_cumprod, packed_cumsum,
                                            packed_volume_render_compression)
from nr3d_lib_amd.bindings import _pack_ops as _backend

__all__ = ['tau_to_alpha', 'sigma_delta_to_alpha', 'packed_alpha_to_vw', 'packed_alpha_to_vw_v1', 'packed_alpha_to_vw_v2',
           'packed_volume_render_compression', 'packed_tau_to_vw', 'packed_tau_alpha_to_vw', 'ray_alpha_to_vw',
           'ray_tau_to_vw', 'ray_tau_alpha_to_vw']


def tau_to_alpha(tau: torch.Tensor) -> torch.Tensor:
    """opacity of an interval with optical depth ``tau`` (nerf_utils.py:23-24)"""
    return 1 - torch.exp(-tau)


class _SigmaDeltaToAlpha(torch.autograd.Function):
    """alpha = 1 - exp(-sigma * delta) as one kernel each way (the ray-query drivers' `tau_to_alpha(sigma * deltas)`,
    nerf_ray_query.py:126,178: three element-wise launches forward and four backward in eager PyTorch)"""

    @staticmethod
    def forward(ctx, sigma, delta):
        ctx.save_for_backward(sigma, delta)
        return _backend.tau_to_alpha_forward(sigma, delta)

    @staticmethod
    def backward(ctx, grad_alpha):
        sigma, delta = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True: a higher-order graph is wanted (the reference expression supports double backward,
            # nerf_utils.py:23-24) -> the gradient as differentiable torch ops on the saved inputs; first-order training
            # never takes this branch
            return grad_alpha * delta * torch.exp(-sigma * delta), None
        return _backend.tau_to_alpha_backward(sigma, delta, grad_alpha.contiguous()), None


def sigma_delta_to_alpha(sigma: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
    """``tau_to_alpha(sigma * delta)``; fused when both are float32 GPU tensors of one shape and ``delta`` needs no gradient
    (interval lengths from the marcher), the plain expression otherwise"""
    if sigma.is_cuda and sigma.dtype == torch.float32 and delta.dtype == torch.float32 and sigma.shape == delta.shape \
            and not delta.requires_grad:
        s, d = sigma.contiguous(), delta.contiguous()
        if torch.is_grad_enabled() and s.requires_grad:
            return _SigmaDeltaToAlpha.apply(s, d)
        return _backend.tau_to_alpha_forward(s.detach(), d)
    return tau_to_alpha(sigma * delta)


def packed_alpha_to_vw_v1(alpha: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """w = alpha * exclusive-cumprod(1 + 1e-10 - alpha) per pack (nerf_utils.py:33-47)"""
    a = alpha.reshape(-1, 1)
    return (a * packed_cumprod((1 + 1e-10) - a, pack_infos, exclusive=True)).squeeze(-1)


def packed_alpha_to_vw_v2(alpha, pack_infos, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0):
    """the fused kernel with early stop and alpha threshold (nerf_utils.py:49-62)"""
    return packed_alpha_to_vw(alpha, pack_infos, early_stop_eps, alpha_thre)


def _exclusive_transmittance(tau, pack_infos):
    return torch.exp(-packed_cumsum(tau, pack_infos, exclusive=True))


def packed_tau_to_vw(tau: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    t = tau.reshape(-1, 1)
    return (tau_to_alpha(t) * _exclusive_transmittance(t, pack_infos)).squeeze(-1)


def packed_tau_alpha_to_vw(tau: torch.Tensor, alpha: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    return (alpha.reshape(-1, 1) * _exclusive_transmittance(tau.reshape(-1, 1), pack_infos)).squeeze(-1)


def _shift_right_with_one(v: torch.Tensor) -> torch.Tensor:
    return torch.cat([torch.ones_like(v[..., :1]), v[..., :-1]], dim=-1)


def ray_alpha_to_vw(alpha: torch.Tensor) -> torch.Tensor:
    """[..., num_pts] batched counterpart of packed_alpha_to_vw_v1 (nerf_utils.py:100-112)"""
    return alpha * torch.cumprod(_shift_right_with_one((1 + 1e-10) - alpha), dim=-1)


def ray_tau_to_vw(tau: torch.Tensor) -> torch.Tensor:
    return tau_to_alpha(tau) * torch.exp(-(torch.cumsum(tau, dim=-1) - tau))


def ray_tau_alpha_to_vw(tau: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return alpha * torch.exp(-(torch.cumsum(tau, dim=-1) - tau))
