"""Pack-wise ops on ragged ("packed") tensors with autograd -- counterpart of the reference's
nr3d_lib/graphics/pack_ops/pack_ops.py (747 lines): same public names / signatures / gradients.

A packed tensor is [num_feats(, feat_dim)] plus ``pack_infos`` int64 [num_packs, 2] = (first index,
length) per pack.  Kernels: nr3d_lib_amd.bindings._pack_ops (HIP, one wave per pack).
"""
from numbers import Number
from typing import Tuple, Union

import torch
from torch.autograd.function import once_differentiable

import nr3d_lib_amd.bindings._pack_ops as _backend
from nr3d_lib_amd._hip import mark_ordered

__all__ = [
    'packed_sort_inplace', 'packed_sort', 'packed_searchsorted', 'packed_searchsorted_packed_vals',
    'packed_sum', 'packed_mean', 'packed_cumprod', 'packed_cumsum', 'packed_diff', 'packed_backward_diff',
    'packed_invert_cdf', 'packed_alpha_to_vw', 'packed_volume_render_compression',
    'packed_volume_render_compression_gather', 'packed_composite',
    'packed_add', 'packed_sub', 'packed_mul', 'packed_div', 'packed_matmul',
    'packed_gt', 'packed_geq', 'packed_lt', 'packed_leq', 'packed_eq', 'packed_neq',
    'interleave_arange_simple', 'interleave_arange', 'interleave_linstep', 'interleave_linspace',
    'interleave_sample_step_wrt_depth_clamped', 'interleave_sample_step_wrt_depth_in_packed_segments',
    'merge_two_packs_sorted_aligned', 'merge_two_packs_sorted_a_includes_b', 'merge_two_packs_sorted',
    'merge_two_batch_a_includes_b', 'merge_two_batch',
    'get_pack_infos_from_boundary', 'get_pack_infos_from_first', 'get_pack_infos_from_n', 'get_pack_infos_from_batch',
    'octree_mark_consecutive_segments', 'mark_pack_boundaries', 'expand_pack_boundary', 'torch_intersect1d_unique',
]


# ------------------------------------------------------------------------------------------------
# pack_infos helpers (pure torch)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def get_pack_infos_from_first(first_inds: torch.Tensor, numel: int):
    nxt = torch.cat([first_inds[1:], first_inds.new_tensor([numel])])
    return torch.stack([first_inds, nxt - first_inds], 1)


@torch.no_grad()
def get_pack_infos_from_boundary(boundary: torch.Tensor):
    # nonzero() is ascending: ordered, disjoint packs by construction (the tag of _hip.mark_ordered)
    return mark_ordered(get_pack_infos_from_first(boundary.nonzero().long()[..., 0], boundary.numel()))


@torch.no_grad()
def get_pack_infos_from_n(n_per_pack: torch.Tensor):
    return mark_ordered(torch.stack([n_per_pack.cumsum(0) - n_per_pack, n_per_pack], 1))


@torch.no_grad()
def get_pack_infos_from_batch(n_batches: int, batch_data_size: int, device=None):
    first = torch.arange(n_batches, device=device, dtype=torch.long) * batch_data_size
    return mark_ordered(torch.stack([first, torch.full_like(first, batch_data_size)], 1))


@torch.no_grad()
def expand_pack_boundary(pack_boundary: torch.Tensor, num_samples: int):
    out = torch.zeros(pack_boundary.shape[0] * num_samples, device=pack_boundary.device, dtype=torch.bool)
    out[pack_boundary.nonzero().long() * num_samples] = 1
    return out


def mark_pack_boundaries(pack_ids):
    return _backend.mark_pack_boundaries_cuda(pack_ids.contiguous()).bool()


@torch.no_grad()
def octree_mark_consecutive_segments(pidx, pack_infos, point_hierarchies):
    return _backend.octree_mark_consecutive_segments(pidx.int().contiguous(), pack_infos, point_hierarchies)


# ------------------------------------------------------------------------------------------------
# sort / search / inverse CDF (no gradients)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def packed_sort_inplace(vals, pack_infos, return_idx=True):
    return _backend.packed_sort_qsort(vals.contiguous(), pack_infos, return_idx)


def packed_sort(vals, pack_infos):
    indices = packed_sort_inplace(vals.data.clone(), pack_infos, return_idx=True)
    return vals[indices], indices


@torch.no_grad()
def packed_searchsorted(bins, vals, pack_infos):
    return _backend.packed_searchsorted(bins.contiguous(), vals.contiguous(), pack_infos)


@torch.no_grad()
def packed_searchsorted_packed_vals(bins, pack_infos, vals, u_pack_infos):
    return _backend.packed_searchsorted_packed_vals(bins.contiguous(), pack_infos.contiguous(), vals.contiguous(),
                                                    u_pack_infos)


@torch.no_grad()
def packed_invert_cdf(bins, cdfs, u_vals, pack_infos) -> Tuple[torch.Tensor, torch.Tensor]:
    return _backend.packed_invert_cdf(bins.contiguous(), cdfs.contiguous(), u_vals.contiguous(), pack_infos)


# ------------------------------------------------------------------------------------------------
# differentiable reductions / scans
# ------------------------------------------------------------------------------------------------
def _lens(pack_infos):
    return pack_infos[..., 1]


class PackedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, pack_infos):
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(pack_infos)
        return _backend.packed_sum(feats, pack_infos)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None
        pack_infos, = ctx.saved_tensors
        return g.repeat_interleave(_lens(pack_infos), dim=0), None


def packed_sum(feats, pack_infos):
    feats = feats.contiguous()
    return PackedSum.apply(feats, pack_infos) if feats.requires_grad else _backend.packed_sum(feats, pack_infos)


def packed_mean(feats, pack_infos):
    return packed_sum(feats, pack_infos) / (pack_infos[:, 1] + 1e-8)


class PackedCumprod(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, pack_infos, exclusive, reverse):
        prod_ = _backend.packed_cumprod(feats, pack_infos, exclusive, reverse)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(feats, pack_infos, prod_)
            ctx.flags = (exclusive, reverse)
        return prod_

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        feats, pack_infos, prod_ = ctx.saved_tensors
        exclusive, reverse = ctx.flags
        # d prod_k / d x_i = prod_k / x_i for i inside the product: opposite-direction cumsum of prod*g, / x
        grad = _backend.packed_cumsum((prod_ * g).contiguous(), pack_infos, exclusive, not reverse) / feats
        grad[grad.isnan()] = 0
        return grad, None, None, None


def packed_cumprod(feats, pack_infos, exclusive: bool = False, reverse: bool = False):
    """Pack-wise cumulative product.  ``exclusive`` follows the reference kernel (see
    bindings._pack_ops.CUMPROD_EXCLUSIVE_DOCUMENTED)."""
    feats = feats.contiguous()
    if feats.requires_grad:
        return PackedCumprod.apply(feats, pack_infos, exclusive, reverse)
    return _backend.packed_cumprod(feats, pack_infos, exclusive, reverse)


class PackedCumsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, pack_infos, exclusive, reverse):
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(pack_infos)
            ctx.flags = (exclusive, reverse)
        return _backend.packed_cumsum(feats, pack_infos, exclusive, reverse)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pack_infos, = ctx.saved_tensors
        exclusive, reverse = ctx.flags
        return _backend.packed_cumsum(g.contiguous(), pack_infos, exclusive, not reverse), None, None, None


def packed_cumsum(feats, pack_infos, exclusive: bool = False, reverse: bool = False):
    feats = feats.contiguous()
    if feats.requires_grad:
        return PackedCumsum.apply(feats, pack_infos, exclusive, reverse)
    return _backend.packed_cumsum(feats, pack_infos, exclusive, reverse)


class PackedDiff(torch.autograd.Function):
    """forward difference d_i = x_{i+1} - x_i; last element of a pack from appends / last_fill / 0"""

    @staticmethod
    def forward(ctx, feats, pack_infos, pack_appends, pack_last_fill):
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(pack_infos)
            ctx.flags = (pack_appends is not None, pack_last_fill is not None)
        return _backend.packed_diff(feats, pack_infos, pack_appends, pack_last_fill)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        has_append, has_fill = ctx.flags
        pack_infos, = ctx.saved_tensors
        first, n = pack_infos[..., 0], pack_infos[..., 1]
        last = first + n - 1
        g = g.contiguous()
        grad_feat = None
        if ctx.needs_input_grad[0]:
            # x_i receives +g_{i-1} - g_i; the first element only -g_first
            grad_feat = -1 * _backend.packed_backward_diff(g, pack_infos, None, g[first].contiguous())
            if not has_append:   # last diff does not depend on x_last
                prev = g[last - 1]
                grad_feat[last] = torch.where(n.view([-1] + [1] * (prev.dim() - 1)) > 1, prev, g.new_tensor([0.]))
        g_app = g[last] if (has_append and ctx.needs_input_grad[2]) else None
        g_fill = g[last] if (has_fill and ctx.needs_input_grad[3]) else None
        return grad_feat, None, g_app, g_fill


def packed_diff(feats, pack_infos, pack_appends=None, pack_last_fill=None):
    feats = feats.contiguous()
    if feats.requires_grad:
        return PackedDiff.apply(feats, pack_infos, pack_appends, pack_last_fill)
    return _backend.packed_diff(feats, pack_infos, pack_appends, pack_last_fill)


class PackedBackwardDiff(torch.autograd.Function):
    """backward difference d_i = x_i - x_{i-1}; first element from prepends / first_fill / 0"""

    @staticmethod
    def forward(ctx, feats, pack_infos, pack_prepends, pack_first_fill):
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(pack_infos)
            ctx.flags = (pack_prepends is not None, pack_first_fill is not None)
        return _backend.packed_backward_diff(feats, pack_infos, pack_prepends, pack_first_fill)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        has_prepend, has_fill = ctx.flags
        pack_infos, = ctx.saved_tensors
        first, n = pack_infos[..., 0], pack_infos[..., 1]
        last = first + n - 1
        g = g.contiguous()
        grad_feat = None
        if ctx.needs_input_grad[0]:
            grad_feat = -1 * _backend.packed_diff(g, pack_infos, None, (-g[last]).contiguous())
            if not has_prepend:
                nxt = g[(first + 1).clamp(max=g.shape[0] - 1)]
                grad_feat[first] = torch.where(n.view([-1] + [1] * (nxt.dim() - 1)) > 1, -nxt, g.new_tensor([0.]))
        g_pre = -g[first] if (has_prepend and ctx.needs_input_grad[2]) else None
        g_fill = g[first] if (has_fill and ctx.needs_input_grad[3]) else None
        return grad_feat, None, g_pre, g_fill


def packed_backward_diff(feats, pack_infos, pack_prepends=None, pack_first_fill=None):
    feats = feats.contiguous()
    if feats.requires_grad:
        return PackedBackwardDiff.apply(feats, pack_infos, pack_prepends, pack_first_fill)
    return _backend.packed_backward_diff(feats, pack_infos, pack_prepends, pack_first_fill)


# ------------------------------------------------------------------------------------------------
# volume rendering
# ------------------------------------------------------------------------------------------------
class PackedAlphaToVW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, pack_infos, early_stop_eps, alpha_thre):
        weights = _backend.packed_alpha_to_vw_forward(alphas, pack_infos, early_stop_eps, alpha_thre, False)[0]
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(alphas, pack_infos, weights)
            ctx.cfg = (early_stop_eps, alpha_thre)
        return weights

    @staticmethod
    def backward(ctx, grad_weights):
        alphas, pack_infos, weights = ctx.saved_tensors
        eps, thre = ctx.cfg
        return _backend.packed_alpha_to_vw_backward(weights, grad_weights.contiguous(), alphas, pack_infos, eps,
                                                    thre), None, None, None


def packed_alpha_to_vw(alpha, pack_infos, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0):
    """visibility weights w_j = alpha_j * prod_{k<j} (1 - alpha_k), with early stop and alpha threshold"""
    alpha = alpha.contiguous()
    if alpha.requires_grad:
        return PackedAlphaToVW.apply(alpha, pack_infos, early_stop_eps, alpha_thre)
    return _backend.packed_alpha_to_vw_forward(alpha, pack_infos, early_stop_eps, alpha_thre, False)[0]


class PackedComposite(torch.autograd.Function):
    """The renderer's alpha-composite chain (reference: nr3d_lib/models/fields/nerf/renderer_mixin.py:298-311 =
    packed_alpha_to_vw -> packed_sum -> packed_div -> packed_sum(. * t) -> packed_sum(. * rgb)) as one kernel each way.
    Outputs: (vw [S], mask [num_rays], depth [num_rays], rgb [num_rays, 3] | None); all four are differentiable with
    respect to alpha, t and rgb."""

    @staticmethod
    def forward(ctx, alphas, t, rgb, pack_infos, rays_inds_hit, num_rays, early_stop_eps, alpha_thre, normalize_depth,
                packs_tile=False):
        vw, mask, depth, rgb_out = _backend.packed_composite_forward(alphas, t, rgb, pack_infos, rays_inds_hit, num_rays,
                                                                      early_stop_eps, alpha_thre, normalize_depth, packs_tile)
        ctx.save_for_backward(alphas, t, rgb, pack_infos, rays_inds_hit, vw, mask, depth)
        ctx.cfg = (early_stop_eps, alpha_thre, normalize_depth, packs_tile)
        ctx.set_materialize_grads(False)
        if rgb is None:
            return vw, mask, depth
        return vw, mask, depth, rgb_out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_vw, g_mask, g_depth, g_rgb=None):
        alphas, t, rgb, pack_infos, rays_inds_hit, vw, mask, depth = ctx.saved_tensors
        eps, thre, normalize, packs_tile = ctx.cfg
        c = lambda g: None if g is None else g.contiguous()
        ga, gt, gr = _backend.packed_composite_backward(alphas, vw, t, rgb, pack_infos, rays_inds_hit, eps, thre, normalize,
                                                        mask, depth, c(g_mask), c(g_depth), c(g_rgb), c(g_vw),
                                                        need_t=ctx.needs_input_grad[1],
                                                        need_rgb=rgb is not None and ctx.needs_input_grad[2],
                                                        packs_tile=packs_tile)
        return ga if ctx.needs_input_grad[0] else None, gt, gr, None, None, None, None, None, None, None


def packed_composite(alpha, t, rgb, pack_infos, rays_inds_hit=None, num_rays=None, early_stop_eps: float = 1e-4,
                     alpha_thre: float = 0.0, normalize_depth: bool = True, packs_tile: bool = False):
    """fused alpha composite of a packed volume buffer -> (vw, mask, depth, rgb | None); see PackedComposite.
    ``rays_inds_hit`` scatters the per-pack results into [num_rays] outputs (other rays stay zero).
    ``packs_tile``: the caller guarantees that the packs cover [0, S) exactly (a marcher's / a compaction's pack_infos do),
    so the per-sample outputs need no zero-fill; otherwise samples outside every pack come back as zeros."""
    if num_rays is None:
        num_rays = pack_infos.shape[0]
    alpha, t = alpha.contiguous().view(-1), t.contiguous().view(-1)
    rgb = None if rgb is None else rgb.contiguous().view(-1, 3)
    if rays_inds_hit is not None:
        rays_inds_hit = rays_inds_hit.contiguous()
    if alpha.requires_grad or t.requires_grad or (rgb is not None and rgb.requires_grad):
        out = PackedComposite.apply(alpha, t, rgb, pack_infos, rays_inds_hit, int(num_rays), early_stop_eps, alpha_thre,
                                    bool(normalize_depth), bool(packs_tile))
        return out if rgb is not None else (*out, None)
    return _backend.packed_composite_forward(alpha, t, rgb, pack_infos, rays_inds_hit, int(num_rays), early_stop_eps,
                                             alpha_thre, bool(normalize_depth), bool(packs_tile))


# True: selector -> compact indices / pack_infos (+ gathers) inside the library, one readback
# (bindings._pack_ops.packed_compression_compact); False: the reference's chain with two nonzero() (cross-check)
FUSED_COMPRESSION = True


@torch.no_grad()
def packed_volume_render_compression(alpha, pack_infos, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0):
    """-> (indices of packs that keep >= 1 sample, their compact pack_infos, indices of kept samples)"""
    if FUSED_COMPRESSION and alpha.is_cuda and alpha.dtype == torch.float32 and pack_infos.dtype == torch.int64:
        nidx_useful, compact_pi, pidx, *_ = _backend.packed_compression_compact(alpha.contiguous().view(-1), pack_infos.contiguous(),
                                                                               early_stop_eps, alpha_thre)
        return nidx_useful, compact_pi, pidx
    _, compact_pi, selector = _backend.packed_alpha_to_vw_forward(alpha.contiguous(), pack_infos, early_stop_eps,
                                                                  alpha_thre, True)
    pidx = selector.nonzero().long()[..., 0]
    nidx_useful = (compact_pi[:, 1] > 0).nonzero()[..., 0]
    return nidx_useful, compact_pi[nidx_useful].long(), pidx


@torch.no_grad()
def packed_volume_render_compression_gather(alpha, pack_infos, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, *,
                                            pack_tag=None, depths=None, deltas=None, samples=None, sample_idx=None):
    """packed_volume_render_compression + what its callers do with the result, in the same pass over the kept samples
    (nr3d_lib/graphics/nerf/nerf_ray_query.py:128-137): ``pack_tag`` int64 [P] (e.g. ray index of every pack) of the
    useful packs, and ``depths`` / ``deltas`` float [S], ``samples`` float [S, 3], ``sample_idx`` int64 [S] of the kept
    samples.  -> (nidx_useful, pack_infos, dict of the gathered arrays under the same names; 'pack_tag' -> tags)"""
    nidx, cpi, _, d1, d2, d3, dl = _backend.packed_compression_compact(
        alpha.contiguous().view(-1), pack_infos.contiguous(), early_stop_eps, alpha_thre, None,
        None if depths is None else depths.contiguous(), None if deltas is None else deltas.contiguous(),
        None if samples is None else samples.contiguous(), None if sample_idx is None else sample_idx.contiguous(), want_pidx=False)
    out = dict(depths=d1, deltas=d2, samples=d3, sample_idx=dl)
    if pack_tag is not None:
        out['pack_tag'] = pack_tag[nidx]
    return nidx, cpi, out


# ------------------------------------------------------------------------------------------------
# per-pack broadcast arithmetic with autograd: feats [N(,F)] (op) other [P(,F)]
# ------------------------------------------------------------------------------------------------
class _PackedArith(torch.autograd.Function):
    """op in {'add','sub','mul','div'}"""

    @staticmethod
    def forward(ctx, feats, other, pack_infos, op):
        ctx.op = op
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if op in ('mul', 'div'):
                ctx.save_for_backward(pack_infos, feats, other)
            else:
                ctx.save_for_backward(pack_infos)
        return getattr(_backend, f"packed_{op}")(feats, other, pack_infos)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        g = g.contiguous()
        op = ctx.op
        pack_infos = ctx.saved_tensors[0]
        gi = go = None
        if op == 'add':
            gi = g if ctx.needs_input_grad[0] else None
            go = _backend.packed_sum(g, pack_infos) if ctx.needs_input_grad[1] else None
        elif op == 'sub':
            gi = g if ctx.needs_input_grad[0] else None
            go = -1 * _backend.packed_sum(g, pack_infos) if ctx.needs_input_grad[1] else None
        elif op == 'mul':
            _, feats, other = ctx.saved_tensors
            gi = _backend.packed_mul(g, other, pack_infos) if ctx.needs_input_grad[0] else None
            go = _backend.packed_sum((g * feats).contiguous(), pack_infos) if ctx.needs_input_grad[1] else None
        else:
            _, feats, other = ctx.saved_tensors
            gi = _backend.packed_div(g, other, pack_infos) if ctx.needs_input_grad[0] else None
            if ctx.needs_input_grad[1]:   # d(x/o)/do = -x / o^2
                go = _backend.packed_sum(_backend.packed_div((-g * feats).contiguous(), (other * other).contiguous(),
                                                             pack_infos), pack_infos)
        return gi, go, None, None


# class names of the reference kept importable
class PackedAdd(_PackedArith): pass
class PackedSub(_PackedArith): pass
class PackedMul(_PackedArith): pass
class PackedDiv(_PackedArith): pass


def packed_add(feats, other, pack_infos):
    return PackedAdd.apply(feats.contiguous(), other.contiguous(), pack_infos.contiguous(), 'add')


def packed_sub(feats, other, pack_infos):
    return PackedSub.apply(feats.contiguous(), other.contiguous(), pack_infos.contiguous(), 'sub')


def packed_mul(feats, other, pack_infos):
    return PackedMul.apply(feats.contiguous(), other.contiguous(), pack_infos.contiguous(), 'mul')


def packed_div(feats, other, pack_infos):
    """pack-wise feats / other"""
    return PackedDiv.apply(feats.contiguous(), other.contiguous(), pack_infos.contiguous(), 'div')


def packed_matmul(feats, other, pack_infos):
    """results[i] = other[pack(i)] @ feats[i]; differentiable when any input requires grad (pure torch,
    like the reference), the HIP kernel otherwise."""
    if feats.requires_grad or other.requires_grad:
        return (feats.unsqueeze(-2) * torch.repeat_interleave(other, pack_infos[:, 1], dim=0)).sum(-1)
    return _backend.packed_matmul(feats.contiguous(), other.contiguous(), pack_infos.contiguous())


def _cmp(name):
    fn = getattr(_backend, f"packed_{name}")

    def f(feats, other, pack_infos):
        return fn(feats.contiguous(), other.contiguous(), pack_infos.contiguous())
    f.__name__ = f"packed_{name}"
    return f


packed_gt, packed_geq, packed_lt = _cmp("gt"), _cmp("geq"), _cmp("lt")
packed_leq, packed_eq, packed_neq = _cmp("leq"), _cmp("eq"), _cmp("neq")


# ------------------------------------------------------------------------------------------------
# interleave producers
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def interleave_arange_simple(stop, return_idx: bool = True):
    ret = _backend.interleave_arange(stop.contiguous(), return_idx)
    return ret if return_idx else ret[0]


@torch.no_grad()
def interleave_linstep(start, num_steps, step_size: Union[torch.Tensor, Number], return_idx: bool = True):
    step = step_size.contiguous() if isinstance(step_size, torch.Tensor) else step_size
    ret = _backend.interleave_linstep(start.contiguous(), num_steps.contiguous(), step, return_idx)
    return ret if return_idx else ret[0]


@torch.no_grad()
def interleave_arange(start, stop, step_size: Union[torch.Tensor, Number], return_idx: bool = True):
    same = (start.is_cuda and start.dim() == 1 and stop.shape == start.shape and stop.dtype == start.dtype
            and start.dtype in (torch.float32, torch.float64, torch.int32, torch.int64)
            and (not isinstance(step_size, torch.Tensor) or (step_size.shape == start.shape and step_size.dtype == start.dtype)))
    if same:
        num_steps = _backend.arange_num_steps(start, stop, step_size)           # one launch, ATen's arithmetic
    else:
        num_steps = stop.subtract(start).div(step_size).ceil().long()
    return interleave_linstep(start, num_steps, step_size, return_idx)


@torch.no_grad()
def interleave_linspace(start, stop, num_steps: Union[torch.Tensor, Number], return_idx: bool = True):
    step_size = (stop - start) / (num_steps - 1)
    if not isinstance(num_steps, torch.Tensor):
        num_steps = torch.full(start.shape, num_steps, device=start.device, dtype=torch.long)
    return interleave_linstep(start, num_steps, step_size, return_idx=return_idx)


def _perturb_inside_intervals(t_samples, deltas, pack_infos):
    """jitter each sample inside its own [t, t+dt) interval; deltas become consecutive differences, the last
    delta of each pack is kept"""
    t_samples = torch.addcmul(t_samples, torch.rand_like(deltas), deltas)
    last = pack_infos[..., 0] + pack_infos[..., 1] - 1
    new_deltas = t_samples.diff(append=t_samples.new_empty([1])).index_put_((last,), deltas[last])
    return t_samples, new_deltas


@torch.no_grad()
def interleave_sample_step_wrt_depth_clamped(near, far, max_steps: int = 512, dt_gamma: float = 0.01,
                                             min_step_size: float = 0.01, max_step_size: float = 1.0,
                                             step_size_factor: float = 1.0, perturb=False):
    t, dt, ridx, pack_infos = _backend.interleave_sample_step_wrt_depth_clamped(
        near.contiguous(), far.contiguous(), max_steps, dt_gamma * step_size_factor,
        min_step_size * step_size_factor, max_step_size * step_size_factor)
    if perturb:
        t, dt = _perturb_inside_intervals(t, dt, pack_infos)
    return t, dt, ridx, pack_infos


@torch.no_grad()
def interleave_sample_step_wrt_depth_in_packed_segments(near, far, entry, exit, seg_pack_infos, max_steps: int = 512,
                                                        dt_gamma: float = 0.01, min_step_size: float = 0.01,
                                                        max_step_size: float = 1e10, step_size_factor: float = 1.0,
                                                        perturb=False):
    n_rays = seg_pack_infos.shape[0]
    near = near if isinstance(near, torch.Tensor) else entry.new_full([n_rays], near)
    far = far if isinstance(far, torch.Tensor) else entry.new_full([n_rays], far)
    t, dt, sidx, ridx, ray_pack_infos = _backend.interleave_sample_step_wrt_depth_in_packed_segments(
        near.contiguous(), far.contiguous(), entry.contiguous(), exit.contiguous(), seg_pack_infos.contiguous(),
        max_steps, dt_gamma * step_size_factor, min_step_size * step_size_factor, max_step_size * step_size_factor)
    out_seg_pack_infos = get_pack_infos_from_boundary(mark_pack_boundaries(sidx))
    if perturb:
        t, dt = _perturb_inside_intervals(t, dt, ray_pack_infos)
    return t, dt, ridx, ray_pack_infos, sidx, out_seg_pack_infos


# ------------------------------------------------------------------------------------------------
# merging sorted packs
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def torch_intersect1d_unique(t1, t2):
    """set relations of two UNIQUE 1-D tensors: (union, inverse of t1, inverse of t2, indices of the common
    elements in t1 / in t2, indices exclusive to t1 / to t2)"""
    assert t1.dim() == t2.dim() == 1, "Requires t1, t2 to be unique 1D Tensors."
    n1 = t1.numel()
    u, inv, cnt = torch.unique(torch.cat([t1, t2]), return_counts=True, return_inverse=True)
    inv1, inv2 = inv[:n1].contiguous(), inv[n1:].contiguous()
    both1, both2 = cnt[inv1] == 2, cnt[inv2] == 2
    nz = lambda m: m.nonzero()[..., 0]
    return u, inv1, inv2, nz(both1), nz(both2), nz(~both1), nz(~both2)


def _scatter_vals(vals_a, pidx_a, vals_b, pidx_b, zeros=False):
    n = vals_a.numel() + vals_b.numel()
    val = vals_a.new_zeros([n]) if zeros else vals_a.new_empty([n])
    val[pidx_a], val[pidx_b] = vals_a, vals_b     # differentiable indexing
    return val


def merge_two_packs_sorted_aligned(vals_a, pack_infos_a, vals_b, pack_infos_b, b_sorted=True, return_val=False):
    """Merge two packed tensors pack by pack (same packs in both; vals_a sorted, vals_b optionally not).
    -> (pidx_a, pidx_b, pack_infos) or (merged vals, pack_infos)"""
    pidx_a, pidx_b, pack_infos = _backend.try_merge_two_packs_sorted_aligned(
        vals_a.contiguous(), pack_infos_a.contiguous(), vals_b.contiguous(), pack_infos_b.contiguous(), b_sorted)
    if return_val:
        return _scatter_vals(vals_a, pidx_a, vals_b, pidx_b), pack_infos
    return pidx_a, pidx_b, pack_infos


def _place_exclusive(pidx, pinfo_src, pinfo_dst_first):
    """packs copied unchanged: element k of such a pack goes to dst_first + k"""
    local, which = interleave_arange_simple(pinfo_src[:, 1].contiguous(), return_idx=True)
    pidx[local + pinfo_src[which, 0]] = local + pinfo_dst_first[which]


def merge_two_packs_sorted_a_includes_b(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b, b_sorted=True,
                                        return_val=False):
    """As merge_two_packs_sorted_aligned, but b's packs are a subset (by pack id ``nidx``) of a's packs.
    nidx_a / nidx_b sorted and unique."""
    assert vals_a.dim() == vals_b.dim() == 1, "Expect batched inputs with dim()==1"
    assert pack_infos_a.shape[0] == nidx_a.numel() and pack_infos_b.shape[0] == nidx_b.numel(), \
        "pack_infos and their nugget indices should have the same length"
    if nidx_a.numel() == nidx_b.numel() and torch.equal(nidx_a, nidx_b):
        return merge_two_packs_sorted_aligned(vals_a, pack_infos_a, vals_b, pack_infos_b, b_sorted=b_sorted,
                                              return_val=return_val)
    with torch.no_grad():
        where_b = torch.searchsorted(nidx_a, nidx_b)
        only_a = torch.ones(nidx_a.numel(), dtype=torch.bool, device=vals_a.device)
        only_a[where_b] = False
        only_a = only_a.nonzero().long()[..., 0]

        n_per_pack = pack_infos_a[:, 1].contiguous()
        n_per_pack.index_add_(0, where_b, pack_infos_b[:, 1])
        pack_infos = get_pack_infos_from_n(n_per_pack)
        pidx_a = pack_infos_a.new_full([vals_a.numel()], -1)

        # packs present in both: merge their compacted copy, then shift to the global layout
        ia = interleave_linstep(pack_infos_a[where_b, 0].contiguous(), pack_infos_a[where_b, 1].contiguous(), 1,
                                return_idx=False)
        pinfo_a_u = get_pack_infos_from_n(pack_infos_a[where_b, 1].contiguous())
        pa_u, pb_u, pinfo_u = merge_two_packs_sorted_aligned(vals_a[ia], pinfo_a_u, vals_b, pack_infos_b,
                                                             b_sorted=b_sorted)
        shift = (pack_infos[where_b, 0] - pinfo_u[:, 0]).contiguous()
        pidx_a[ia] = packed_add(pa_u, shift, pinfo_a_u)
        pidx_b = packed_add(pb_u, shift, pack_infos_b)
        if only_a.numel() > 0:
            _place_exclusive(pidx_a, pack_infos_a[only_a], pack_infos[only_a, 0])
    if return_val:
        return _scatter_vals(vals_a, pidx_a, vals_b, pidx_b, zeros=True), pack_infos
    return pidx_a, pidx_b, pack_infos


def merge_two_packs_sorted(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b, return_val=False):
    """Merge two sorted packed tensors whose pack id sets (sorted, unique) may differ arbitrarily."""
    if nidx_a.numel() == nidx_b.numel() and torch.equal(nidx_a, nidx_b):
        return merge_two_packs_sorted_aligned(vals_a, pack_infos_a, vals_b, pack_infos_b, b_sorted=True,
                                              return_val=return_val)
    if (vals_a.is_cuda and vals_a.dim() == 1 and vals_b.dim() == 1 and nidx_a.numel() > 0 and nidx_b.numel() > 0
            and vals_a.dtype == vals_b.dtype and vals_a.dtype in (torch.float32, torch.float64, torch.int32, torch.int64)):
        # the union of the pack lists in aligned form + the aligned merge kernel: a handful of launches and one readback
        pidx_a, pidx_b, pack_infos = _backend.merge_two_packs_sorted_general(
            vals_a.contiguous(), pack_infos_a.contiguous(), nidx_a.long().contiguous(), vals_b.contiguous(),
            pack_infos_b.contiguous(), nidx_b.long().contiguous())
        if return_val:
            return _scatter_vals(vals_a, pidx_a, vals_b, pidx_b, zeros=True), pack_infos
        return pidx_a, pidx_b, pack_infos
    return _merge_two_packs_sorted_torch(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b, return_val)


def _merge_two_packs_sorted_torch(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b, return_val=False):
    """the same result through set arithmetic in torch (the reference's own formulation; the cross-check of the tests)"""
    with torch.no_grad():
        u, inv_a, inv_b, com_a, com_b, only_a, only_b = torch_intersect1d_unique(nidx_a, nidx_b)
        n_per_pack = pack_infos_a.new_zeros([u.numel()])
        n_per_pack[inv_a] += pack_infos_a[:, 1]
        n_per_pack[inv_b] += pack_infos_b[:, 1]
        pack_infos = get_pack_infos_from_n(n_per_pack)
        pidx_a = pack_infos_a.new_full([vals_a.numel()], -1)
        pidx_b = pack_infos_b.new_full([vals_b.numel()], -1)
        if com_a.numel() > 0:
            ia, wa = interleave_linstep(pack_infos_a[com_a, 0].contiguous(), pack_infos_a[com_a, 1].contiguous(), 1, True)
            ib, wb = interleave_linstep(pack_infos_b[com_b, 0].contiguous(), pack_infos_b[com_b, 1].contiguous(), 1, True)
            pa_u, pb_u, pinfo_u = merge_two_packs_sorted_aligned(
                vals_a[ia], get_pack_infos_from_n(pack_infos_a[com_a, 1].contiguous()),
                vals_b[ib], get_pack_infos_from_n(pack_infos_b[com_b, 1].contiguous()))
            shift = pack_infos[inv_a[com_a], 0] - pinfo_u[:, 0]
            pidx_a[ia] = pa_u + shift[wa]
            pidx_b[ib] = pb_u + shift[wb]
        if only_a.numel() > 0:
            _place_exclusive(pidx_a, pack_infos_a[only_a], pack_infos[inv_a[only_a], 0])
        if only_b.numel() > 0:
            _place_exclusive(pidx_b, pack_infos_b[only_b], pack_infos[inv_b[only_b], 0])
    if return_val:
        return _scatter_vals(vals_a, pidx_a, vals_b, pidx_b, zeros=True), pack_infos
    return pidx_a, pidx_b, pack_infos


def merge_two_batch_a_includes_b(vals_a, nidx_a, vals_b, nidx_b, a_sorted=True, return_val=False):
    """Merge two BATCHED value sets ([n, width]) into packs; b's rows are a subset (by row id) of a's rows."""
    assert vals_a.dim() == vals_b.dim() == 2, "Expect batched inputs with dim()==2"
    assert vals_a.shape[0] == nidx_a.numel() and vals_b.shape[0] == nidx_b.numel(), \
        "Values and their nugget indices should have the same length"
    dev = vals_a.device
    n_a, wa, wb = nidx_a.numel(), vals_a.shape[-1], vals_b.shape[-1]
    if n_a == nidx_b.numel() and torch.equal(nidx_a, nidx_b):
        vals, order = torch.cat([vals_a, vals_b], -1).sort(-1)
        pack_infos = get_pack_infos_from_n(nidx_a.new_full([n_a], wa + wb))
        if return_val:
            return vals, pack_infos
        rank = order.argsort(-1)
        first = pack_infos[:, 0].unsqueeze(-1)
        return first + rank[:, :wa], first + rank[:, wa:], pack_infos
    where_b = torch.searchsorted(nidx_a, nidx_b)
    only_a = torch.ones(n_a, dtype=torch.bool, device=dev)
    only_a[where_b] = False
    only_a = only_a.nonzero().long()[..., 0]
    n_per_pack = nidx_a.new_full([n_a], wa)
    n_per_pack.index_fill_(0, where_b, wa + wb)
    pack_infos = get_pack_infos_from_n(n_per_pack)
    pidx_a = nidx_a.new_full(vals_a.shape, -1)
    rank = torch.cat([vals_a[where_b], vals_b], dim=1).argsort(-1).argsort(-1)
    first_u = pack_infos[where_b, 0].unsqueeze(-1)
    pidx_a[where_b] = first_u + rank[:, :wa]
    pidx_b = first_u + rank[:, wa:]
    if only_a.numel() > 0:
        local = torch.arange(wa, device=dev).unsqueeze(0) if a_sorted else vals_a[only_a].argsort(-1).argsort(-1)
        pidx_a[only_a] = pack_infos[only_a, 0].unsqueeze(-1) + local
    if return_val:
        return _scatter_vals(vals_a, pidx_a, vals_b, pidx_b), pack_infos
    return pidx_a, pidx_b, pack_infos


def merge_two_batch(vals_a, nidx_a, vals_b, nidx_b):
    raise NotImplementedError
