"""March -> (prune) -> query -> volume buffer: the driver that chains the three hot-path pieces.

Counterpart of ``nerf_ray_query_march_occ`` (nr3d_lib/graphics/nerf/nerf_ray_query.py:28-188) with the same
signature, ``ray_tested`` keys, model attribute protocol (``use_ts`` / ``use_fidx`` / ``use_bidx`` / ``use_pix`` /
``use_h_appear`` / ``use_view_dirs`` and their ``fwd_density_*`` twins), returned ``volume_buffer`` keys and
``details``; plus ``composite_packed_volume_buffer``, the packed branch of the renderer's composite
(nr3d_lib/models/fields/nerf/renderer_mixin.py:298-311).

Data flow on the device (no host sync except the marcher's single total readback and the two nonzero() compactions of
the pruning stage):  occ-grid march (HIP) -> density query on all marched samples under no_grad (LoTD forward, HIP) ->
alpha -> packed_volume_render_compression (HIP alpha_to_vw in compaction mode) keeps the samples that are still
visible (T >= early_stop_eps) -> the differentiable query runs only on those -> alpha -> caller composites."""
from typing import Callable, Dict, Optional, Tuple

import torch

from nr3d_lib_amd.graphics.nerf.nerf_utils import (packed_alpha_to_vw, packed_volume_render_compression, sigma_delta_to_alpha,
                                                    tau_to_alpha)
from nr3d_lib_amd.graphics.pack_ops import (packed_composite, packed_div, packed_sum,
                                            packed_volume_render_compression_gather)
from nr3d_lib_amd.profile import profile
from nr3d_lib_amd import _hip as _H

__all__ = ['nerf_ray_query_march_occ', 'composite_packed_volume_buffer']

_PER_RAY_KEYS = (('ts', 'rays_ts'), ('fidx', 'rays_fidx'), ('bidx', 'rays_bidx'), ('pix', 'rays_pix'),
                 ('h_appear', 'rays_h_appear'))
_PASSTHROUGH = ('flow_fwd', 'flow_fwd_pred_bwd', 'flow_bwd', 'flow_bwd_pred_fwd', 'sigma_static', 'rgb_static',
                'sigma_dynamic', 'rgb_dynamic')


# The differentiable query may run on the rendered samples in SPATIAL order (a Morton curve over their bounding box,
# nr3d_spatial_order) with its outputs put back into ray order.  Ray-major samples of neighbouring rays share grid cells but sit
# a ray's length apart; along the curve they are consecutive lanes, which the LoTD backward merges before its scatter and the
# forward's gathers coalesce.  Results are the same values in the same places IF the field is evaluated point by point; parameter
# gradients differ by summation order only.
# OPT-IN (round 6, advisor): the reference always queries in ray order (nerf_ray_query.py:150-166), and the reordering is only valid
# for a model whose forward is strictly point-wise and returns every per-sample output as a TOP-LEVEL tensor of its dict.  A model
# declares that with the attribute ``pointwise_forward = True`` (tools/demo_field.py does); without it the query runs in ray order.
# A per-sample tensor nested in a dict / tuple / list of the output cannot be put back and raises instead of coming out permuted.
# SPATIAL_ORDER_BITS: 0 = off for every model; otherwise bits per dimension of the curve's grid.
# Measured on the full loop (262 144 rays, 1.67 M rendered samples; profiles/r05e_full_loop_*): stage B of dL/dparam 673 -> 450 us, the
# encoder forward on the rendered samples 309 -> 198 us; the order itself costs 190 us (sort 3 x 31, moving inputs / outputs / gradients
# 35 + 44 + 29, keys and bounds 20) -- 5.45 -> 5.42 ms per iteration at 8 bits (5.49-5.56 at 5-7 bits): it pays for itself, no more.
SPATIAL_ORDER_BITS = 8
SPATIAL_ORDER_MIN_SAMPLES = 1 << 19      # below this the backward is launch bound and the order does not pay


class _MoveRows(torch.autograd.Function):
    """one or two per-sample float32 arrays between the spatial order and the samples' own order (one launch; scatter:
    out[order[k]] = in[k], else out[k] = in[order[k]]).  order is a permutation, so neither direction accumulates and the
    backward is the same function in the other direction -- applied as a Function, so a create_graph=True backward keeps its graph."""

    @staticmethod
    def forward(ctx, order, a, b, scatter):
        ctx.save_for_backward(order)
        ctx.scatter = scatter
        a_out, b_out = _H.order_move_rows(order, a, b, scatter=scatter)
        return (a_out, b_out) if b is not None else a_out

    @staticmethod
    def backward(ctx, ga, gb=None):
        order, = ctx.saved_tensors
        back = not ctx.scatter
        if ga is not None and gb is not None:
            g1, g2 = _MoveRows.apply(order, ga, gb, back)
            return None, g1, g2, None
        g1 = _MoveRows.apply(order, ga, None, back) if ga is not None else None
        g2 = _MoveRows.apply(order, gb, None, back) if gb is not None else None
        return None, g1, g2, None


def _nested_per_sample(v, n):
    if isinstance(v, dict):
        return any(_nested_per_sample(u, n) or (torch.is_tensor(u) and u.dim() >= 1 and u.shape[0] == n) for u in v.values())
    if isinstance(v, (list, tuple)):
        return any(_nested_per_sample(u, n) or (torch.is_tensor(u) and u.dim() >= 1 and u.shape[0] == n) for u in v)
    return False


def _from_spatial_order(order, net_out, n):
    """every per-sample tensor of the field's output dict back in the samples' own order (float32: two per launch)"""
    for k, v in net_out.items():
        if _nested_per_sample(v, n):
            raise RuntimeError(f"nerf_ray_query_march_occ: output {k!r} of a pointwise_forward model nests per-sample tensors in a "
                               "container; the spatial-order query can only put top-level tensors back into ray order "
                               "(drop `pointwise_forward`, or return them at the top level)")
    keys = [k for k, v in net_out.items() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n]
    out = dict(net_out)
    fast = [k for k in keys if net_out[k].dtype == torch.float32 and net_out[k].is_cuda]
    slow = [k for k in keys if k not in fast]
    if slow:                                               # other dtypes (half features, integer tags): one inverse, plain index ops
        inv = torch.empty(n, dtype=torch.int64, device=order.device)
        inv[order.long()] = torch.arange(n, device=order.device)
        for k in slow:
            out[k] = net_out[k].index_select(0, inv)
    for i in range(0, len(fast), 2):
        pair = fast[i:i + 2]
        if len(pair) == 2:
            out[pair[0]], out[pair[1]] = _MoveRows.apply(order, net_out[pair[0]], net_out[pair[1]], True)
        else:
            out[pair[0]] = _MoveRows.apply(order, net_out[pair[0]], None, True)
    return out


def _flag(model, name):
    return bool(getattr(model, name, False))


def _march(model, ray_tested, rays_o, rays_d, near, far, perturb, march_cfg):
    """dispatch on what the accel's ray_march accepts (single / dynamic / batched / batched-dynamic accels of the
    reference differ only in the extra per-ray arguments, nerf_ray_query.py:84-104)"""
    accel = model.accel
    if hasattr(accel, 'cur_batch__ray_march'):
        extra = [ray_tested['rays_bidx']] + ([ray_tested['rays_ts']] if getattr(accel, 'is_dynamic', False) else [])
        return accel.cur_batch__ray_march(rays_o, rays_d, *extra, near=near, far=far, perturb=perturb, **march_cfg)
    if getattr(accel, 'is_dynamic', False):
        return accel.ray_march(rays_o, rays_d, ray_tested['rays_ts'], near=near, far=far, perturb=perturb, **march_cfg)
    return accel.ray_march(rays_o, rays_d, near=near, far=far, perturb=perturb, **march_cfg)


def nerf_ray_query_march_occ(model, ray_tested: Dict[str, torch.Tensor], with_rgb: bool = True, perturb: bool = False,
                             march_cfg=dict(), compression=True, bypass_sigma_fn: Optional[Callable] = None,
                             bypass_alpha_fn: Optional[Callable] = None, forward_params: dict = {}) -> Tuple[dict, dict]:
    assert hasattr(model, 'forward'), "model.forward() is requried"
    assert getattr(model, 'accel', None) is not None, "model.accel is required"
    if not with_rgb:
        assert hasattr(model, 'forward_density'), "model.forward_density() is requried"
    if compression:
        assert hasattr(model, 'query_density'), "model.query_density() is requried"
    assert (bypass_sigma_fn is None) or (bypass_alpha_fn is None), \
        "Please pass at most one of bypass_sigma_fn or bypass_alpha_fn"

    # which per-ray attributes go into the density-only query and into the full query
    density_uses = {k: _flag(model, 'use_' + k) if k in ('ts', 'fidx', 'bidx') else _flag(model, 'fwd_density_use_' + k)
                    for k, _ in _PER_RAY_KEYS}
    full_uses = {k: density_uses[k] or (_flag(model, 'use_' + k) and with_rgb) for k, _ in _PER_RAY_KEYS}
    density_view = _flag(model, 'fwd_density_use_view_dirs')
    full_view = density_view or (_flag(model, 'use_view_dirs') and with_rgb)

    empty = dict(type='empty', rays_inds_hit=[])
    if ray_tested['num_rays'] == 0:
        return empty, {}
    rays_o, rays_d = ray_tested['rays_o'], ray_tested['rays_d']
    near, far, rays_inds = ray_tested['near'], ray_tested['far'], ray_tested['rays_inds']
    assert rays_o.dim() == 2 and rays_d.dim() == 2
    dtype = rays_o.dtype
    view_dirs = None
    if full_view:
        view_dirs = rays_d / rays_d.detach().norm(dim=-1, keepdim=True).clamp_min(1.0e-10)

    def query_kwargs(samples, ridx, uses, use_view):
        kw = dict(x=samples)
        for k, src in _PER_RAY_KEYS:
            if uses[k]:
                kw[k] = ray_tested[src][ridx]
        if use_view:
            kw['v'] = view_dirs[ridx]
        return kw

    with profile("Ray marching"):
        marched = _march(model, ray_tested, rays_o, rays_d, near, far, perturb, march_cfg)
    if marched.ridx_hit is None:
        return empty, {}

    pack_infos, ridx_hit, ridx_all = marched.pack_infos, marched.ridx_hit, marched.ridx
    depth_samples, deltas, samples = marched.depth_samples, marched.deltas, marched.samples
    details = {'march.num_per_ray': marched.pack_infos[:, 1]}
    nidx_useful = None
    if compression:
        with torch.no_grad(), profile("Visibility pruning"):
            kw = query_kwargs(samples, ridx_all, density_uses, density_view)
            if bypass_alpha_fn is not None:
                alphas = bypass_alpha_fn(**kw)
            else:
                sigmas = bypass_sigma_fn(**kw) if bypass_sigma_fn is not None else model.query_density(**kw)
                alphas = sigma_delta_to_alpha(sigmas.view(-1), deltas.view(-1))
            fused = (FUSED_PRUNE and alphas.is_cuda and alphas.dtype == torch.float32 and depth_samples.dtype == torch.float32
                     and samples.dtype == torch.float32 and ridx_all.dtype == torch.int64)
            if fused:
                # selector -> compact packs + the kept samples' depth / delta / position / ray index in one pass, one readback
                nidx_useful, pack_infos, kept = packed_volume_render_compression_gather(
                    alphas, marched.pack_infos, pack_tag=ridx_hit, depths=depth_samples.view(-1), deltas=deltas.view(-1),
                    samples=samples.view(-1, 3), sample_idx=ridx_all.view(-1))
            else:
                nidx_useful, pack_infos, pidx_useful = packed_volume_render_compression(alphas, marched.pack_infos)
        if nidx_useful.numel() == 0:
            return empty, {}
        details['render.num_per_ray0'] = marched.pack_infos[:, 1]
        if fused:
            ridx_hit, ridx_all = kept['pack_tag'], kept['sample_idx']
            depth_samples, deltas, samples = kept['depths'], kept['deltas'], kept['samples']
        else:
            ridx_hit, ridx_all = ridx_hit[nidx_useful], ridx_all[pidx_useful]
            depth_samples, deltas, samples = depth_samples[pidx_useful], deltas[pidx_useful], samples[pidx_useful]
    details['render.num_per_ray'] = pack_infos[:, 1]

    # packs_tile: the packs of the marcher / the compaction cover every sample exactly once (no zero-fill in the composite)
    volume_buffer = dict(type='packed', packs_tile=True, rays_inds_hit=rays_inds[ridx_hit], pack_infos_hit=pack_infos,
                         t=depth_samples.to(dtype))
    if full_uses['bidx'] and nidx_useful is not None:
        volume_buffer['rays_bidx_hit'] = ray_tested['rays_bidx'][nidx_useful]     # indexing as in the reference (:148)

    with profile("Query"):
        n_render = samples.shape[0]
        spatial = (SPATIAL_ORDER_BITS and _flag(model, 'pointwise_forward') and n_render >= SPATIAL_ORDER_MIN_SAMPLES and samples.is_cuda and samples.dtype == torch.float32
                   and samples.dim() == 2 and samples.shape[1] == 3 and ridx_all.dtype == torch.int64 and torch.is_grad_enabled()
                   and not samples.requires_grad and (view_dirs is None or (view_dirs.dtype == torch.float32 and not view_dirs.requires_grad)))
        if not spatial:
            kw = query_kwargs(samples, ridx_all, full_uses, full_view)
            net_out = (model.forward if with_rgb else model.forward_density)(**kw, **forward_params)
        else:
            order = _H.spatial_order(samples.contiguous(), SPATIAL_ORDER_BITS)
            # position, ray index and view direction of the sample at every position of the order: one launch
            x_s, ridx_s, v_s = _H.order_gather_inputs(order, samples.contiguous(), ridx_all.contiguous(),
                                                      view_dirs.contiguous() if full_view else None)
            kw = dict(x=x_s)
            for k, src in _PER_RAY_KEYS:
                if full_uses[k]:
                    kw[k] = ray_tested[src][ridx_s]
            if full_view:
                kw['v'] = v_s
            net_out = _from_spatial_order(order, (model.forward if with_rgb else model.forward_density)(**kw, **forward_params), n_render)
    if with_rgb:
        volume_buffer['rgb'] = net_out['rgb'].to(dtype)
    volume_buffer['deltas'] = deltas.to(dtype)
    volume_buffer['sigma'] = net_out['sigma'].to(dtype)
    volume_buffer['opacity_alpha'] = sigma_delta_to_alpha(volume_buffer['sigma'], volume_buffer['deltas'])
    for k in _PASSTHROUGH:
        if k in net_out:
            volume_buffer[k] = net_out[k].to(dtype)
    return volume_buffer, details


# True: pruning = compaction + gathers inside the library (packed_volume_render_compression_gather, one readback);
# False: packed_volume_render_compression + index gathers as in the reference (cross-check)
FUSED_PRUNE = True

# True: one fused kernel each way (graphics.pack_ops.packed_composite); False: the reference's op chain
# (packed_alpha_to_vw -> packed_sum -> packed_div -> packed_sum x2), kept for A/B measurements and as a cross-check
FUSED_COMPOSITE = True


def composite_packed_volume_buffer(volume_buffer: dict, num_rays: int, with_rgb: bool = True,
                                   depth_use_normalized_vw: bool = True, device=None, dtype=torch.float32) -> dict:
    """alpha-composite a packed volume buffer into per-ray mask / depth / rgb (renderer_mixin.py:270-311):
    rays that were not hit keep zeros."""
    if device is None:
        device = volume_buffer['pack_infos_hit'].device if volume_buffer['type'] != 'empty' else 'cpu'
    if volume_buffer['type'] == 'empty':
        out = dict(mask_volume=torch.zeros(num_rays, device=device, dtype=dtype),
                   depth_volume=torch.zeros(num_rays, device=device, dtype=dtype))
        if with_rgb:
            out['rgb_volume'] = torch.zeros(num_rays, 3, device=device, dtype=dtype)
        return out
    assert volume_buffer['type'] == 'packed', "composite_packed_volume_buffer: packed buffers only"
    pi, hit = volume_buffer['pack_infos_hit'], volume_buffer['rays_inds_hit']
    alpha = volume_buffer['opacity_alpha']
    if FUSED_COMPOSITE and alpha.dtype == torch.float32 and dtype == torch.float32 and alpha.is_cuda:
        rgb = volume_buffer['rgb'].view(-1, 3) if with_rgb else None
        vw, mask, depth, rgb_out = packed_composite(alpha.view(-1), volume_buffer['t'].view(-1), rgb, pi, hit.long(), num_rays,
                                                    normalize_depth=depth_use_normalized_vw,
                                                    packs_tile=bool(volume_buffer.get('packs_tile', False)))
        volume_buffer['vw'] = vw
        out = dict(mask_volume=mask, depth_volume=depth)
        if with_rgb:
            out['rgb_volume'] = rgb_out
        return out
    out = dict(mask_volume=torch.zeros(num_rays, device=device, dtype=dtype),
               depth_volume=torch.zeros(num_rays, device=device, dtype=dtype))
    if with_rgb:
        out['rgb_volume'] = torch.zeros(num_rays, 3, device=device, dtype=dtype)
    volume_buffer['vw'] = vw = packed_alpha_to_vw(alpha, pi)
    vw_sum = packed_sum(vw.view(-1), pi)
    out['mask_volume'][hit] = vw_sum
    w_depth = packed_div(vw, vw_sum + 1e-10, pi) if depth_use_normalized_vw else vw.view(-1)
    out['depth_volume'][hit] = packed_sum(w_depth * volume_buffer['t'].view(-1), pi)
    if with_rgb:
        out['rgb_volume'][hit] = packed_sum(vw.view(-1, 1) * volume_buffer['rgb'].view(-1, 3), pi)
    return out
