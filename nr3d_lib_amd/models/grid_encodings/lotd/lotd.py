"""LoTD autograd surface -- counterpart of the reference's
nr3d_lib/models/grid_encodings/lotd/lotd.py (LoDType :31-38, generate_meta :40-45, LoTDFunction :48-119,
LoTDFunctionFwdDydx :121-191, LoTDFunctionBwdDydx :193-268, functional wrappers :270-319, LoTD :321-502).

Same public names, call signatures and semantics:
  * inputs are clamped to [1e-6, 1-1e-6] and flattened over their leading dims;
  * ``loss_scale`` (128 for fp16 params, else 1) multiplies dL/dy on the way in and divides every gradient on the way out -- as two
    extra passes only with ``APPLY_LOSS_SCALE`` (below: the accumulation here is exact, the scale changes no bit);
  * first-order gradients are once-differentiable; second-order gradients (eikonal / nablas) go through
    LoTDFunctionFwdDydx + LoTDFunctionBwdDydx, which back-propagate to dL/dy and to the params.
"""
from enum import Enum
from math import prod
from typing import List, Optional, Union

import numpy as np
import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from nr3d_lib_amd.profile import profile
import nr3d_lib_amd.bindings._lotd as _backend

__all__ = ['LoDType', 'generate_meta', 'LoTDFunction', 'LoTDFunctionFwdDydx', 'LoTDFunctionBwdDydx',
           'lotd_encoding', 'lotd_encoding_fwd_dydx', 'lotd_encoding_bwd_dydx', 'lotd_get_grid_index', 'LoTD']


class LoDType(Enum):
    Dense = int(_backend.LoDType.Dense)
    VectorMatrix = int(_backend.LoDType.VectorMatrix)
    CP = int(_backend.LoDType.CP)
    CPfast = int(_backend.LoDType.CPfast)
    NPlaneMul = int(_backend.LoDType.NPlaneMul)
    NPlaneSum = int(_backend.LoDType.NPlaneSum)
    Hash = int(_backend.LoDType.Hash)


def generate_meta(n_input_dim, lod_res, lod_n_feats, lod_types, hashmap_size=None, use_smooth_step=False):
    n = len(lod_res)
    if isinstance(lod_n_feats, int):
        lod_n_feats = [lod_n_feats] * n
    if isinstance(lod_types, str):
        lod_types = [lod_types] * n
    return _backend.LoDMeta(n_input_dim, lod_res, lod_n_feats, lod_types, hashmap_size, use_smooth_step)


_LO, _HI = 1.0e-6, 1 - 1.0e-6


def _prep(x, bidx):
    """clamp + flatten the points; flatten the optional per-point batch index"""
    prefix = x.shape[:-1]
    xc = x.clamp(_LO, _HI)
    if bidx is not None:
        bidx = bidx.contiguous().long().flatten()
    return prefix, xc, bidx


def _unflat(t, prefix):
    return None if t is None else t.unflatten(0, prefix)


# The reference multiplies dL/dy by ``loss_scale`` (2^7 for half tables) on the way in and divides every gradient by it on the way out
# (lotd.py:96-119) to keep its half ``atomicAdd`` scatter out of the underflow range.  The kernels here read half dL/dy as it is,
# accumulate exactly (64-bit fixed point / fp64) and round each gradient ONCE, so scaling by a power of two and back changes no bit of a
# normal result (it commutes with the rounding; it would only add an overflow to inf at |dL/dy| > 512) -- and costs two passes over
# [N, E] and one over the table gradient (78 + 10 us per backward of the full loop).  False: the protocol's arithmetic without those
# passes; True: the reference's protocol literally.  ``LoTD.loss_scale`` keeps the reference's value either way.
APPLY_LOSS_SCALE = False


def _scaled(t, s):
    return t if (t is None or s == 1.0 or not APPLY_LOSS_SCALE) else t / s


def _times(t, s):
    """t * loss_scale without the extra pass over t when the scale is 1 (fp32) or not applied (APPLY_LOSS_SCALE)"""
    return t if (s == 1.0 or not APPLY_LOSS_SCALE) else t * s


class LoTDFunction(torch.autograd.Function):
    """y = LoTD(x, grid) with first-order gradients wrt x and grid."""

    @staticmethod
    def forward(ctx, meta, x, grid, bidx=None, batch_offsets=None, batch_data_size=None, loss_scale=1.0,
                max_level=None):
        need_x, need_g = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        with profile(f"LoTDFunction.forward(with_grad={need_x})"):
            ctx.set_materialize_grads(False)
            prefix, xc, bidx = _prep(x, bidx)
            y, dy_dx = _backend.lod_fwd(meta, xc.flatten(0, -2), grid, bidx, batch_offsets, batch_data_size,
                                        max_level, need_x)
            if need_x or need_g:
                ctx.save_for_backward(xc, grid, dy_dx, bidx, batch_offsets)
                ctx.cfg = (meta, prefix, batch_data_size, loss_scale, max_level)
            return y.unflatten(0, prefix)

    @staticmethod
    @once_differentiable
    @profile
    def backward(ctx, dL_dy):
        dL_dx = dL_dgrid = None
        need_x, need_g = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if dL_dy is not None and (need_x or need_g):
            xc, grid, dy_dx, bidx, batch_offsets = ctx.saved_tensors
            meta, prefix, bds, ls, max_level = ctx.cfg
            dL_dx, dL_dgrid = _backend.lod_bwd(meta, _times(dL_dy.flatten(0, -2), ls), xc.flatten(0, -2), grid, dy_dx,
                                               bidx, batch_offsets, bds, max_level, need_x, need_g)
            dL_dx, dL_dgrid = _scaled(_unflat(dL_dx, prefix), ls), _scaled(dL_dgrid, ls)
        return None, dL_dx, dL_dgrid, None, None, None, None, None


class LoTDFunctionFwdDydx(torch.autograd.Function):
    """(y, dy_dx) = LoTD(x, grid).  dy_dx is a non-differentiable by-product that LoTDFunctionBwdDydx turns
    into nablas; backward() gives the usual first-order gradients of y."""

    @staticmethod
    @profile
    def forward(ctx, meta, x, grid, bidx=None, batch_offsets=None, batch_data_size=None, loss_scale=1.0,
                max_level=None, need_dL_dinput: Optional[bool] = None):
        if need_dL_dinput is None:
            need_dL_dinput = torch.is_grad_enabled() and x.requires_grad
        ctx.set_materialize_grads(False)
        prefix, xc, bidx = _prep(x, bidx)
        y, dy_dx = _backend.lod_fwd(meta, xc.flatten(0, -2), grid, bidx, batch_offsets, batch_data_size, max_level, True)
        ctx.save_for_backward(xc, grid, dy_dx, bidx, batch_offsets)
        ctx.cfg = (meta, prefix, batch_data_size, loss_scale, max_level, need_dL_dinput)
        ctx.mark_non_differentiable(dy_dx)
        return y.unflatten(0, prefix), dy_dx

    @staticmethod
    @once_differentiable
    @profile
    def backward(ctx, dL_dy, _unused):
        dL_dx = dL_dgrid = None
        if dL_dy is not None:
            xc, grid, dy_dx, bidx, batch_offsets = ctx.saved_tensors
            meta, prefix, bds, ls, max_level, need_dL_dinput = ctx.cfg
            with torch.no_grad():
                # x's gradient is governed by `need_dL_dinput`, not by autograd's needs_input_grad
                dL_dx, dL_dgrid = _backend.lod_bwd(meta, _times(dL_dy.flatten(0, -2), ls), xc.flatten(0, -2), grid, dy_dx,
                                                   bidx, batch_offsets, bds, max_level, need_dL_dinput,
                                                   ctx.needs_input_grad[2])
                dL_dx, dL_dgrid = _scaled(_unflat(dL_dx, prefix), ls), _scaled(dL_dgrid, ls)
        return None, dL_dx, dL_dgrid, None, None, None, None, None, None


class LoTDFunctionBwdDydx(torch.autograd.Function):
    """nablas = dL_dy . dy_dx as a differentiable op: backward() yields the second-order terms
    d(nablas)/d(dL_dy), d(nablas)/d(grid) (and, not requested by default, d(nablas)/dx)."""

    @staticmethod
    @profile
    def forward(ctx, meta, dL_dy, x, grid, dy_dx, bidx, batch_offsets, batch_data_size, loss_scale, max_level,
                grad_guard):
        ctx.set_materialize_grads(False)
        prefix, xc, bidx = _prep(x, bidx)
        dL_dx, _ = _backend.lod_bwd(meta, _times(dL_dy.flatten(0, -2), loss_scale), xc.flatten(0, -2), grid, dy_dx, bidx,
                                    batch_offsets, batch_data_size, max_level, True, False)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            ctx.save_for_backward(dL_dy, xc, grid, dy_dx, bidx, batch_offsets)
            ctx.cfg = (meta, batch_data_size, loss_scale, max_level, grad_guard)
        return _scaled(_unflat(dL_dx, prefix), loss_scale)

    @staticmethod
    @once_differentiable
    @profile
    def backward(ctx, dL_ddLdx):
        g_dLdy = g_x = g_grid = None
        if dL_ddLdx is not None:
            dL_dy, xc, grid, dy_dx, bidx, batch_offsets = ctx.saved_tensors
            meta, bds, ls, max_level, grad_guard = ctx.cfg
            prefix = xc.shape[:-1]
            # loss-scale bookkeeping: d/d(dL_dy) is linear in dL_ddLdx only; the other two also carry dL_dy * ls
            g_dLdy, g_grid, g_x = _backend.lod_bwd_bwd_input(
                meta, dL_ddLdx.flatten(0, -2).contiguous(), _times(dL_dy.flatten(0, -2), ls), xc.flatten(0, -2), grid, dy_dx,
                bidx, batch_offsets, bds, max_level, ctx.needs_input_grad[1], ctx.needs_input_grad[3], False)
            g_dLdy = _unflat(g_dLdy, prefix)
            g_grid = _scaled(g_grid, ls)
            g_x = _scaled(_unflat(g_x, prefix), ls)
            if grad_guard is not None and (g_grid is not None or g_dLdy is not None):
                grad_guard.custom_grad_clip_step(dL_ddLdx, dy_dx, g_grid, g_dLdy)
        return None, g_dLdy, g_x, g_grid, None, None, None, None, None, None, None


def _batch_mode(input, bidx, input_batched):
    if input_batched:
        return None, prod(input.shape[1:-1])
    return bidx, 0


def _loss_scale(params):
    return 128.0 if params.dtype == torch.float16 else 1.0


def lotd_encoding(input, params, bidx=None, batch_offsets=None, input_batched=False, max_level=None, meta=None,
                  n_input_dim=None, lod_res=None, lod_n_feats=None, lod_types=None):
    if meta is None:
        meta = generate_meta(n_input_dim, lod_res, lod_n_feats, lod_types)
    bidx, bds = _batch_mode(input, bidx, input_batched)
    return LoTDFunction.apply(meta, input, params, bidx, batch_offsets, bds, _loss_scale(params), max_level)


def lotd_encoding_fwd_dydx(input, params, bidx=None, batch_offsets=None, input_batched=False, max_level=None,
                           need_dL_dinput=None, meta=None, n_input_dim=None, lod_res=None, lod_n_feats=None,
                           lod_types=None):
    if need_dL_dinput is None:
        need_dL_dinput = torch.is_grad_enabled() and input.requires_grad
    if meta is None:
        meta = generate_meta(n_input_dim, lod_res, lod_n_feats, lod_types)
    bidx, bds = _batch_mode(input, bidx, input_batched)
    y, dy_dx = LoTDFunctionFwdDydx.apply(meta, input, params, bidx, batch_offsets, bds, _loss_scale(params), max_level,
                                         need_dL_dinput)
    return y, dy_dx, meta


def lotd_encoding_bwd_dydx(meta, dL_dy, dy_dx, input, params, bidx=None, batch_offsets=None, input_batched=False,
                           max_level=None):
    bidx, bds = _batch_mode(input, bidx, input_batched)
    # NOTE: the reference passes 10 positional args here (its apply() then fails on the missing grad_guard);
    # grad_guard=None is supplied explicitly.
    return LoTDFunctionBwdDydx.apply(meta, dL_dy, input, params, dy_dx, bidx, batch_offsets, bds, _loss_scale(params),
                                     max_level, None)


def lotd_get_grid_index(meta, input, bidx=None, batch_offsets=None, input_batched=False, max_level=None):
    bidx, bds = _batch_mode(input, bidx, input_batched)
    return _backend.lod_get_grid_index(meta, input, bidx, batch_offsets, bds, max_level)


# True: LoTD.forward_decoded runs the fused encode + decode kernel (csrc/lotd_mlp.hip) where it applies; False (default): the two ops.
# Round-6 experiment, MEASURED SLOWER and therefore off: on the 6.9 M ray-coherent marched samples of a 262 144-ray pass the fused
# kernel takes 3.03 ms against 1.95 ms for encoder + decoder (profiles/r06_fused_query_experiment.txt) -- a wave that owns its samples
# walks the 16 levels one after another, so every CU gathers from all 16 tables at once (46 MiB against 4 MiB of L2 per XCD), where
# the level-major forward keeps ONE table hot per XCD and 32 waves of gathers in flight per CU; the 883 MB of features it saves
# are worth less than that.  Values are bit-identical to the two ops; kept for small batches' launch count and as a cross-check.
FUSE_DECODED = False


class LoTD(nn.Module):
    """Parameter-free encoder module (the grid is passed to forward); mirrors the reference's LoTD
    (lotd.py:321-502) including its pickling protocol and read-only meta properties."""

    def __init__(self, in_features, lod_res: Union[List[int], List[List[int]]], lod_n_feats: Union[int, List[int]],
                 lod_types: Union[str, List[str]], hashmap_size: int = None, log2_hashmap_size: int = None,
                 use_smooth_step=False, use_profile=False, dtype=torch.half, device=None):
        super().__init__()
        assert dtype in (torch.float, torch.float16), "dtype must be one of torch.float or torch.float16"
        self.params = dict(in_features=in_features, lod_res=lod_res, lod_n_feats=lod_n_feats, lod_types=lod_types,
                           hashmap_size=hashmap_size, log2_hashmap_size=log2_hashmap_size,
                           use_smooth_step=use_smooth_step, use_profile=use_profile, dtype=dtype, device=device)
        self.dtype = dtype
        self.loss_scale = 128.0 if dtype == torch.float16 else 1.0
        if log2_hashmap_size is not None:
            assert hashmap_size is None, "Do not specify `hashmap_size` when `log2_hashmap_size` is already specified."
            hashmap_size = 2 ** log2_hashmap_size
        self.meta = generate_meta(in_features, lod_res, lod_n_feats, lod_types, hashmap_size, use_smooth_step)
        self.meta.c_profile = bool(use_profile)

    # ---- read-only views of the meta ----
    in_features = property(lambda self: self.meta.n_dims_to_encode)
    out_features = property(lambda self: self.meta.n_encoded_dims)
    n_levels = property(lambda self: self.meta.n_levels)
    n_params = property(lambda self: self.meta.n_params)
    level_res_multidim = property(lambda self: self.meta.level_res_multidim)
    level_types = property(lambda self: [LoDType(t) for t in self.meta.level_types])
    level_types_str = property(lambda self: self.meta.level_types_str)
    level_sizes = property(lambda self: self.meta.level_sizes)
    level_offsets = property(lambda self: self.meta.level_offsets)
    level_n_feats = property(lambda self: self.meta.level_n_feats)
    level_n_params = property(lambda self: self.meta.level_n_params)

    @property
    def level_res(self):
        """per-level side length, or None when any level is not cubic"""
        rs = np.array(self.meta.level_res_multidim)
        return rs[:, 0].tolist() if (rs == rs[:, :1]).all() else None

    def _bds(self, input, bidx, input_batched):
        if not input_batched:
            return 0
        assert bidx is None, 'bidx is only taken care of when input is not batched.'
        return prod(input.shape[1:-1])

    @profile
    def forward(self, input, params, bidx=None, batch_offsets=None, input_batched=False, max_level=None):
        return LoTDFunction.apply(self.meta, input, params.to(self.dtype), bidx, batch_offsets,
                                  self._bds(input, bidx, input_batched), self.loss_scale, max_level)

    @profile
    def forward_dydx(self, input, params, bidx=None, batch_offsets=None, input_batched=False, max_level=None,
                     need_dL_dinput=None):
        if need_dL_dinput is None:
            need_dL_dinput = torch.is_grad_enabled() and input.requires_grad
        return LoTDFunctionFwdDydx.apply(self.meta, input, params.to(self.dtype), bidx, batch_offsets,
                                         self._bds(input, bidx, input_batched), self.loss_scale, max_level,
                                         need_dL_dinput)

    @profile
    def backward_dydx(self, dL_dy, dy_dx, input, params, bidx=None, batch_offsets=None, input_batched=False,
                      max_level=None, grad_guard=None):
        return LoTDFunctionBwdDydx.apply(self.meta, dL_dy, input, params.to(self.dtype), dy_dx, bidx, batch_offsets,
                                         self._bds(input, bidx, input_batched), self.loss_scale, max_level, grad_guard)

    def forward_decoded(self, input, params, decoder, out_cols=None, max_level=None):
        """``decoder(self(input, params))[..., :out_cols]`` WITHOUT gradients -- what a density query under no_grad computes
        (nerf_ray_query.py:105-127: query_density on every marched sample, to prune).  When the pair is inside the fused kernel's range
        (``bindings._lotd.lod_mlp_fwd_ok``: 3-D Dense / Hash levels of 2 features, <= 32 encoded dims, an fp32 ``models.blocks.MLP`` of
        hidden width <= 64 with <= 32 outputs) ONE kernel encodes, decodes and stores only the ``out_cols`` requested columns: the
        [N, n_encoded_dims] features never reach memory.  Otherwise: the two ops, sliced.  No reference counterpart (the reference runs
        the encoder and the decoder as separate autograd ops); no autograd here -- a caller that needs gradients uses ``forward``."""
        n_out = decoder.out_features
        out_cols = n_out if out_cols is None else int(out_cols)
        desc = decoder.fused_desc() if (hasattr(decoder, "fused_desc") and getattr(decoder, "dtype", None) in (None, torch.float32)) else None
        if (FUSE_DECODED and desc is not None and input.is_cuda and input.dtype == torch.float32 and max_level is None
                and not torch.is_autocast_enabled() and _backend.lod_mlp_fwd_ok(self.meta, desc)):
            from nr3d_lib_amd.bindings import _mlp
            with torch.no_grad():
                prefix, xc, _ = _prep(input, None)
                ws = [l.weight for l in decoder.layers]
                bs = [l.bias for l in decoder.layers]
                packed = _mlp.pack(desc, ws, bs, with_backward=False)
                out = _backend.lod_mlp_fwd(self.meta, xc.flatten(0, -2), params.to(self.dtype), desc, packed, out_cols)
            return out.unflatten(0, prefix)
        with torch.no_grad():
            feat = self.forward(input, params, max_level=max_level)
            ddt = getattr(decoder, "dtype", None) or torch.float32
            feat = feat if feat.dtype == ddt else feat.to(ddt)
            if hasattr(decoder, "forward_columns"):
                return decoder.forward_columns(feat, out_cols)      # only those columns leave the decoder's kernel
            h = decoder(feat)
        return h[..., :out_cols]

    def __getstate__(self):
        return self.params

    def __setstate__(self, state):
        self.__init__(**state)

    def extra_repr(self):
        m = self.meta
        esize = {torch.float32: 4, torch.float16: 2}[self.dtype]
        n = np.array(m.level_n_params, dtype=np.float64)
        head = (f"in_dim={m.n_dims_to_encode}, out_dim={m.n_encoded_dims}, num_levels={m.n_levels}, "
                f"num_params={m.n_params}, params_size={m.n_params * esize / 1024 ** 2:.3f} MiB, dtype={self.dtype}")
        rows = [("lod_res_multidim", m.level_res_multidim), ("lod_n_feats", m.level_n_feats),
                ("lod_types", m.level_types_str), ("lod_n_params", m.level_n_params),
                ("lod_n_params_cumsum_ratio", np.round(np.cumsum(n) / max(n.sum(), 1), 3).tolist())]
        return head + "\n" + "\n".join(f"{k}={v}" for k, v in rows)
