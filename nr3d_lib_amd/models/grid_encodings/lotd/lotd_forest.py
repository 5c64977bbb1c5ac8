"""LoTD over a forest of blocks -- counterpart of nr3d_lib/models/grid_encodings/lotd/lotd_forest.py
(functional wrappers :29-74, LoTDForestEncoding :76-285).

One LoTD parameter set per block, stored as ``forest_flattened_params`` [n_trees, n_params] (fp32); points are given in
their block's [-1,1]^3 cube together with the block index (or batched per block).  With continuity enabled a level of
resolution R interpolates on R+2 nodes per dim, the outer ones read from the neighbouring blocks, so the encoding is
continuous across block faces.  The autograd functions are the plain LoTD ones fed with ``metas = (lod_meta,
forest_meta)``; ``bindings._lotd`` dispatches tuples to the forest kernels."""
from math import prod, sqrt
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from nr3d_lib_amd.models.spatial import ForestBlockSpace
from .lotd import LoDType, LoTD, LoTDFunction, LoTDFunctionBwdDydx, LoTDFunctionFwdDydx

__all__ = ['lotd_forest_encoding', 'lotd_forest_fwd_dydx', 'lotd_forest_bwd_dydx', 'LoTDForestEncoding']


def _mode(metas, input, params, block_inds, input_batched, loss_scale):
    assert isinstance(metas, tuple) and len(metas) == 2, "`metas` should be a tuple of (lod_meta, forest_meta)"
    if input_batched:
        batch_data_size, block_inds = prod(input.shape[1:-1]), None
    else:
        batch_data_size = 0
    if loss_scale is None:
        loss_scale = 128.0 if params.dtype == torch.float16 else 1.
    return block_inds, batch_data_size, loss_scale


def lotd_forest_encoding(metas, input: torch.Tensor, params: torch.Tensor, block_inds: torch.Tensor = None,
                         block_offsets: torch.Tensor = None, input_batched=False, loss_scale: float = None,
                         max_level: int = None) -> torch.Tensor:
    block_inds, bds, loss_scale = _mode(metas, input, params, block_inds, input_batched, loss_scale)
    return LoTDFunction.apply(metas, input, params, block_inds, block_offsets, bds, loss_scale, max_level)


def lotd_forest_fwd_dydx(metas, input: torch.Tensor, params: torch.Tensor, block_inds: torch.Tensor = None,
                         block_offsets: torch.Tensor = None, input_batched=False, loss_scale: float = None,
                         max_level: int = None, need_dL_dinput: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    if need_dL_dinput is None:
        need_dL_dinput = torch.is_grad_enabled() and input.requires_grad
    block_inds, bds, loss_scale = _mode(metas, input, params, block_inds, input_batched, loss_scale)
    return LoTDFunctionFwdDydx.apply(metas, input, params, block_inds, block_offsets, bds, loss_scale, max_level, need_dL_dinput)


def lotd_forest_bwd_dydx(metas, dL_dy: torch.Tensor, dy_dx: torch.Tensor, input: torch.Tensor, params: torch.Tensor,
                         block_inds: torch.Tensor = None, block_offsets: torch.Tensor = None, input_batched=False,
                         loss_scale: float = None, max_level: int = None, grad_guard=None) -> torch.Tensor:
    block_inds, bds, loss_scale = _mode(metas, input, params, block_inds, input_batched, loss_scale)
    return LoTDFunctionBwdDydx.apply(metas, dL_dy, input, params, dy_dx, block_inds, block_offsets, bds, loss_scale,
                                     max_level, grad_guard)


class LoTDForestEncoding(nn.Module):
    def __init__(self, input_ch=3, *, lotd_cfg=dict(), anneal_cfg: dict = None,
                 param_init_cfg={'type': 'uniform_to_type', 'bound': 1.0e-4}, clip_level_grad_ema_factor: float = 0,
                 dtype=torch.half, device=None) -> None:
        super().__init__()
        self.dtype = dtype if isinstance(dtype, torch.dtype) else getattr(torch, str(dtype).replace('torch.', ''))
        self.loss_scale = 128.0 if self.dtype == torch.float16 else 1.0
        self.space = ForestBlockSpace(dtype=torch.float, device=device)           # the valid representing space
        self.lotd = LoTD(input_ch, **lotd_cfg, dtype=self.dtype, device=device)
        self.in_features, self.out_features = input_ch, self.lotd.out_features
        self.register_parameter("forest_flattened_params", None)
        self.clip_level_grad_ema_factor = clip_level_grad_ema_factor
        self.param_init_cfg = param_init_cfg
        self.annealer = None                      # coarse-to-fine schedule (lotd_forest.py:101-105); set_anneal_iter()
        if anneal_cfg is not None:
            from ..multires_annealer import MultiresAnnealer
            self.annealer = MultiresAnnealer(self.lotd.level_n_feats, **anneal_cfg, dtype=self.dtype, device=device)
        self.window: torch.Tensor = None          # optional soft mask on the output features
        self.max_level: int = None                # levels above it are skipped (-1: all of them)
        if clip_level_grad_ema_factor > 0:
            self.register_buffer("level_grad_norm_ema", torch.full([self.lotd.n_levels], 0.1, dtype=torch.float, device=device))
        self._register_load_state_dict_pre_hook(self._before_load_state_dict)

    def _before_load_state_dict(self, state_dict, prefix, *unused):
        p = state_dict[prefix + 'forest_flattened_params']
        cur = self.forest_flattened_params
        if cur is None or list(cur.shape) != list(p.shape):
            self.forest_flattened_params = nn.Parameter(torch.zeros(p.shape, dtype=p.dtype, device=self.device))

    device = property(lambda self: self.space.device)
    lod_meta = property(lambda self: self.lotd.meta)
    forest_meta = property(lambda self: self.space.meta)
    metas = property(lambda self: (self.lotd.meta, self.space.meta))
    active_forest_params = property(lambda self: self.forest_flattened_params)

    def populate(self, **kwargs):
        self.space.populate(**kwargs)
        self._populate_params()

    def _populate_params(self):
        """[n_trees, n_params] fp32, initialised per level type like the single-block encoder (lotd_forest.py:133-187)"""
        p = torch.zeros([self.forest_meta.n_trees, self.lod_meta.n_params], dtype=torch.float, device=self.device)
        self.forest_flattened_params = nn.Parameter(p, requires_grad=True)
        cfg, kind = self.param_init_cfg, self.param_init_cfg['type']
        with torch.no_grad():
            if kind == 'uniform':
                p.uniform_(-cfg['bound'], cfg['bound'])
            elif kind == 'normal':
                p.normal_(0., cfg['std'])
            elif kind in ('uniform_to_type', 'normal_to_type'):
                uniform = kind == 'uniform_to_type'
                base = cfg['bound'] if uniform else cfg['std']
                for l, tp in enumerate(self.lotd.level_types):
                    tp = LoDType(tp)
                    if tp in (LoDType.Dense, LoDType.Hash):
                        s = base
                    elif tp == LoDType.VectorMatrix:
                        s = sqrt(base)
                    elif tp == LoDType.NPlaneSum:
                        s = base / 3. if uniform else base
                    elif tp in (LoDType.NPlaneMul, LoDType.CP, LoDType.CPfast):
                        s = base ** (1 / 3.)
                    else:
                        raise RuntimeError(f"Invalid tp={tp}")
                    lvl = self.get_level_param(slice(None), l)
                    lvl.uniform_(-s, s) if uniform else lvl.normal_(0., s)
            else:
                raise RuntimeError(f"Invalid param_init_method={kind}")

    def set_anneal_iter(self, cur_it: int):
        if self.annealer is not None:
            self.max_level, self.window = self.annealer(cur_it)

    def _params(self):
        return self.active_forest_params.flatten().to(self.dtype)

    def _mask(self, out):
        return out if self.window is None else out * self.window

    def forward(self, block_x: torch.Tensor, block_inds: torch.Tensor = None, block_offsets: torch.Tensor = None,
                max_level: int = None):
        """block_x in [-1,1]^3 inside block `block_inds` (None: batched, block_x [n_trees, ..., 3])"""
        out = lotd_forest_encoding(self.metas, block_x / 2. + 0.5, self._params(), block_inds, block_offsets,
                                   input_batched=block_inds is None, max_level=(max_level or self.max_level),
                                   loss_scale=self.loss_scale)
        return self._mask(out)

    def forward_dydx(self, block_x: torch.Tensor, block_inds: torch.Tensor = None, block_offsets: torch.Tensor = None,
                     max_level: int = None, need_dL_dinput: Optional[bool] = None):
        out, dy_dx = lotd_forest_fwd_dydx(self.metas, block_x / 2. + 0.5, self._params(), block_inds, block_offsets,
                                          input_batched=block_inds is None, max_level=(max_level or self.max_level),
                                          loss_scale=self.loss_scale, need_dL_dinput=need_dL_dinput)
        return self._mask(out), dy_dx

    def backward_dydx(self, dL_dy: torch.Tensor, dy_dx: torch.Tensor, block_x: torch.Tensor, block_inds: torch.Tensor = None,
                      block_offsets: torch.Tensor = None, max_level: int = None):
        nablas = lotd_forest_bwd_dydx(self.metas, dL_dy, dy_dx, block_x / 2. + 0.5, self._params(), block_inds,
                                      block_offsets, input_batched=block_inds is None,
                                      max_level=(max_level or self.max_level), loss_scale=self.loss_scale)
        return nablas / 2.

    # ---- parameter access: whole levels (and Dense 'vol' views), for any selection of blocks ----------------------------
    def _level_view(self, l: int, op: str = None, dim: int = None):
        from .lotd_helpers import level_param_index_shape
        index, shape = level_param_index_shape(self.lod_meta, l, op, dim)
        return index[0], shape

    def get_level_param(self, bid: Union[int, List[int], slice, torch.Tensor], l: int, op: str = None, dim: int = None, grad=False):
        sl, shape = self._level_view(l, op, dim)
        src = self.forest_flattened_params.grad if grad else self.forest_flattened_params
        sel = src.data[bid, sl] if not grad else src[bid, sl]
        return sel.view(*sel.shape[:-1], *shape)

    def set_level_param(self, bid, l: int, op: str = None, dim: int = None, value: torch.Tensor = None):
        sl, shape = self._level_view(l, op, dim)
        with torch.no_grad():
            tgt = self.forest_flattened_params[bid, sl]
            self.forest_flattened_params[bid, sl] = value.contiguous().reshape(*tgt.shape[:-1], prod(shape))
