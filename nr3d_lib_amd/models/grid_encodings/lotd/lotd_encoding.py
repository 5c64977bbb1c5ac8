"""``LoTDEncoding``: the module users put in a field -- owns the fp32 ``flattened_params`` (its checkpoint format) and
feeds them to the HIP-backed ``LoTD`` functions.

Counterpart of nr3d_lib/models/grid_encodings/lotd/lotd_encoding.py:37-326 for the hot path: constructor keywords,
``forward`` / ``forward_dydx`` / ``backward_dydx`` on inputs in [-1, 1] (mapped to [0, 1]; nablas halved), the four
``param_init_cfg`` schemes, ``max_level`` / ``window`` masking, ``get_level_param`` / ``set_level_param`` for whole
levels (and Dense volumes), ``inference_param``, and the ``lotd_cfg`` extra state.  Out of scope here (they raise):
``space_cfg`` (space classes), ``anneal_cfg`` (annealer), sub-level ``op`` slices of VM/CP tables, ``rescale_volume``."""
from math import sqrt
from typing import Any, Optional, Tuple

import torch
import torch.nn as nn

from .lotd import LoDType, LoTD
from .lotd_cfg import get_lotd_cfg

__all__ = ['LoTDEncoding']


def _as_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    return {'half': torch.half, 'float16': torch.half, 'float': torch.float, 'float32': torch.float}[str(dtype)]


class LoTDEncoding(nn.Module):
    def __init__(self, input_ch=3, *, lotd_cfg: dict = None, lotd_auto_compute_cfg: dict = None, lotd_use_cuboid=False,
                 space: nn.Module = None, space_cfg: dict = None, anneal_cfg: dict = None,
                 param_init_cfg={'type': 'uniform_to_type', 'bound': 1.0e-4}, clip_level_grad_ema_factor: float = 0,
                 dtype=torch.half, device=None) -> None:
        super().__init__()
        if space_cfg is not None:
            raise NotImplementedError("nr3d_lib_amd: `space_cfg` needs the space classes (not on the hot path); "
                                      "pass a ready `space` module or normalise the inputs yourself")
        if anneal_cfg is not None:
            raise NotImplementedError("nr3d_lib_amd: the multires annealer is not on the hot path; set "
                                      "`max_level` / `window` on the module instead")
        assert (lotd_cfg is not None) != (lotd_auto_compute_cfg is not None), \
            "Please specify one and only one of `lotd_cfg` and `lotd_auto_compute_cfg`"
        self.dtype = _as_dtype(dtype)
        self.space = space
        self.clip_level_grad_ema_factor = clip_level_grad_ema_factor
        self.param_init_cfg = param_init_cfg
        if lotd_auto_compute_cfg is not None:
            stretch = 1 if not lotd_use_cuboid else (self.space.radius3d * 2).tolist()
            lotd_cfg = get_lotd_cfg(**lotd_auto_compute_cfg, input_ch=input_ch, stretch=stretch)
        self.lotd_cfg = lotd_cfg if isinstance(lotd_cfg, dict) else lotd_cfg.to_dict()
        self.lotd = LoTD(input_ch, **self.lotd_cfg, dtype=self.dtype, device=device)
        self.in_features: int = input_ch
        self.out_features: int = self.lotd.out_features
        # parameters are always stored in fp32; `dtype` only decides what the kernels are fed
        self.flattened_params = nn.Parameter(torch.zeros(self.lotd.n_params, device=device, dtype=torch.float))
        self.init_param_random()
        self.annealer = None
        self.window: Optional[torch.Tensor] = None      # optional soft mask on the output features
        self.max_level: Optional[int] = None            # levels above it are skipped (-1: all of them)
        if clip_level_grad_ema_factor > 0:
            self.register_buffer("level_grad_norm_ema", torch.full([self.lotd.n_levels], 0.1, device=device))

    device = property(lambda self: self.flattened_params.device)
    level_n_feats = property(lambda self: self.lotd.level_n_feats)
    meta = property(lambda self: self.lotd.meta)
    lod_meta = property(lambda self: self.lotd.meta)
    inference_param = property(lambda self: self.flattened_params.data.to(self.dtype))

    def set_anneal_iter(self, cur_it: int):
        if self.annealer is not None:
            self.max_level, self.window = self.annealer(cur_it)

    def _lvl(self, max_level):
        return max_level or self.max_level        # same precedence (and same `0 -> fall back`) as the reference

    def _mask(self, out):
        return out if self.window is None else out * self.window

    def forward(self, input: torch.Tensor, max_level: int = None) -> torch.Tensor:
        """features at positions ``input`` in [-1, 1]^D -> [..., out_features]"""
        return self._mask(self.lotd.forward(input / 2. + 0.5, self.flattened_params, max_level=self._lvl(max_level)))

    def forward_dydx(self, input: torch.Tensor, max_level: int = None,
                     need_dL_dinput: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """features and the stored Jacobian dy/dx (w.r.t. the [0, 1] coordinates) for ``backward_dydx``"""
        out, dy_dx = self.lotd.forward_dydx(input / 2. + 0.5, self.flattened_params, max_level=self._lvl(max_level),
                                            need_dL_dinput=need_dL_dinput)
        return self._mask(out), dy_dx

    def backward_dydx(self, dL_dy: torch.Tensor, dy_dx: torch.Tensor, input: torch.Tensor, max_level: int = None,
                      grad_guard=None) -> torch.Tensor:
        """nablas dL/d(input) from dL/dy and the Jacobian of ``forward_dydx`` (differentiable once more: second order)"""
        nablas = self.lotd.backward_dydx(dL_dy, dy_dx, input / 2. + 0.5, self.flattened_params,
                                         max_level=self._lvl(max_level), grad_guard=grad_guard)
        return nablas / 2.                         # d(input / 2 + 0.5) / d(input)

    # ---- parameter access ---------------------------------------------------------------------------------------
    def _level_slice(self, l: int, op: str = None, dim: int = None):
        m = self.lod_meta
        assert 0 <= l < m.n_levels
        sl = slice(m.level_offsets[l], m.level_offsets[l] + m.level_n_params[l])
        if op is None:
            return sl, (m.level_sizes[l], m.level_n_feats[l])
        if op == 'vol' and LoDType(int(m.level_types[l])) == LoDType.Dense:
            return sl, (*m.level_res_multidim[l], m.level_n_feats[l])
        raise NotImplementedError(f"nr3d_lib_amd: get/set_level_param(op={op!r}, dim={dim!r}) -- only whole levels and "
                                  f"Dense 'vol' views are provided")

    def get_level_param(self, l: int, op: str = None, dim: int = None, grad=False) -> torch.Tensor:
        sl, shape = self._level_slice(l, op, dim)
        return (self.flattened_params.grad if grad else self.flattened_params)[sl].view(shape)

    def set_level_param(self, l: int, op: str = None, dim: int = None, value: torch.Tensor = ...):
        sl, _ = self._level_slice(l, op, dim)
        with torch.no_grad():
            self.flattened_params[sl] = value.contiguous().reshape(-1)

    @torch.no_grad()
    def init_param_random(self):
        """'uniform' / 'normal' over everything, or per level type ('*_to_type'): product-type tables are scaled so
        that the PRODUCT of their factors has the requested magnitude (lotd_encoding.py:244-288)"""
        cfg = self.param_init_cfg
        kind = cfg['type']
        if kind == 'uniform':
            self.flattened_params.uniform_(-cfg['bound'], cfg['bound'])
        elif kind == 'normal':
            self.flattened_params.normal_(0, cfg['std'])
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
                    s = base if uniform else base / 3.
                elif tp in (LoDType.NPlaneMul, LoDType.CP, LoDType.CPfast):
                    s = base ** (1 / 3.)
                else:
                    raise RuntimeError(f"Invalid tp={tp}")
                p = self.get_level_param(l)
                p.uniform_(-s, s) if uniform else p.normal_(0., s)
        else:
            raise RuntimeError(f"Invalid param_init_method={kind}")

    # the layout description travels with the checkpoint (lotd_encoding.py:321-326)
    def get_extra_state(self) -> Any:
        return self.lotd_cfg

    def set_extra_state(self, state: Any):
        self.lotd_cfg = state

    def rescale_volume(self, new_aabb: torch.Tensor):
        raise NotImplementedError("nr3d_lib_amd: rescale_volume (table re-interpolation) is model tooling, not hot path")
