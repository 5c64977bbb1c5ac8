"""``LoTDEncoding``: the module users put in a field -- owns the fp32 ``flattened_params`` (its checkpoint format) and
feeds them to the HIP-backed ``LoTD`` functions.

Counterpart of nr3d_lib/models/grid_encodings/lotd/lotd_encoding.py:37-326 for the hot path: constructor keywords,
``forward`` / ``forward_dydx`` / ``backward_dydx`` on inputs in [-1, 1] (mapped to [0, 1]; nablas halved), the four
``param_init_cfg`` schemes, ``max_level`` / ``window`` masking driven by ``anneal_cfg`` (``MultiresAnnealer``,
``set_anneal_iter``), ``space_cfg`` of type 'aabb' / 'batched' / 'unbounded', ``get_level_param`` / ``set_level_param`` on whole levels
and on every line / plane / volume table (``lotd_helpers``), ``rescale_volume``, ``inference_param``, the ``lodN`` / ``lodN_vec`` / ``lodN_matK``
attribute views, ``clip_grad_and_update_ema``, ``stat_param``, and the ``lotd_cfg`` extra state.  Not provided:
``init_param_from_net``."""
import re
from math import sqrt
from typing import Any, Dict, Optional, Tuple

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
        if space is None and space_cfg is not None:              # lotd_encoding.py:59-75
            space_cfg = dict(space_cfg)
            space_type = space_cfg.pop('type').lower()
            if space_type == 'aabb':
                from nr3d_lib_amd.models.spatial import AABBSpace
                space = AABBSpace(**space_cfg)
            elif space_type in ('unbounded', 'none'):
                space = None
            elif space_type == 'batched':
                from nr3d_lib_amd.models.spatial import BatchedBlockSpace
                space = BatchedBlockSpace(**space_cfg)
            else:
                raise RuntimeError(f"Invalid space_type={space_type}")
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
        # coarse-to-fine schedule of `max_level` / `window` (lotd_encoding.py:99-103); set_anneal_iter() advances it
        self.annealer = None
        if anneal_cfg is not None:
            from ..multires_annealer import MultiresAnnealer
            self.annealer = MultiresAnnealer(self.lotd.level_n_feats, **anneal_cfg, dtype=self.dtype, device=device)
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
        """(slice, shape) of a level or of one of its line / plane / volume tables (lotd_helpers.level_param_index_shape)"""
        from .lotd_helpers import level_param_index_shape
        index, shape = level_param_index_shape(self.lod_meta, l, op, dim)
        return index[0], shape

    def get_level_param(self, l: int, op: str = None, dim: int = None, grad=False) -> torch.Tensor:
        sl, shape = self._level_slice(l, op, dim)
        return (self.flattened_params.grad if grad else self.flattened_params)[sl].view(shape)

    def set_level_param(self, l: int, op: str = None, dim: int = None, value: torch.Tensor = ...):
        sl, _ = self._level_slice(l, op, dim)
        with torch.no_grad():
            self.flattened_params[sl] = value.contiguous().reshape(-1)

    # ``enc.lod3`` / ``enc.lod3_vec`` / ``enc.lod3_mat1`` read (and ``enc.lod3 = t`` writes) the parameters of a level or of
    # one of its line / plane tables: views into ``flattened_params`` (reference lotd_encoding.py:121, :291-320)
    _LOD_ATTR = re.compile(r"^lod(?P<level>[0-9]+)(_(?P<op>[a-z]+)(?P<dim>[0-9]+)?)?$")

    def _lod_attr(self, name: str):
        m = self._LOD_ATTR.match(name) if name.startswith('lod') else None
        if m is None or 'flattened_params' not in self._parameters:
            return None
        dim = m.group('dim')
        return int(m.group('level')), m.group('op') or None, None if dim in (None, '') else int(dim)

    def __getattr__(self, name: str):
        key = self._lod_attr(name)
        if key is not None:
            return self.get_level_param(*key)
        return super().__getattr__(name)

    def __setattr__(self, name: str, value) -> None:
        key = self._lod_attr(name) if isinstance(value, torch.Tensor) else None
        if key is not None:
            self.set_level_param(*key, value=value)
        else:
            super().__setattr__(name, value)

    @torch.no_grad()
    def clip_grad_and_update_ema(self, val: float = None):
        """per-level gradient-norm clipping against a running norm (reference lotd_encoding.py:471-486): the EMA of every
        level's gradient 2-norm moves 1 % towards the current norm, then the level's gradient is rescaled so that its
        norm does not exceed ``clip_level_grad_ema_factor`` x that EMA (``clip_grad_norm_``'s rule).  No-op unless the
        encoder was built with ``clip_level_grad_ema_factor > 0``; ``val`` is accepted and ignored like in the reference."""
        if not self.clip_level_grad_ema_factor > 0 or self.flattened_params.grad is None:
            return
        L = self.lotd.n_levels
        gnorm = torch.stack([self.get_level_param(l, grad=True).norm() for l in range(L)])
        ema = self.level_grad_norm_ema.copy_(gnorm.lerp(self.level_grad_norm_ema, 0.99))
        for l in range(L):
            g = self.get_level_param(l, grad=True)
            max_norm = self.clip_level_grad_ema_factor * ema[l]
            g.mul_((max_norm / (gnorm[l] + 1e-6)).clamp(max=1.0))

    @torch.no_grad()
    def stat_param(self, with_grad: bool = False, prefix: str = '') -> Dict[str, float]:
        """mean / std / min / max / absmax / norm of all parameters and of every level (VM levels: lines and planes apart),
        optionally of their gradients, plus the gradient-norm EMAs -- the keys of the reference's logger
        (lotd_encoding.py:488-508, tensor_statistics utils.py:767-795)"""
        def stats(t: torch.Tensor, key: str):
            t = t.detach().float()
            if t.numel() == 1:
                return {f"{key}.val": t.item(), f"{key}.mean": t.item()}
            return {f"{key}.mean": t.mean().item(), f"{key}.std": t.std().item(), f"{key}.min": t.min().item(),
                    f"{key}.max": t.max().item(), f"{key}.absmax": t.abs().max().item(), f"{key}.norm": t.norm().item()}
        pre = prefix + ('.' if prefix and not prefix.endswith('.') else '')
        with_grad = with_grad and self.flattened_params.grad is not None
        out = stats(self.flattened_params, pre + 'total')
        if with_grad:
            out.update(stats(self.flattened_params.grad, pre + 'grad_total'))
        for l, tp in enumerate(self.lotd.level_types):
            parts = [('vec', '.vec'), ('mat', '.mat')] if LoDType(tp) == LoDType.VectorMatrix else [(None, '')]
            for op, tag in parts:
                out.update(stats(self.get_level_param(l, op), f"{pre}lv.{l}{tag}"))
                if with_grad:
                    out.update(stats(self.get_level_param(l, op, grad=True), f"{pre}grad.lv.{l}{tag}"))
        if self.clip_level_grad_ema_factor > 0:
            out.update({f"{pre}grad.lv.{l}.ema": self.level_grad_norm_ema[l].item() for l in range(self.lotd.n_levels)})
        return out

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

    @torch.no_grad()
    def rescale_volume(self, new_aabb: torch.Tensor):
        """Shrink the encoded space to ``new_aabb`` ([2, 3], inside the current one) and resample every table on its
        full resolution over the new box (lotd_encoding.py:329-401).  Cubic levels of the tensor types only: a Hash
        level has no spatial layout to resample.  Each table is sampled at the positions of its new vertices, expressed
        in the old box: a Dense volume in 3-D, the line tables of VM / CP levels along their own axis, the plane tables
        of VM / NPlane levels in the two axes they span (ascending, first one slowest).  As in the reference, the line
        and plane vertices are taken at the cell centres of the new box (``param_vertices(..., is_forest=True)``) while
        the old tables are sampled with the plain-level vertex convention."""
        from .lotd_helpers import param_interpolate, param_vertices
        m = self.lod_meta
        assert m.level_res[0] > 0, f"Expects equal resolution on each level! \nWhile current lod_res={m.level_res_multidim}"
        dev = self.flattened_params.device
        new_aabb = new_aabb.view(2, 3).to(dev, torch.float)
        old_aabb = self.space.aabb.view(2, 3).to(dev, torch.float)
        c_old, h_old = (old_aabb[1] + old_aabb[0]) / 2., (old_aabb[1] - old_aabb[0]) / 2.
        c_new, h_new = (new_aabb[1] + new_aabb[0]) / 2., (new_aabb[1] - new_aabb[0]) / 2.

        def to_old(v, axes):
            """normalised coordinates of the new box along ``axes`` -> normalised coordinates of the old box"""
            ax = torch.as_tensor(axes, device=dev)
            return (v * h_new[ax] + c_new[ax] - c_old[ax]) / h_old[ax]

        def lines(R, tables):                                   # [3, R, M]: table k runs along axis k
            v = param_vertices(R, 1, is_forest=True, device=dev).view(1, R, 1).expand(3, R, 1)
            x = torch.stack([to_old(v[k], [k]) for k in range(3)], 0)
            return param_interpolate(tables, x, R, False).contiguous()

        def planes(R, tables):                                  # [3, R, R, M]: table k spans the two axes other than k
            v = param_vertices(R, 2, is_forest=True, device=dev)
            x = torch.stack([to_old(v, [a for a in range(3) if a != k]) for k in range(3)], 0)
            return param_interpolate(tables, x, R, False).contiguous()

        for l, (R, kind) in enumerate(zip(m.level_res, m.level_types)):
            kind = LoDType(int(kind))
            if kind == LoDType.Dense:
                x = to_old(param_vertices(R, 3, False, device=dev, dtype=torch.float), [0, 1, 2])
                vol = self.get_level_param(l, 'vol')
                self.set_level_param(l, value=param_interpolate(vol.unsqueeze(0), x.unsqueeze(0), R, False).squeeze(0).contiguous())
            elif kind == LoDType.VectorMatrix:
                self.set_level_param(l, 'vec', value=lines(R, self.get_level_param(l, 'vec')))
                self.set_level_param(l, 'mat', value=planes(R, self.get_level_param(l, 'mat')))
            elif kind in (LoDType.NPlaneMul, LoDType.NPlaneSum):
                self.set_level_param(l, value=planes(R, self.get_level_param(l, 'plane')))
            elif kind in (LoDType.CP, LoDType.CPfast):
                self.set_level_param(l, value=lines(R, self.get_level_param(l, 'line')))
            elif kind == LoDType.Hash:
                raise RuntimeError("LoDType==hash does not support spatial operations.")
            else:
                raise RuntimeError(f"Invalid lod_type={kind}")
