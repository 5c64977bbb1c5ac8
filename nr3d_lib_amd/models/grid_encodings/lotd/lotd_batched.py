"""``LoTDBatched``: one LoTD table set per batch entry, produced from latents by a "grower" network, queried either
with batched inputs ``[B, ..., D]`` or with flat inputs + a per-point batch index.

Counterpart of nr3d_lib/models/grid_encodings/lotd/lotd_batched.py:29-148 for the hot path: ``grow(z)`` / ``clear()``,
``forward`` / ``forward_dydx`` / ``backward_dydx`` on inputs in [-1, 1] (nablas halved), ``max_level`` / ``window``.
The reference builds its grower from ``grower_cfg`` (hyper-network zoo, out of scope); here any module mapping
latents ``[B, z_dim]`` to ``[B, n_params]`` is passed in together with the level layout."""
from typing import Optional

import torch
import torch.nn as nn

from .lotd import LoTD

__all__ = ['LoTDBatched']


class LoTDBatched(nn.Module):
    def __init__(self, input_ch, *, lotd_cfg: dict, grower: nn.Module, space: nn.Module = None, device=None,
                 dtype=torch.float) -> None:
        super().__init__()
        self.dtype = dtype
        self.space = space
        self.lotd_grower = grower
        self.lotd = LoTD(input_ch, **lotd_cfg, dtype=dtype, device=device)
        self.z_dim = getattr(grower, 'z_dim', None)
        self.in_features = input_ch
        self.out_features = self.lotd.out_features
        self.annealer = None
        self.window: Optional[torch.Tensor] = None
        self.max_level: Optional[int] = None

    lod_meta = property(lambda self: self.lotd.meta)

    @property
    def device(self) -> torch.device:
        return next(self.lotd_grower.parameters()).device

    def grow(self, z: torch.Tensor, max_level: int = None):
        """run the grower: ``lod_params`` [B * n_params] stays attached to its graph (gradients reach the grower)"""
        z = (z.unsqueeze(0) if z.dim() == 1 else z).to(device=self.device, dtype=torch.float)
        assert z.dim() == 2, "latent input must be 2D"
        self.B, self.z_per_batch = z.shape[0], z
        grown = self.lotd_grower(z)
        assert grown.shape == (self.B, self.lotd.n_params), \
            f"grower must return [B, n_params] = [{self.B}, {self.lotd.n_params}], got {list(grown.shape)}"
        self.lod_params = grown.flatten().to(self.dtype)

    def clear(self):
        for name in ('lod_params', 'z_per_batch'):
            if hasattr(self, name):
                delattr(self, name)

    def _route(self, input, bidx):
        """(bidx, input_batched) for the LoTD call: batched input [B, ...] or flat input with per-point bidx"""
        if bidx is None:
            assert input.shape[0] == self.B, f"input should have a batch size of {self.B}"
            return None, True
        assert list(bidx.shape) == list(input.shape[:-1]), "bidx and input does not match."
        return bidx, False

    def forward(self, input: torch.Tensor, bidx: torch.Tensor = None, max_level: int = None):
        bidx, batched = self._route(input, bidx)
        out = self.lotd(input / 2. + 0.5, self.lod_params, bidx, input_batched=batched,
                        max_level=(max_level or self.max_level))
        return out if self.window is None else out * self.window

    def forward_dydx(self, input: torch.Tensor, bidx: torch.Tensor = None, max_level: int = None,
                     need_dL_dinput: Optional[bool] = None):
        bidx, batched = self._route(input, bidx)
        return self.lotd.forward_dydx(input / 2. + 0.5, self.lod_params, bidx, input_batched=batched,
                                      max_level=(max_level or self.max_level), need_dL_dinput=need_dL_dinput)

    def backward_dydx(self, dL_dy: torch.Tensor, dy_dx: torch.Tensor, input: torch.Tensor, bidx: torch.Tensor = None,
                      max_level: int = None):
        bidx, batched = self._route(input, bidx)
        nablas = self.lotd.backward_dydx(dL_dy, dy_dx, input / 2. + 0.5, self.lod_params, bidx, input_batched=batched,
                                         max_level=(max_level or self.max_level))
        return nablas / 2.
