"""Dense layer + activation lookup -- counterpart of the parts of nr3d_lib/models/layers.py the MLP blocks use
(``get_nonlinearity`` :185-225 for the torch-native activations, ``DenseLayer`` :228-300)."""
import math
from collections import namedtuple
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

__all__ = ['get_nonlinearity', 'DenseLayer']

_NL = namedtuple('namedtuple_nl_gain_init_firstinit', 'nl gain init first_init')
_nop = lambda *a, **k: None

_ACTIVATIONS = dict(relu=lambda **p: nn.ReLU(inplace=True, **p), elu=lambda **p: nn.ELU(inplace=True, **p),
                    selu=lambda **p: nn.SELU(inplace=True, **p), leaky_relu=lambda **p: nn.LeakyReLU(inplace=True, **p),
                    softplus=nn.Softplus, sigmoid=nn.Sigmoid, tanh=nn.Tanh)


def get_nonlinearity(config: Optional[Union[str, dict]]):
    """name / {'type': name, **params} / None -> (module | None, gain, init_fn, first_init_fn)"""
    if config is None or (isinstance(config, str) and config.lower() == 'none'):
        return _NL(None, 1, _nop, _nop)
    if isinstance(config, str):
        config = dict(type=config.lower())
    if not isinstance(config, dict):
        raise RuntimeError(f"Invalid nonlinearity={config}")
    assert 'type' in config, 'You should provide the type of the nonlinearity'
    name = config['type'].lower()
    param = {k: v for k, v in config.items() if k not in ('type', 'gain')}
    if name not in _ACTIVATIONS:
        raise NotImplementedError(f"nr3d_lib_amd: nonlinearity {name!r} (siren / trunc_* / cliptanh live outside the path)")
    nl = _ACTIVATIONS[name](**param)
    gain = config.get('gain')
    if gain is None:
        if name == 'leaky_relu':
            gain = init.calculate_gain(name, nl.negative_slope)
        elif name == 'softplus':
            gain = math.sqrt(2) if nl.beta >= 5.0 else 1.
        else:
            gain = init.calculate_gain(name)
    nl.name = name
    return _NL(nl, gain, _nop, _nop)


def _as_dtype(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    return getattr(torch, str(dtype).split('.')[-1])


class DenseLayer(nn.Module):
    """``y = act((x W^T) g_w + b g_b)`` with fp32 parameters; ``dtype`` only selects the autocast type of the forward.

    Behaviour of the reference's ``DenseLayer`` (nr3d_lib/models/layers.py:228-310), written from its contract:

    * default initialisation is ``nn.Linear``'s -- ``W ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` drawn first, then
      ``b ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` -- with ``W`` scaled by ``weight_init / lr_mul``, the bias bound by
      ``bias_init / lr_mul``, and both multiplied by ``lr_mul`` again at run time (so ``lr_mul`` only changes the
      effective learning rate);
    * ``equal_lr`` (StyleGAN-style): ``W ~ U(-weight_init/lr_mul, +weight_init/lr_mul)``, run-time gain
      ``lr_mul / sqrt(in_features)``;
    * ``forward(x, max_channel=c)`` uses the first ``c`` input columns of ``W`` (progressive input widths) and -- like
      the reference -- cuts the bias with the same bound."""

    def __init__(self, in_features: int, out_features: int, *, bias: bool = True,
                 activation: Union[str, dict, nn.Module] = None, should_init=True, equal_lr=False, lr_mul: float = 1,
                 weight_init: float = 1, bias_init: float = 1, dtype: Union[str, torch.dtype] = torch.float, device=None):
        super().__init__()
        self.in_features, self.out_features, self.equal_lr = in_features, out_features, equal_lr
        self.dtype = _as_dtype(dtype)
        self.activation = get_nonlinearity(activation).nl if isinstance(activation, (str, dict)) else activation
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=torch.float32, device=device))
        self.bias = nn.Parameter(torch.empty(out_features, dtype=torch.float32, device=device)) if bias else None
        self.weight_gain, self.bias_gain = 1, 1
        if should_init:
            self.reset_parameters(lr_mul=lr_mul, weight_init=weight_init, bias_init=bias_init)

    @torch.no_grad()
    def reset_parameters(self, lr_mul: float = 1, weight_init: float = 1, bias_init: float = 1):
        fan_in = self.in_features
        if self.equal_lr:
            self.weight.uniform_(-weight_init / lr_mul, weight_init / lr_mul)
            self.weight_gain = lr_mul / np.sqrt(fan_in)
        else:
            # nn.Linear: kaiming-uniform with a = sqrt(5), i.e. bound = sqrt(3) * sqrt(2 / (1 + 5)) / sqrt(fan_in)
            bound = math.sqrt(3.0) * (math.sqrt(2.0 / 6.0) / math.sqrt(fan_in)) if fan_in > 0 else 0.0
            self.weight.uniform_(-bound, bound).mul_(weight_init / lr_mul)
            self.weight_gain = lr_mul
        if self.bias is not None:
            b = (1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0) * (bias_init / lr_mul)
            self.bias.uniform_(-b, b)
            self.bias_gain = lr_mul

    @property
    def device(self) -> torch.device:
        return self.weight.device

    def get_weight_reg(self, norm_type: float = 2.0):
        return torch.stack([p.norm(p=norm_type) for p in self.parameters()])

    def _effective(self, max_channel):
        w, b = self.weight, self.bias
        if max_channel is not None:
            w = w[:, :max_channel]
            b = None if b is None else b[:max_channel]
        if self.weight_gain != 1:
            w = w * self.weight_gain
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        return w, b

    def forward(self, x: torch.Tensor, max_channel: int = None):
        low_precision = self.dtype in (torch.float16, torch.bfloat16)
        with torch.autocast(device_type='cuda', dtype=self.dtype, enabled=low_precision):
            y = F.linear(x, *self._effective(max_channel))
            return y if self.activation is None else self.activation(y)

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"equal_lr={self.equal_lr}")
