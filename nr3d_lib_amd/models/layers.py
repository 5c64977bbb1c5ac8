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


class DenseLayer(nn.Module):
    def __init__(self, in_features: int, out_features: int, *, bias: bool = True,
                 activation: Union[str, dict, nn.Module] = None, should_init=True, equal_lr=False, lr_mul: float = 1,
                 weight_init: float = 1, bias_init: float = 1, dtype: Union[str, torch.dtype] = torch.float, device=None):
        super().__init__()
        self.dtype = dtype if isinstance(dtype, torch.dtype) or dtype is None else getattr(torch, str(dtype).replace('torch.', ''))
        self.in_features, self.out_features, self.equal_lr = in_features, out_features, equal_lr
        # parameters are always stored in fp32; `dtype` is only respected when forward
        self.weight = nn.Parameter(torch.empty((out_features, in_features), device=device, dtype=torch.float))
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=torch.float)) if bias else None
        if isinstance(activation, (str, dict)):
            activation = get_nonlinearity(activation).nl
        self.activation = activation
        self.weight_gain = self.bias_gain = 1
        if should_init:
            if equal_lr:
                bound = weight_init / lr_mul
                init.uniform_(self.weight, -bound, bound)
                self.weight_gain = lr_mul / np.sqrt(self.in_features)
            else:                                   # nn.Linear.reset_parameters()
                init.kaiming_uniform_(self.weight, a=math.sqrt(5))
                with torch.no_grad():
                    self.weight *= (weight_init / lr_mul)
                self.weight_gain = lr_mul
            if self.bias is not None:
                fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
                bound = (1 / math.sqrt(fan_in) if fan_in > 0 else 0) * (bias_init / lr_mul)
                init.uniform_(self.bias, -bound, bound)
                self.bias_gain = lr_mul

    @property
    def device(self) -> torch.device:
        return self.weight.device

    def get_weight_reg(self, norm_type: float = 2.0):
        return torch.stack([p.norm(p=norm_type) for n, p in self.named_parameters()])

    def forward(self, x: torch.Tensor, max_channel: int = None):
        with torch.autocast(device_type='cuda', dtype=self.dtype, enabled=self.dtype in (torch.float16, torch.bfloat16)):
            weight = self.weight[:, :max_channel] if max_channel is not None else self.weight
            bias = self.bias[:max_channel] if (max_channel is not None and self.bias is not None) else self.bias
            if self.weight_gain == 1 and (self.bias is None or self.bias_gain == 1):
                out = F.linear(x, weight, bias)
            else:
                out = F.linear(x, weight * self.weight_gain, None if bias is None else bias * self.bias_gain)
            return out if self.activation is None else self.activation(out)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, equal_lr={self.equal_lr}"
