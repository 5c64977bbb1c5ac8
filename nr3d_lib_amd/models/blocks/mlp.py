"""MLP blocks -- counterpart of ``MLP`` / ``FCBlock`` (nr3d_lib/models/blocks/mlp.py:27-127): D hidden ``DenseLayer``s of
width(s) W and an output layer, optional skip connections.  Same constructor, parameter names (``layers.{i}.weight`` /
``.bias``: checkpoints carry over) and forward signature.

Where the reference runs one GEMM + one activation kernel per layer (or hands the network to tiny-cuda-nn), this module
runs the whole network in ONE kernel on the fp32 MFMA (csrc/mlp.hip) -- or, for ``dtype=torch.half`` (the reference's autocast
layers / its tcnn FullyFusedMLP), on the f16 MFMA (csrc/mlp_half.hip) -- whenever it can: on a GPU, ReLU / no
activations, no skips / weight norm / equal_lr, every width <= 128.  Forward: activations stay in registers.  Backward
(hidden width <= 64): the forward is recomputed from x inside the backward kernel, so autograd keeps x and nothing else.
Everything else takes the layer-by-layer torch path below, with identical semantics."""
from typing import List, Union

import torch
import torch.nn as nn

from nr3d_lib_amd.models.layers import DenseLayer, get_nonlinearity
from nr3d_lib_amd.profile import profile

__all__ = ['MLP', 'FCBlock', 'FusedMLPFunction', 'FusedMLPHalfFunction']

USE_FUSED = True                       # False: always the layer-by-layer path (A/B measurements, debugging)
# Reuse of the MFMA-ordered weight copy between calls.  OFF by default: packing is one ~3 us kernel, and the only cheap
# change detector -- the parameters' (data_ptr, _version) -- does not see in-place edits made through `.data`
# (EMA swaps `p.data.copy_(shadow)`, weight clipping, `.data.normal_()` re-initialisation), after which a cached copy
# would silently be stale.  Opt in for inference loops over frozen weights; `MLP.invalidate_packed()` (also called by
# `train()` / `eval()` / `load_state_dict()`) drops the copy explicitly.
CACHE_PACKED = False


class FusedMLPFunction(torch.autograd.Function):
    """y = MLP(x) through nr3d_mlp_forward; backward through nr3d_mlp_backward (recomputes the forward).
    args: desc, need (bool: a gradient may be asked for -> also pack the transposed layers), x, W_0, b_0 | None, W_1, ...

    Higher order (``create_graph=True``, e.g. the eikonal term on nablas = d sdf / dx): backward() then runs with grad
    mode on; in that case the gradients are produced by differentiating a layer-by-layer PyTorch evaluation of the
    same network on the saved inputs, which autograd can differentiate again.  First-order training never takes that
    branch."""

    @staticmethod
    def forward(ctx, desc, need, x, *params):
        from nr3d_lib_amd.bindings import _mlp
        ws, bs = list(params[0::2]), list(params[1::2])
        # opt-in (CACHE_PACKED): the packed copy is keyed on the parameters alone; a copy packed with the transposed
        # layers (with_backward) also serves forward-only calls, a forward-only copy is upgraded when `need` first is True
        packed = None
        if CACHE_PACKED:
            key = tuple((p.data_ptr(), p._version) for p in params if p is not None)
            cached = getattr(desc, '_packed_cache', None)
            if cached is not None and cached[0] == key and (cached[2] or not need):
                packed = cached[1]
        if packed is None:
            packed = _mlp.pack(desc, ws, bs, with_backward=need)
            if CACHE_PACKED:
                desc._packed_cache = (key, packed, bool(need))
        if need:
            ctx.save_for_backward(x, packed, *[p for p in params if p is not None])
            ctx.desc, ctx.has_bias = desc, [b is not None for b in bs]
        return _mlp.forward(desc, x, packed)

    @staticmethod
    def backward(ctx, dL_dy):
        from nr3d_lib_amd.bindings import _mlp
        x, packed, *flat = ctx.saved_tensors
        n_layers = len(ctx.has_bias)
        if torch.is_grad_enabled():
            return (None, None, *FusedMLPFunction._differentiable_backward(ctx, x, flat, dL_dy))
        dx, dWs, dbs = _mlp.backward(ctx.desc, x, dL_dy.float(), packed, need_dx=ctx.needs_input_grad[2], has_bias=ctx.has_bias)
        grads = []
        for i in range(n_layers):
            grads += [dWs[i] if ctx.needs_input_grad[3 + 2 * i] else None,
                      dbs[i] if (dbs[i] is not None and ctx.needs_input_grad[4 + 2 * i]) else None]
        return (None, None, dx, *grads)

    @staticmethod
    def _differentiable_backward(ctx, x, flat, dL_dy):
        """(dL/dx, dL/dW_0, dL/db_0, ...) as differentiable functions of x, the parameters and dL/dy"""
        from nr3d_lib_amd.bindings import _mlp
        it = iter(flat)
        ws, bs = [], []
        for hb in ctx.has_bias:
            ws.append(next(it))
            bs.append(next(it) if hb else None)
        with torch.enable_grad():
            h = x if x.requires_grad else x.detach().requires_grad_(ctx.needs_input_grad[2])
            x_in = h
            for l, (W, b) in enumerate(zip(ws, bs)):
                h = torch.nn.functional.linear(h, W, b)
                act = ctx.desc.hidden_activation if l + 1 < len(ws) else ctx.desc.output_activation
                if act == _mlp.ACT_RELU:
                    h = torch.relu(h)
            wanted, slots = [], []
            if ctx.needs_input_grad[2]:
                wanted.append(x_in); slots.append(0)
            for l, (W, b) in enumerate(zip(ws, bs)):
                if ctx.needs_input_grad[3 + 2 * l]:
                    wanted.append(W); slots.append(1 + 2 * l)
                if b is not None and ctx.needs_input_grad[4 + 2 * l]:
                    wanted.append(b); slots.append(2 + 2 * l)
            got = torch.autograd.grad(h, wanted, dL_dy, create_graph=True, allow_unused=True) if wanted else ()
        out = [None] * (1 + 2 * len(ws))
        for s_, g in zip(slots, got):
            out[s_] = g
        return out


class FusedMLPHalfFunction(torch.autograd.Function):
    """y = MLP(x) in half precision through nr3d_mlp_half_forward / _backward (csrc/mlp_half.hip, f16 MFMA): what the
    reference gets from tiny-cuda-nn's FullyFusedMLP (``use_tcnn_backend``, nr3d_lib/models/tcnn_adapter.py:37-51,74-146) and,
    numerically, what its ``DenseLayer(dtype=half)`` chain computes under autocast (half operands, fp32 accumulation, half
    activations between the layers).  The parameters stay fp32 (as in the reference's DenseLayer): they are rounded to half when
    packed, their gradients come back as the fp32 sums the kernel accumulated.  x may be float or half; y is half; dL/dx has
    x's dtype.  args: desc, need, x, W_0, b_0 | None, W_1, ...  Higher order: as FusedMLPFunction."""

    @staticmethod
    def forward(ctx, desc, need, x, *params):
        from nr3d_lib_amd.bindings import _mlp
        ws, bs = list(params[0::2]), list(params[1::2])
        packed = _mlp.pack_half(desc, ws, bs, with_backward=need)
        xh = x if x.dtype == torch.float16 else x.half()
        if need:
            # the CALLER's x is saved, not its rounded copy: the create_graph branch of the backward differentiates through it (a
            # detached x.half() was a fresh leaf: higher-order terms w.r.t. x never reached the caller's tensor -- round-4 advisor)
            ctx.save_for_backward(x, packed, *[p for p in params if p is not None])
            ctx.desc, ctx.has_bias, ctx.x_dtype = desc, [b is not None for b in bs], x.dtype
        return _mlp.forward_half(desc, xh, packed)

    @staticmethod
    def backward(ctx, dL_dy):
        from nr3d_lib_amd.bindings import _mlp
        x, packed, *flat = ctx.saved_tensors
        n_layers = len(ctx.has_bias)
        if torch.is_grad_enabled():
            # differentiable form: ONE dtype for x, the (fp32) parameters and dL/dy -- F.linear(half, float) raises outside
            # autocast; the casts are differentiable, so the graph reaches the caller's x whatever its dtype
            out = FusedMLPFunction._differentiable_backward(ctx, x.float(), flat, dL_dy.float())
            out[0] = None if out[0] is None else out[0].to(ctx.x_dtype)
            return (None, None, *out)
        xh = x if x.dtype == torch.float16 else x.half()
        dx, dWs, dbs = _mlp.backward_half(ctx.desc, xh, dL_dy.half(), packed, need_dx=ctx.needs_input_grad[2], has_bias=ctx.has_bias)
        grads = []
        for i in range(n_layers):
            grads += [dWs[i] if ctx.needs_input_grad[3 + 2 * i] else None,
                      dbs[i] if (dbs[i] is not None and ctx.needs_input_grad[4 + 2 * i]) else None]
        return (None, None, None if dx is None else dx.to(ctx.x_dtype), *grads)


class MLP(nn.Module):
    def __init__(self, in_features: int, out_features: int, *, D: int = 4, W: Union[int, List[int]] = 128, skips: List[int] = [],
                 activation: Union[str, dict] = 'relu', output_activation: Union[str, dict] = None, bias=True,
                 last_bias: bool = None, equal_lr=False, weight_norm=False, dtype: Union[str, torch.dtype] = None,
                 device: torch.device = None):
        super().__init__()
        self.dtype = dtype = (dtype if isinstance(dtype, torch.dtype) or dtype is None
                              else getattr(torch, str(dtype).replace('torch.', '')))
        nl, gain, init_fn, first_init_fn = get_nonlinearity(activation)
        last_nl, last_gain, last_init_fn, _ = get_nonlinearity(output_activation)
        if last_bias is None:
            last_bias = bias
        self.D = D
        self.Ws = [W] * D if isinstance(W, int) else W
        if self.D >= 1:
            assert len(self.Ws) == D, f"The length of list W={self.Ws} should be D={D}."
        self.in_features, self.out_features = in_features, out_features
        self.skips, self.activation, self.output_activation = skips, activation, output_activation
        layers = []
        for l in range(self.D + 1):
            out_dim = out_features if l == self.D else self.Ws[l]
            in_dim = in_features if l == 0 else (in_features + self.Ws[l - 1] if l in self.skips else self.Ws[l - 1])
            last = l == self.D
            layer = DenseLayer(in_dim, out_dim, activation=last_nl if last else nl, bias=last_bias if last else bias,
                               dtype=self.dtype or torch.float, device=device, equal_lr=equal_lr)
            layer.apply(last_init_fn if last else (first_init_fn if l == 0 else init_fn))
            if weight_norm:
                layer = nn.utils.weight_norm(layer)
            layers.append(layer)
        self.layers = nn.ModuleList(layers)
        self._plain = not (skips or weight_norm or equal_lr)
        self._desc = None

    @property
    def device(self) -> torch.device:
        return self.layers[0].weight.device

    def invalidate_packed(self):
        """drop the cached MFMA-ordered weight copy (see CACHE_PACKED): call after editing parameters through `.data`"""
        if self._desc:
            self._desc._packed_cache = None

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def get_weight_reg(self, norm_type: float = 2.0):
        return torch.stack([p.norm(p=norm_type) for n, p in self.layers.named_parameters()])

    # ---- the fused path ------------------------------------------------------------------------------------------------
    @staticmethod
    def _act_code(layer):
        from nr3d_lib_amd.bindings import _mlp
        a = layer.activation
        if a is None:
            return _mlp.ACT_NONE
        return _mlp.ACT_RELU if isinstance(a, nn.ReLU) else None

    def fused_desc(self):
        """the kernel-side description of this network, or None when the fused kernels do not apply to it"""
        if self._desc is None:
            from nr3d_lib_amd.bindings import _mlp
            ok = self._plain and self.D >= 1 and self.dtype in (None, torch.float32, torch.float16)
            hid = {self._act_code(l) for l in self.layers[:-1]}
            out = self._act_code(self.layers[-1])
            if ok and len(hid) == 1 and None not in hid and out is not None and len(self.layers) <= _mlp.MAX_LAYERS:
                d = _mlp.MLPDesc([self.in_features, *self.Ws, self.out_features], hid.pop(), out)
                self._desc = d if (d.half_fusable if self.dtype == torch.float16 else d.fusable) else False
            else:
                self._desc = False
        return self._desc or None

    def _fused_ok(self, x, return_last, input_max_channel):
        half = self.dtype == torch.float16        # dtype=half: the layers run under autocast in the reference -> the f16-MFMA kernels
        if not (USE_FUSED and x.is_cuda and (x.dtype == torch.float32 or (half and x.dtype == torch.float16))
                and not return_last and input_max_channel is None):
            return None
        if torch.is_autocast_enabled():
            return None
        desc = self.fused_desc()
        if desc is None:
            return None
        self._needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.layers.parameters()))
        return desc if (not self._needs_grad or (desc.half_backward_fusable if half else desc.backward_fusable)) else None

    def forward_columns(self, x: torch.Tensor, n_cols: int):
        """The first ``n_cols`` output columns (fp32: contiguous) -- for queries that need a few numbers per sample and no
        gradient (the density column of a pruning query: nr3d_lib/graphics/nerf/nerf_ray_query.py:105-127 slices it out of the
        full output).  On the fused path the SAME kernels run on the network whose last layer is cut to its first rows: the columns
        they compute are bit-identical (every output column is its own dot product), the other 4 (out - n_cols) bytes per sample
        are neither written nor read back through a strided view (6.9 M samples x 16 outputs: 0.44 GB each way).  With a gradient in
        play, or off the fused path: ``self(x)[..., :n_cols]``."""
        n_cols = int(n_cols)
        desc = None
        if n_cols < self.out_features and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.layers.parameters()))):
            desc = self._fused_ok(x, False, None)
        if desc is None:
            return self(x)[..., :n_cols]
        from nr3d_lib_amd.bindings import _mlp
        # (the half kernels store rows of whole 8-byte pieces on their fast path: four columns at least there, the view below drops the rest)
        n_keep = n_cols
        if self.dtype == torch.float16:
            n_cols = min(self.out_features, (n_cols + 3) // 4 * 4)
        sub = _mlp.MLPDesc([self.in_features, *self.Ws, n_cols], desc.hidden_activation, desc.output_activation)
        params = []
        for layer in self.layers[:-1]:
            params += [layer.weight, layer.bias]
        last = self.layers[-1]
        params += [last.weight[:n_cols], None if last.bias is None else last.bias[:n_cols]]
        fn = FusedMLPHalfFunction if self.dtype == torch.float16 else FusedMLPFunction
        with torch.no_grad():
            return fn.apply(sub, False, x, *params)[..., :n_keep]

    @profile
    def forward(self, x: torch.Tensor, return_last: bool = False, input_max_channel: int = None):
        desc = self._fused_ok(x, return_last, input_max_channel)
        if desc is not None:
            params = []
            for layer in self.layers:
                params += [layer.weight, layer.bias]
            fn = FusedMLPHalfFunction if self.dtype == torch.float16 else FusedMLPFunction
            return fn.apply(desc, self._needs_grad, x, *params)
        # layer-by-layer path: layer 0 may see a truncated input, skip layers see [h, x], the input of the output layer
        # (index D) is what return_last hands back
        h, last_h = self.layers[0](x, max_channel=input_max_channel), None
        for i in range(1, len(self.layers)):
            if i == self.D and i not in self.skips:
                last_h = h
            h = self.layers[i](torch.cat([h, x], dim=-1) if i in self.skips else h)
        return (h, last_h) if return_last else h


FCBlock = MLP
