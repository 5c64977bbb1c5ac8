"""ctypes front of the fused fully-connected decoder (include/nr3d_hip.h: nr3d_mlp_*).  No reference pybind module
corresponds to it -- the reference's MLP (nr3d_lib/models/blocks/mlp.py) is plain torch, its fused option is
tiny-cuda-nn (nr3d_lib/models/tcnn_adapter.py); this is the kernel side of ``nr3d_lib_amd.models.blocks.MLP``."""
import ctypes as C

import torch

from .. import _hip as H

MAX_LAYERS = 8
ACT_NONE, ACT_RELU = 0, 1


class _CDesc(C.Structure):
    _fields_ = [("n_layers", C.c_uint32), ("dims", C.c_uint32 * (MAX_LAYERS + 1)), ("hidden_activation", C.c_uint32),
                ("output_activation", C.c_uint32)]


class MLPDesc:
    """dims = [in_features, hidden..., out_features]"""

    def __init__(self, dims, hidden_activation=ACT_RELU, output_activation=ACT_NONE):
        self.dims = [int(d) for d in dims]
        self.hidden_activation, self.output_activation = int(hidden_activation), int(output_activation)
        c = _CDesc()
        n_layers = len(self.dims) - 1
        c.n_layers = n_layers if 1 <= n_layers <= MAX_LAYERS else 0
        for i, d in enumerate(self.dims[:MAX_LAYERS + 1]):
            c.dims[i] = d
        c.hidden_activation, c.output_activation = self.hidden_activation, self.output_activation
        self._c = c
        l = H.lib()
        for fn in (l.nr3d_mlp_packed_floats, l.nr3d_mlp_backward_packed_floats, l.nr3d_mlp_half_packed_bytes,
                   l.nr3d_mlp_half_backward_packed_bytes):
            fn.restype = C.c_uint64
        self.packed_floats = int(l.nr3d_mlp_packed_floats(C.byref(c))) if c.n_layers else 0
        self.backward_floats = int(l.nr3d_mlp_backward_packed_floats(C.byref(c))) if self.packed_floats else 0
        # the half-precision twin (csrc/mlp_half.hip, f16 MFMA): sizes in bytes
        self.half_packed_bytes = int(l.nr3d_mlp_half_packed_bytes(C.byref(c))) if c.n_layers else 0
        self.half_backward_bytes = int(l.nr3d_mlp_half_backward_packed_bytes(C.byref(c))) if self.half_packed_bytes else 0

    @property
    def fusable(self) -> bool:
        return self.packed_floats > 0

    @property
    def backward_fusable(self) -> bool:
        return self.backward_floats > 0

    @property
    def half_fusable(self) -> bool:
        return self.half_packed_bytes > 0

    @property
    def half_backward_fusable(self) -> bool:
        return self.half_backward_bytes > 0


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def pack(desc: MLPDesc, weights, biases, with_backward=False) -> torch.Tensor:
    """weights[l] [dims[l+1], dims[l]], biases[l] [dims[l+1]] | None (fp32, contiguous, on one GPU) -> packed buffer"""
    if not desc.fusable:
        raise RuntimeError("mlp.pack: network outside the fused kernels' range")
    dev = weights[0].device
    H.require_gpu(*weights)
    ws = [w.detach().float().contiguous() for w in weights]
    bs = [None if b is None else b.detach().float().contiguous() for b in biases]
    for l, w in enumerate(ws):
        if tuple(w.shape) != (desc.dims[l + 1], desc.dims[l]):
            raise RuntimeError(f"mlp.pack: weights[{l}] has shape {list(w.shape)}, expected {[desc.dims[l + 1], desc.dims[l]]}")
    if with_backward and not desc.backward_fusable:
        raise RuntimeError("mlp.pack: the fused backward does not apply to this network")
    packed = H.empty(desc.packed_floats + (desc.backward_floats if with_backward else 0), dtype=torch.float32, device=dev)
    with H.on_device(dev):
        H.check(H.lib().nr3d_mlp_pack(C.byref(desc._c), _ptr_array(ws), _ptr_array(bs), H.ptr(packed), C.c_int(int(with_backward)),
                                      H.stream_of(packed)))
    return packed


def _layout(t2: torch.Tensor):
    """[n, w] tensor -> (tensor, row stride, feature stride) the kernels can read in place: row-major (rows may be strided)
    or feature-major ([w, n] storage viewed as [n, w], what the LoTD forward returns); anything else is copied"""
    n, w = t2.shape
    if w == 1 or t2.stride(1) == 1:
        return t2, (t2.stride(0) if n > 1 else w), 1
    if n > 1 and t2.stride(0) == 1:
        return t2, 1, t2.stride(1)
    return t2.contiguous(), w, 1


def forward(desc: MLPDesc, x: torch.Tensor, packed: torch.Tensor) -> torch.Tensor:
    """x [..., in] fp32 (row-major with any row stride, or feature-major) -> y [..., out]"""
    H.require_gpu(x, packed)
    if x.dtype != torch.float32 or x.shape[-1] != desc.dims[0]:
        raise RuntimeError(f"mlp.forward: expected fp32 input with {desc.dims[0]} features, got {x.dtype} {list(x.shape)}")
    x2, xs, xf = _layout(x.reshape(-1, x.shape[-1]))
    n = x2.shape[0]
    y = H.empty((n, desc.dims[-1]), dtype=torch.float32, device=x.device)
    with H.on_device(x.device):
        H.check(H.lib().nr3d_mlp_forward(C.byref(desc._c), C.c_uint64(n), H.ptr(x2), H.i64(xs), H.i64(xf),
                                         H.ptr(packed), H.ptr(y), H.i64(y.shape[1]), H.stream_of(x)))
    return y.view(*x.shape[:-1], desc.dims[-1])


def backward(desc: MLPDesc, x: torch.Tensor, dL_dy: torch.Tensor, packed: torch.Tensor, need_dx=True, has_bias=None):
    """-> (dL_dx | None, [dL_dW_l], [dL_db_l | None]); `packed` from pack(..., with_backward=True).  dL_dx has the layout
    of x: for a feature-major x (the LoTD features) it is feature-major too, which is what the LoTD parameter-gradient
    pass reads without a transposition."""
    H.require_gpu(x, dL_dy, packed)
    n_layers = len(desc.dims) - 1
    x2 = x.reshape(-1, desc.dims[0])
    g2 = dL_dy.reshape(-1, desc.dims[-1])
    if x2.dtype != torch.float32 or g2.dtype != torch.float32 or x2.shape[0] != g2.shape[0]:
        raise RuntimeError("mlp.backward: expected fp32 x [n, in] and dL_dy [n, out]")
    x2, xs, xf = _layout(x2)
    g2 = g2 if (g2.stride(-1) == 1 or g2.shape[1] == 1) else g2.contiguous()
    n, dev = x2.shape[0], x.device
    has_bias = [True] * n_layers if has_bias is None else list(has_bias)
    # the kernel ADDS its workgroups' partial sums into dW / db (one atomic per element and workgroup): all of them are views
    # of ONE zero-filled buffer -- one fill launch instead of 2 per layer
    sizes = [desc.dims[l + 1] * desc.dims[l] for l in range(n_layers)] + [desc.dims[l + 1] if has_bias[l] else 0 for l in range(n_layers)]
    pool = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    parts = pool.split(sizes)
    dWs = [parts[l].view(desc.dims[l + 1], desc.dims[l]) for l in range(n_layers)]
    dbs = [parts[n_layers + l] if has_bias[l] else None for l in range(n_layers)]
    dx, gxs, gxf = None, desc.dims[0], 1
    if need_dx and xf != 1:
        dx, gxs, gxf = H.empty((desc.dims[0], n), dtype=torch.float32, device=dev).t(), 1, n
    elif need_dx:
        dx = H.empty((n, desc.dims[0]), dtype=torch.float32, device=dev)
    with H.on_device(dev):
        H.check(H.lib().nr3d_mlp_backward(
            C.byref(desc._c), C.c_uint64(n), H.ptr(x2), H.i64(xs), H.i64(xf), H.ptr(g2),
            H.i64(g2.stride(0) if n > 1 else desc.dims[-1]), H.ptr(packed), H.ptr(dx), H.i64(gxs), H.i64(gxf), _ptr_array(dWs),
            _ptr_array(dbs), H.stream_of(x)))
    return (None if dx is None else dx.reshape(x.shape)), dWs, dbs


# ------------------------------------------------------------------------------------------------
# half precision on the f16 MFMA (nr3d_mlp_half_*): half x / weights / y, fp32 accumulation inside a layer, activations
# rounded to half between the layers -- the contract of the reference's tcnn FullyFusedMLP (models/tcnn_adapter.py:37-51)
# ------------------------------------------------------------------------------------------------
def pack_half(desc: MLPDesc, weights, biases, with_backward=False) -> torch.Tensor:
    """weights[l] [dims[l+1], dims[l]], biases[l] [dims[l+1]] | None (half, on one GPU) -> packed byte buffer"""
    if not desc.half_fusable:
        raise RuntimeError("mlp.pack_half: network outside the fused kernels' range")
    dev = weights[0].device
    H.require_gpu(*weights)
    ws = [w.detach().half().contiguous() for w in weights]
    bs = [None if b is None else b.detach().half().contiguous() for b in biases]
    for l, w in enumerate(ws):
        if tuple(w.shape) != (desc.dims[l + 1], desc.dims[l]):
            raise RuntimeError(f"mlp.pack_half: weights[{l}] has shape {list(w.shape)}, expected {[desc.dims[l + 1], desc.dims[l]]}")
    if with_backward and not desc.half_backward_fusable:
        raise RuntimeError("mlp.pack_half: the fused backward does not apply to this network")
    packed = H.empty(desc.half_packed_bytes + (desc.half_backward_bytes if with_backward else 0), dtype=torch.uint8, device=dev)
    with H.on_device(dev):
        H.check(H.lib().nr3d_mlp_half_pack(C.byref(desc._c), _ptr_array(ws), _ptr_array(bs), H.ptr(packed), C.c_int(int(with_backward)),
                                           H.stream_of(packed)))
    return packed


def forward_half(desc: MLPDesc, x: torch.Tensor, packed: torch.Tensor) -> torch.Tensor:
    """x [..., in] half (row-major with any row stride, or feature-major) -> y [..., out] half"""
    H.require_gpu(x, packed)
    if x.dtype != torch.float16 or x.shape[-1] != desc.dims[0]:
        raise RuntimeError(f"mlp.forward_half: expected half input with {desc.dims[0]} features, got {x.dtype} {list(x.shape)}")
    x2, xs, xf = _layout(x.reshape(-1, x.shape[-1]))
    n = x2.shape[0]
    y = H.empty((n, desc.dims[-1]), dtype=torch.float16, device=x.device)
    with H.on_device(x.device):
        H.check(H.lib().nr3d_mlp_half_forward(C.byref(desc._c), C.c_uint64(n), H.ptr(x2), H.i64(xs), H.i64(xf),
                                              H.ptr(packed), H.ptr(y), H.i64(y.shape[1]), H.stream_of(x)))
    return y.view(*x.shape[:-1], desc.dims[-1])


def backward_half(desc: MLPDesc, x: torch.Tensor, dL_dy: torch.Tensor, packed: torch.Tensor, need_dx=True, has_bias=None):
    """-> (dL_dx half | None, [dL_dW_l fp32], [dL_db_l fp32 | None]); `packed` from pack_half(..., with_backward=True).  The
    parameter gradients are the fp32 sums the kernel accumulated (the caller rounds them to the parameters' dtype)."""
    H.require_gpu(x, dL_dy, packed)
    n_layers = len(desc.dims) - 1
    x2 = x.reshape(-1, desc.dims[0])
    g2 = dL_dy.reshape(-1, desc.dims[-1])
    if x2.dtype != torch.float16 or g2.dtype != torch.float16 or x2.shape[0] != g2.shape[0]:
        raise RuntimeError("mlp.backward_half: expected half x [n, in] and dL_dy [n, out]")
    x2, xs, xf = _layout(x2)
    g2 = g2 if (g2.stride(-1) == 1 or g2.shape[1] == 1) else g2.contiguous()
    n, dev = x2.shape[0], x.device
    has_bias = [True] * n_layers if has_bias is None else list(has_bias)
    sizes = [desc.dims[l + 1] * desc.dims[l] for l in range(n_layers)] + [desc.dims[l + 1] if has_bias[l] else 0 for l in range(n_layers)]
    pool = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    parts = pool.split(sizes)
    dWs = [parts[l].view(desc.dims[l + 1], desc.dims[l]) for l in range(n_layers)]
    dbs = [parts[n_layers + l] if has_bias[l] else None for l in range(n_layers)]
    dx, gxs, gxf = None, desc.dims[0], 1
    if need_dx and xf != 1:
        dx, gxs, gxf = H.empty((desc.dims[0], n), dtype=torch.float16, device=dev).t(), 1, n
    elif need_dx:
        dx = H.empty((n, desc.dims[0]), dtype=torch.float16, device=dev)
    with H.on_device(dev):
        H.check(H.lib().nr3d_mlp_half_backward(
            C.byref(desc._c), C.c_uint64(n), H.ptr(x2), H.i64(xs), H.i64(xf), H.ptr(g2),
            H.i64(g2.stride(0) if n > 1 else desc.dims[-1]), H.ptr(packed), H.ptr(dx), H.i64(gxs), H.i64(gxf), _ptr_array(dWs),
            _ptr_array(dbs), H.stream_of(x)))
    return (None if dx is None else dx.reshape(x.shape)), dWs, dbs
