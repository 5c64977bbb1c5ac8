"""GPU: the fused MLP kernels (csrc/mlp.hip through bindings._mlp / models.blocks.MLP) against plain PyTorch.

The f32 MFMA is an fmaf chain, so the fused network differs from torch's fp32 GEMM path by summation order only.  ReLU
makes the function discontinuous in the pre-activations, hence the yardstick: the error against an fp64 evaluation must
stay within a few times the error of torch's OWN fp32 path against the same fp64 evaluation (and within 1e-5 of the
output scale when no unit sits on a kink)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # dims, n, hidden act, out act, bias
    ([32, 32, 16], 4099, "relu", None, True),
    ([32, 64, 64, 16], 10007, "relu", None, True),
    ([18, 64, 3], 777, "relu", "relu", True),
    ([16, 32, 32, 32, 7], 4097, "relu", None, False),
    ([40, 48, 33], 33, None, None, True),
    ([3, 8, 1], 1, "relu", None, True),
    ([64, 64, 64], 2048, "relu", None, True),
]


def _net(dims, hidden, out, bias, dev, seed=0):
    from nr3d_lib_amd.models.blocks import MLP
    torch.manual_seed(seed)
    m = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1:-1], activation=hidden or "none", output_activation=out, bias=bias,
            dtype=torch.float, device=dev)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.4 if p.dim() > 1 else 0.2))
    return m


def _reference(m, x, gy, dtype):
    """layer-by-layer torch evaluation in `dtype` -> (y, dx, [dW], [db])"""
    h = x.detach().to(dtype).requires_grad_(True)
    h0 = h
    ws = [l.weight.detach().to(dtype).requires_grad_(True) for l in m.layers]
    bs = [None if l.bias is None else l.bias.detach().to(dtype).requires_grad_(True) for l in m.layers]
    for l, W, b in zip(m.layers, ws, bs):
        h = torch.nn.functional.linear(h, W, b)
        if l.activation is not None:
            h = torch.relu(h)
    h.backward(gy.to(dtype))
    return h.detach(), h0.grad, [w.grad for w in ws], [None if b is None else b.grad for b in bs]


def _check(name, got, ref64, ref32):
    scale = float(ref64.abs().max()) or 1.0
    err = float((got.double() - ref64).abs().max()) / scale
    err32 = float((ref32.double() - ref64).abs().max()) / scale
    assert err <= max(1e-5, 4 * err32), f"{name}: rel err {err:.2e} (torch fp32 path: {err32:.2e})"


@pytest.mark.parametrize("dims,n,hidden,out,bias", CASES)
def test_fused_forward_backward_match_torch(dev, dims, n, hidden, out, bias):
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    m = _net(dims, hidden, out, bias, dev)
    desc = m.fused_desc()
    assert desc is not None and desc.backward_fusable
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, dims[0], generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    y64, dx64, dW64, db64 = _reference(m, x, gy, torch.float64)
    y32, dx32, dW32, db32 = _reference(m, x, gy, torch.float32)
    y = m(x)
    assert y.grad_fn is not None and "FusedMLPFunction" in type(y.grad_fn).__name__
    y.backward(gy)
    _check("y", y.detach(), y64, y32)
    _check("dL_dx", x.grad, dx64, dx32)
    for l, layer in enumerate(m.layers):
        _check(f"dL_dW{l}", layer.weight.grad, dW64[l], dW32[l])
        if bias:
            _check(f"dL_db{l}", layer.bias.grad, db64[l], db32[l])
    # inference (no autograd): same kernel, no transposed weights packed
    with torch.no_grad():
        _check("y (no_grad)", m(x), y64, y32)
    # the layer-by-layer path of the same module agrees
    mlp_mod.USE_FUSED = False
    try:
        _check("unfused y", m(x).detach(), y64, y32)
    finally:
        mlp_mod.USE_FUSED = True


@pytest.mark.parametrize("dims,n,hidden,out,bias", CASES)
def test_feature_major_input_is_consumed_in_place(dev, dims, n, hidden, out, bias):
    """x as the LoTD forward returns it ([features, n] storage viewed as [n, features]): the kernels read it without the
    transposing copy (16-byte aligned output rows or not: both backward kernels), and dL/dx comes back in the same layout"""
    from nr3d_lib_amd.bindings import _mlp
    m = _net(dims, hidden, out, bias, dev, seed=3)
    g = torch.Generator(device="cpu").manual_seed(4)
    xt = torch.randn(dims[0], n, generator=g).to(dev).requires_grad_(True)
    x = xt.t()
    assert n == 1 or x.stride() == (1, n)
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    y64, dx64, dW64, db64 = _reference(m, x, gy, torch.float64)
    y32, dx32, dW32, db32 = _reference(m, x, gy, torch.float32)
    y = m(x)
    y.backward(gy)
    _check("y", y.detach(), y64, y32)
    _check("dL_dx", xt.grad.t(), dx64, dx32)
    for l, layer in enumerate(m.layers):
        _check(f"dL_dW{l}", layer.weight.grad, dW64[l], dW32[l])
        if bias:
            _check(f"dL_db{l}", layer.bias.grad, db64[l], db32[l])
    # the binding hands dL/dx back feature-major (what the LoTD parameter-gradient pass reads without transposing)
    desc = m.fused_desc()
    packed = _mlp.pack(desc, [l.weight for l in m.layers], [l.bias for l in m.layers], with_backward=True)
    dx, _, _ = _mlp.backward(desc, x.detach(), gy, packed, need_dx=True, has_bias=[bias] * len(m.layers))
    assert n == 1 or dx.stride() == (1, n)
    _check("dL_dx (binding)", dx, dx64, dx32)
    with torch.no_grad():
        _check("y (no_grad)", m(x), y64, y32)


def test_fused_handles_strided_rows_leading_dims_and_frozen_inputs(dev):
    m = _net([32, 64, 16], "relu", None, True, dev)
    g = torch.Generator(device="cpu").manual_seed(2)
    big = torch.randn(6, 50, 40, generator=g).to(dev)
    x = big[..., 3:35]                                   # rows of 32 floats at stride 40, not 16-byte aligned
    gy = torch.randn(6, 50, 16, generator=g).to(dev)
    y64, _, dW64, _ = _reference(m, x.reshape(-1, 32), gy.reshape(-1, 16), torch.float64)
    y32, _, dW32, _ = _reference(m, x.reshape(-1, 32), gy.reshape(-1, 16), torch.float32)
    y = m(x)                                             # x does not require grad: dL/dx is not computed
    assert tuple(y.shape) == (6, 50, 16)
    y.backward(gy)
    _check("y", y.detach().reshape(-1, 16), y64, y32)
    _check("dW0", m.layers[0].weight.grad, dW64[0], dW32[0])
    # empty batch
    assert tuple(m(torch.zeros(0, 32, device=dev)).shape) == (0, 16)
    # parameter gradients accumulate across calls like any autograd op
    before = m.layers[1].weight.grad.clone()
    m(x).backward(gy)
    torch.testing.assert_close(m.layers[1].weight.grad, 2 * before, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float, torch.half])
@pytest.mark.parametrize("shape", [(32, 16, 1, 32), (32, 16, 2, 64), (18, 3, 1, 32)])
def test_forward_columns_is_the_full_forward_sliced(dev, hip_option, dtype, shape):
    """MLP.forward_columns (round 6; the pruning query keeps one of 16 outputs): the network with its last layer cut to the asked rows on
    the same kernels -- bit-identical to the slice of the full forward on every route (f32 MFMA, bf16 x3, f16), contiguous; with a
    gradient in play it IS the slice of the differentiable forward"""
    from nr3d_lib_amd.models.blocks import MLP
    fin, fout, D, W = shape
    torch.manual_seed(5)
    net = MLP(fin, fout, D=D, W=W, dtype=dtype, device=dev)
    x = torch.randn(70001, fin, device=dev).to(dtype)
    for x3 in ((1, 0) if dtype == torch.float else (1,)):
        hip_option("mlp_x3", x3)
        with torch.no_grad():
            full = net(x)
            for k in (1, min(3, fout)):
                part = net.forward_columns(x, k)
                assert (part.is_contiguous() or dtype == torch.half) and tuple(part.shape) == (x.shape[0], k) and part.dtype == full.dtype
                assert torch.equal(part, full[:, :k]), f"x3={x3} k={k}"
            assert torch.equal(net.forward_columns(x, fout), full)
    y = net.forward_columns(x[:100], 1)                                        # grad mode, parameters require grad: the differentiable route
    assert y.requires_grad and tuple(y.shape) == (100, 1)


def test_networks_outside_the_fused_range_take_the_torch_path(dev):
    from nr3d_lib_amd.bindings import _mlp
    from nr3d_lib_amd.models.blocks import MLP, get_blocks
    x = torch.randn(100, 32, device=dev)
    wide = MLP(32, 4, D=2, W=128, dtype=torch.float, device=dev)              # forward fuses, backward does not (W > 64)
    assert wide.fused_desc() is not None and not wide.fused_desc().backward_fusable
    with torch.no_grad():
        y_f = wide(x)
    y_t = wide(x)                                                             # needs grad -> torch path
    assert "FusedMLP" not in type(y_t.grad_fn).__name__
    torch.testing.assert_close(y_f, y_t.detach(), rtol=1e-4, atol=1e-5)
    for kw in (dict(skips=[1]), dict(activation="softplus"), dict(weight_norm=True), dict(D=0, W=[]), dict(W=200)):
        args = dict(D=2, W=32, dtype=torch.float, device=dev); args.update(kw)
        net = MLP(32, 4, **args)
        assert net.fused_desc() is None
        assert tuple(net(x).shape) == (100, 4)
    # get_blocks(..., dtype=half, use_tcnn_backend=True) -- the reference's request for its tcnn FullyFusedMLP -- gets the f16-MFMA
    # kernels since round 4 (rounds 1-3: the torch autocast path); bf16 still takes the torch path
    half = get_blocks(32, 4, D=1, W=64, dtype=torch.half, device=dev, use_tcnn_backend=True, weight_norm=False)
    d = half.fused_desc()
    assert d is not None and d.half_fusable and d.half_backward_fusable
    yh = half(x)
    assert yh.dtype == torch.float16 and "FusedMLPHalfFunction" in type(yh.grad_fn).__name__
    bf = MLP(32, 4, D=1, W=64, dtype=torch.bfloat16, device=dev)
    assert bf.fused_desc() is None and bf(x).dtype == torch.bfloat16
    assert _mlp.MLPDesc([32, 64, 16]).fusable and not _mlp.MLPDesc([32, 16]).fusable and not _mlp.MLPDesc([300, 64, 3]).fusable
    with pytest.raises(RuntimeError, match="outside the fused"):
        _mlp.pack(_mlp.MLPDesc([32, 16]), [torch.zeros(16, 32, device=dev)], [None])
    with pytest.raises(RuntimeError, match="backward does not apply"):
        _mlp.pack(_mlp.MLPDesc([32, 128, 4]), [torch.zeros(128, 32, device=dev), torch.zeros(4, 128, device=dev)], [None, None], True)


def test_second_order_through_the_fused_block(dev):
    """eikonal-style use: nablas = d(sum y)/dx with create_graph, then a loss on the nablas back-propagated to the
    parameters -- the fused block has to give what the layer-by-layer path gives"""
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    m = _net([16, 32, 32, 4], "relu", None, True, dev, seed=5)
    g = torch.Generator(device="cpu").manual_seed(3)
    x0 = torch.randn(513, 16, generator=g).to(dev)

    def run():
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = m(x)
        nablas, = torch.autograd.grad(y[:, 0].sum(), x, create_graph=True)
        loss = ((nablas.norm(dim=-1) - 1.0) ** 2).mean() + y.square().mean()
        loss.backward()
        return y.detach(), nablas.detach(), [p.grad.clone() for p in m.parameters()], x.grad.clone()
    yf, nf, gf, xf = run()
    mlp_mod.USE_FUSED = False
    try:
        yt, nt, gt, xt = run()
    finally:
        mlp_mod.USE_FUSED = True
    torch.testing.assert_close(yf, yt, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(nf, nt, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xf, xt, rtol=1e-3, atol=1e-5)
    for a, b in zip(gf, gt):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("x_dtype", [torch.float32, torch.float16])
def test_second_order_through_the_half_block(dev, x_dtype):
    """MLP(dtype=half) under create_graph (round-4 advisor: the differentiable branch mixed half x with fp32 parameters and raised,
    and for float x it differentiated a detached copy, so nothing of the second order reached the caller's x): nablas with
    create_graph, a loss on them, gradients to the parameters AND to x, against the same network evaluated in fp32"""
    from nr3d_lib_amd.models.blocks import MLP
    torch.manual_seed(11)
    m = MLP(16, 4, D=2, W=32, dtype=torch.half, device=dev)
    ref = MLP(16, 4, D=2, W=32, dtype=torch.float, device=dev)
    ref.load_state_dict(m.state_dict())
    g = torch.Generator(device="cpu").manual_seed(4)
    x0 = torch.randn(777, 16, generator=g).to(dev)

    def run(net, dt):
        net.zero_grad(set_to_none=True)
        x = x0.to(dt).clone().requires_grad_(True)
        y = net(x)
        nablas, = torch.autograd.grad(y[:, 0].float().sum(), x, create_graph=True)
        assert nablas.dtype == dt and nablas.requires_grad
        # sums, not means: a half network's backward carries dL/dy and every dL/d(pre-activation) in half -- gradients of 1e-5
        # would sit in half's subnormal range (the reference's half decoders run under a loss scale for the same reason)
        loss = ((nablas.float().norm(dim=-1) - 1.0) ** 2).sum() + y.float().square().sum()
        loss.backward()
        return nablas.detach().float(), [p.grad.float().clone() for p in net.parameters()], x.grad.float().clone()
    nh, gh, xh = run(m, x_dtype)
    nr, gr, xr = run(ref, torch.float32)
    # half against fp32: values agree to half precision EXCEPT in samples where a hidden unit sits within half rounding of its ReLU
    # kink -- there the two networks switch the unit differently and a whole weight path enters or leaves that sample's gradient
    # (test_half_fused_forward_backward moves its inputs off the kinks; here the fp32 network is the reference, so such rows are
    # counted instead: a few per cent at most).  Parameter gradients are sums over all samples: compared as a whole.
    def rows_off(a, b, tol=2e-2):
        return float(((a - b).abs().amax(1) > tol * float(b.abs().max())).float().mean())
    assert torch.isfinite(nh).all() and torch.isfinite(xh).all() and float(xh.abs().max()) > 0
    assert rows_off(nh, nr) < 0.05, f"nablas: {rows_off(nh, nr):.3f} of the rows differ"
    assert rows_off(xh, xr) < 0.05, f"x.grad: {rows_off(xh, xr):.3f} of the rows differ"
    for i, (a, b) in enumerate(zip(gh, gr)):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0, f"param {i}"
        err = float((a - b).norm() / b.norm())
        assert err < 5e-2, f"param {i}: relative difference of the gradient {err:.3e}"


def _layerwise(m, x):
    ref = x
    for i, l in enumerate(m.layers):
        ref = torch.nn.functional.linear(ref, l.weight, l.bias)
        ref = torch.relu(ref) if i + 1 < len(m.layers) else ref
    return ref


def test_edits_through_dot_data_are_seen_by_default(dev):
    """`.data` writes do not bump tensor version counters (EMA swaps, weight clipping, re-initialisation): with the
    default settings the fused block packs the weights on every call and cannot go stale"""
    from nr3d_lib_amd.bindings import _mlp
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    assert mlp_mod.CACHE_PACKED is False
    m = _net([32, 32, 8], "relu", None, True, dev, seed=7)
    x = torch.randn(257, 32, device=dev)
    with torch.no_grad():
        y1 = m(x)
        v0 = m.layers[0].weight._version
        m.layers[0].weight.data.mul_(2.0)
        m.layers[1].bias.data.copy_(torch.ones_like(m.layers[1].bias))
        assert m.layers[0].weight._version == v0                           # the edit is invisible to the version counter
        y2 = m(x)
    assert not torch.equal(y1, y2)
    torch.testing.assert_close(y2, _layerwise(m, x), rtol=1e-4, atol=1e-5)
    xr = x.clone().requires_grad_(True)
    m(xr).sum().backward()                                                 # the backward's packed copy is fresh too
    xt = x.clone().requires_grad_(True)
    _layerwise(m, xt).sum().backward()
    torch.testing.assert_close(xr.grad, xt.grad, rtol=1e-4, atol=1e-5)


def test_opt_in_packed_weight_cache(dev):
    """CACHE_PACKED = True: the packed copy is keyed on the parameters only -- a copy packed for a differentiable call also
    serves no-grad calls, a forward-only copy is upgraded when gradients are first wanted; version bumps, train() / eval(),
    load_state_dict() and invalidate_packed() drop it"""
    from nr3d_lib_amd.bindings import _mlp
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    m = _net([32, 32, 8], "relu", None, True, dev, seed=7)
    x = torch.randn(257, 32, device=dev)
    calls = []
    orig = _mlp.pack
    _mlp.pack = lambda *a, **k: (calls.append(k.get("with_backward")), orig(*a, **k))[1]
    mlp_mod.CACHE_PACKED = True
    try:
        with torch.no_grad():
            y1 = m(x); y2 = m(x)
        assert calls == [False] and torch.equal(y1, y2)
        m(x).sum().backward()                                             # gradients wanted: upgraded to a with_backward copy
        assert calls == [False, True]
        with torch.no_grad():
            y3 = m(x)                                                     # ... which serves the forward-only call as well
        assert calls == [False, True] and torch.equal(y1, y3)
        with torch.no_grad():
            m.layers[0].weight.mul_(2.0)                                  # in-place write bumps the version counter
            y4 = m(x)
        assert len(calls) == 3 and not torch.equal(y1, y4)
        m.layers[0].weight.data.mul_(0.5)                                 # invisible edit: the documented caveat ...
        with torch.no_grad():
            stale = m(x)
            assert len(calls) == 3 and torch.equal(stale, y4)
            m.invalidate_packed()                                         # ... and its remedy
            y5 = m(x)
        assert len(calls) == 4
        torch.testing.assert_close(y5, _layerwise(m, x), rtol=1e-4, atol=1e-5)
        m.eval()
        with torch.no_grad():
            m(x)
        assert len(calls) == 5
        m.load_state_dict(m.state_dict())
        with torch.no_grad():
            m(x)
        assert len(calls) == 6
    finally:
        _mlp.pack = orig
        mlp_mod.CACHE_PACKED = False


@pytest.mark.parametrize("name", ["plain", "ragged_out_relu"])
def test_fused_block_reproduces_the_reference_modules_outputs(dev, name):
    """weights and outputs of the reference's own MLP module (tests/golden/ref_blocks.npz) through the fused kernels"""
    from test_golden_blocks_cpu import GOLD, load_mlp
    gold = np.load(GOLD)
    m = load_mlp(gold, name, device=dev)
    assert m.fused_desc() is not None
    x = torch.from_numpy(gold[f"mlp_{name}_x"]).to(dev)
    with torch.no_grad():
        y = m(x)
    want = gold[f"mlp_{name}_y"]
    assert float(np.abs(y.cpu().numpy() - want).max()) <= 1e-5 * max(1.0, float(np.abs(want).max()))


# ------------------------------------------------------------------------------------------------------------------------
# half precision on the f16 MFMA (csrc/mlp_half.hip): MLP(dtype=torch.half)
# ------------------------------------------------------------------------------------------------------------------------
HALF_CASES = CASES + [([32, 128, 128, 16], 3001, "relu", None, True), ([128, 64, 128], 515, "relu", None, True),
                      # round 5: 64-wide hidden layers run the backward with dW split over the waves of the workgroup (k_mlph_bwd_split):
                      # every tile-count combination of its two-hidden-layer form, several rounds per workgroup (n > 65 536), an
                      # input narrower than its tile, an output ReLU, no bias
                      ([64, 64, 64, 64], 70001, "relu", None, True), ([24, 64, 64, 40], 5000, "relu", None, False),
                      ([64, 48, 64, 8], 3000, "relu", "relu", True), ([32, 64, 64, 16], 200001, "relu", None, True),
                      ([33, 64, 50], 66000, None, None, True)]


def _half_reference(m, x, gy):
    """the half contract in fp64: weights / biases / x rounded to half, every layer's output rounded to half (straight-through
    for the gradient) -> (y, dx, [dW], [db]) as fp64 tensors"""
    rnd = lambda t: t.half().double()

    class _Round(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return rnd(t)

        @staticmethod
        def backward(ctx, g):
            return g
    h0 = rnd(x.detach()).requires_grad_(True)
    h = h0
    ws = [rnd(l.weight.detach()).requires_grad_(True) for l in m.layers]
    bs = [None if l.bias is None else rnd(l.bias.detach()).requires_grad_(True) for l in m.layers]
    for l, W, b in zip(m.layers, ws, bs):
        h = torch.nn.functional.linear(h, W, b)
        if l.activation is not None:
            h = torch.relu(h)
        h = _Round.apply(h)
    h.backward(rnd(gy))
    return h.detach(), h0.grad, [w.grad for w in ws], [None if b is None else b.grad for b in bs]


def _check_half(name, got, ref64, tol):
    scale = float(ref64.abs().max()) or 1.0
    err = float((got.double() - ref64).abs().max()) / scale
    assert torch.isfinite(got).all() and err <= tol, f"{name}: rel err {err:.2e} > {tol:.2e}"


@pytest.mark.parametrize("layout", ["row_major", "feature_major"])
@pytest.mark.parametrize("dims,n,hidden,out,bias", HALF_CASES)
def test_half_fused_forward_backward(dev, dims, n, hidden, out, bias, layout):
    """MLP(dtype=half) runs on the f16-MFMA kernels: y against the half contract evaluated in fp64 at 2^-9 of the output scale
    (one half rounding of the result + the odd activation that rounds the other way after fp32 instead of exact accumulation),
    gradients at 2^-7 (dL/d(pre-activation) is rounded to half between the layers, as in the reference's half networks; a
    pre-activation within half rounding of a ReLU kink flips a unit -- the inputs are moved off the kinks below); parameters
    stay fp32 and so do their gradients.  Row-major and feature-major x, fused backward where the shape allows it, the torch
    autocast path otherwise."""
    from nr3d_lib_amd.models.blocks import MLP
    torch.manual_seed(0)
    m = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1:-1], activation=hidden or "none", output_activation=out, bias=bias,
            dtype=torch.half, device=dev)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.4 if p.dim() > 1 else 0.2))
    desc = m.fused_desc()
    assert desc is not None and desc.half_fusable
    g = torch.Generator(device="cpu").manual_seed(1)
    xs = torch.randn(n, dims[0], generator=g).to(dev)
    if layout == "feature_major":
        xt = xs.t().contiguous().requires_grad_(True)
        x = xt.t()
    else:
        xt = x = xs.clone().requires_grad_(True)
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    y64, dx64, dW64, db64 = _half_reference(m, x, gy)
    y = m(x)
    assert y.dtype == torch.float16
    fused_bwd = desc.half_backward_fusable
    assert (y.grad_fn is not None and "FusedMLPHalfFunction" in type(y.grad_fn).__name__) == fused_bwd
    y.backward(gy.half())
    _check_half("y", y.detach(), y64, 2.0 ** -9)
    with torch.no_grad():
        yn = m(x)
        assert yn.dtype == torch.float16
        _check_half("y (no_grad: fused forward for every shape)", yn, y64, 2.0 ** -9)
    # rows whose ReLU pattern differs between the kernel and the fp64 reference (a pre-activation within rounding of zero) are
    # excluded from the gradient comparison by construction: compare on the sum over rows, where a flipped unit is O(1/n)
    tol = 2.0 ** -7
    gx = (xt.grad.t() if layout == "feature_major" else xt.grad)
    assert gx.dtype == torch.float32 and gx.shape == dx64.shape
    bad = ((gx.double() - dx64).abs().amax(1) > tol * float(dx64.abs().max()))
    assert float(bad.float().mean()) <= 0.02, f"dL_dx: {int(bad.sum())} of {n} rows off"
    for l, layer in enumerate(m.layers):
        assert layer.weight.grad.dtype == torch.float32
        _check_half(f"dL_dW{l}", layer.weight.grad, dW64[l], 4 * tol if n < 100 else tol)
        if bias:
            _check_half(f"dL_db{l}", layer.bias.grad, db64[l], 4 * tol if n < 100 else tol)


def test_half_fused_agrees_with_the_autocast_layers(dev):
    """the same module with USE_FUSED off runs the reference's way (DenseLayer under autocast: half GEMMs): the fused half
    kernels must agree with it to half precision, forward and parameter gradients"""
    from nr3d_lib_amd.models.blocks import MLP
    from nr3d_lib_amd.models.blocks import mlp as mlp_mod
    torch.manual_seed(3)
    m = MLP(32, 16, D=2, W=64, dtype=torch.half, device=dev)
    x = torch.randn(20000, 32, device=dev)
    gy = torch.randn(20000, 16, device=dev).half()
    outs = {}
    for fused in (True, False):
        mlp_mod.USE_FUSED = fused
        try:
            m.zero_grad(set_to_none=True)
            xr = x.clone().requires_grad_(True)
            y = m(xr)
            y.backward(gy)
            outs[fused] = (y.detach().float(), xr.grad.float(), [l.weight.grad.clone() for l in m.layers])
        finally:
            mlp_mod.USE_FUSED = True
    scale = float(outs[False][0].abs().max())
    assert float((outs[True][0] - outs[False][0]).abs().max()) <= 2.0 ** -8 * scale
    for a, b in zip(outs[True][2], outs[False][2]):
        assert float((a - b).abs().max()) <= 2.0 ** -6 * float(b.abs().max())


@pytest.mark.parametrize("dims", [[32, 64, 64, 16], [18, 32, 3], [64, 64, 64, 64], [100, 128, 128, 7], [5, 17, 9]])
def test_fp32_forward_on_the_bf16_mfma_is_fp32_grade(dev, hip_option, dims):
    """the fp32 forward's two routes (round 5): the f32 MFMA (mlp_x3 = 0) and the bf16 MFMA on three-piece splits of every value with the
    six significant piece products (1, default) -- against the same layers in float64: the split route must be as close to it as the
    f32 route is (both carry fp32 rounding of the accumulations; the split drops terms below 2^-26 of a product)"""
    m = _net(dims, "relu", None, True, dev, seed=13)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn(4099, dims[0], generator=g) * torch.logspace(-3, 3, dims[0])[None, :].clamp(1e-2, 30)).to(dev)    # columns of very different scale
    with torch.no_grad():
        ref = x.double()
        for i, l in enumerate(m.layers):
            ref = torch.nn.functional.linear(ref, l.weight.double(), None if l.bias is None else l.bias.double())
            if i + 1 < len(m.layers):
                ref = torch.relu(ref)
    scale = float(ref.abs().max())
    err = {}
    for mode in (0, 1):
        hip_option("mlp_x3", mode)
        with torch.no_grad():
            y = m(x)
        assert torch.isfinite(y).all()
        err[mode] = float((y.double() - ref).abs().max()) / scale
    assert err[0] < 5e-6 and err[1] < 5e-6, err
    assert err[1] <= 3.0 * err[0] + 2e-7, f"three-piece bf16 route {err[1]:.2e} against the f32 MFMA's {err[0]:.2e}"


@pytest.mark.parametrize("dims", [[32, 64, 64, 16], [32, 32, 16], [18, 32, 3], [27, 32, 32, 32, 1], [32, 64, 48], [64, 64, 64, 64],
                                  [64, 64, 64], [32, 64, 64, 64], [64, 64, 64, 16], [40, 64, 7]])
def test_fp32_backward_on_the_bf16_mfma_is_fp32_grade(dev, hip_option, dims):
    """round 6: the fp32 backward's routes -- the f32 MFMA (mlp_x3 = 0: one padded LDS copy of the forward layers, the dH = W^T dPre
    chain reads it transposed) and the bf16 MFMA on three-piece splits (1, default: forward recomputation, dH chain through the
    transposing LDS read of the forward layers' planes, sample contraction dW = dPre^T H; csrc/mlp.hip backward_x3() takes it where
    the planes leave as many waves as the f32 copy, i.e. not for 64-wide inputs AND outputs) -- against the same network
    differentiated in float64: the split route must be as close to it as the f32 route is, for dL/dx, every dL/dW and dL/db"""
    from nr3d_lib_amd.bindings import _mlp
    m = _net(dims, "relu", None, True, dev, seed=17)
    desc = m.fused_desc()
    assert desc is not None and desc.backward_fusable
    g = torch.Generator(device="cpu").manual_seed(11)
    n = 8205
    x = (torch.randn(n, dims[0], generator=g) * torch.logspace(-2, 1, dims[0])[None, :]).to(dev)
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    m64 = [(l.weight.detach().double().requires_grad_(True), l.bias.detach().double().requires_grad_(True)) for l in m.layers]
    x64 = x.double().requires_grad_(True)
    h = x64
    for i, (w, b) in enumerate(m64):
        h = torch.nn.functional.linear(h, w, b)
        if i + 1 < len(m64):
            h = torch.relu(h)
    h.backward(gy.double())
    ref = [x64.grad] + [w.grad for w, _ in m64] + [b.grad for _, b in m64]
    err = {}
    for mode in (0, 1):
        hip_option("mlp_x3", mode)
        xr = x.detach().requires_grad_(True)
        m.zero_grad(set_to_none=True)
        m(xr).backward(gy)
        got = [xr.grad] + [l.weight.grad for l in m.layers] + [l.bias.grad for l in m.layers]
        assert all(torch.isfinite(t).all() for t in got)
        # a ReLU unit whose pre-activation rounds to the other side of zero flips one sample's contribution on either route: compare
        # by the norm of the difference relative to the norm of the reference, per tensor
        err[mode] = [float((a.double() - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(got, ref)]
    assert max(err[0]) < 2e-5 and max(err[1]) < 2e-5, err
    for e0, e1 in zip(err[0], err[1]):
        assert e1 <= 3.0 * e0 + 1e-6, f"three-piece bf16 backward {e1:.2e} against the f32 MFMA's {e0:.2e}"
