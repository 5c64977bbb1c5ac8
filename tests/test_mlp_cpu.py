"""CPU: the MLP block's host semantics (layer wiring, parameter names, skips, return_last, get_blocks) -- the fused
kernels never run without a GPU, so this is the layer-by-layer path."""
import torch


def test_mlp_matches_a_hand_written_stack():
    from nr3d_lib_amd.models.blocks import FCBlock, MLP, get_blocks, get_mlp
    torch.manual_seed(0)
    m = MLP(6, 3, D=3, W=[8, 10, 12], skips=[2], activation="relu", output_activation="sigmoid", dtype=torch.float)
    assert FCBlock is MLP and get_mlp is get_blocks
    names = [n for n, _ in m.named_parameters()]
    assert names == [f"layers.{i}.{k}" for i in range(4) for k in ("weight", "bias")]
    assert [tuple(l.weight.shape) for l in m.layers] == [(8, 6), (10, 8), (12, 10 + 6), (3, 12)]
    x = torch.randn(5, 7, 6)
    h = torch.relu(torch.nn.functional.linear(x, m.layers[0].weight, m.layers[0].bias))
    h = torch.relu(torch.nn.functional.linear(h, m.layers[1].weight, m.layers[1].bias))
    h = torch.relu(torch.nn.functional.linear(torch.cat([h, x], -1), m.layers[2].weight, m.layers[2].bias))
    y = torch.sigmoid(torch.nn.functional.linear(h, m.layers[3].weight, m.layers[3].bias))
    out, last = m(x, return_last=True)
    torch.testing.assert_close(out, y); torch.testing.assert_close(last, h)
    # first-layer channel cut: the reference slices the BIAS with the same bound (layers.py:304), so it only works for
    # max_channel >= out_features -- replicated
    wide_in = MLP(12, 3, D=1, W=8, dtype=torch.float)
    x12 = torch.randn(4, 12)
    l0 = wide_in.layers[0]
    want = torch.relu(torch.nn.functional.linear(x12[:, :9], l0.weight[:, :9], l0.bias))
    torch.testing.assert_close(l0(x12[:, :9], max_channel=9), want)
    assert m.get_weight_reg().shape == (8,)
    nb = get_blocks(6, 2, D=1, W=4, bias=False, last_bias=True, dtype="float")
    assert nb.layers[0].bias is None and nb.layers[1].bias is not None
    eq = MLP(6, 2, D=1, W=4, equal_lr=True, dtype=torch.float)
    assert abs(eq.layers[0].weight_gain - 1 / 6 ** 0.5) < 1e-12 and tuple(eq(x).shape) == (5, 7, 2)


def test_packed_sizes_of_the_fused_decoder():
    """host logic of csrc/mlp.hip, no kernel runs: the packed buffer is [f32 layers | their bf16 x3 planes when they fit LDS]; the
    backward reads those same layers (transposed, from one padded LDS copy) and adds only a 4-float header (non-zero = "the fused
    backward applies"); 0 outside the fused backward's range"""
    from nr3d_lib_amd.bindings import _mlp

    def layer(ni, no):                    # tiles of 32
        return no * ni * 1024 + no * 32, no * ni * 1536 + no * 32

    def tiles(d):
        return (d + 31) // 32
    for dims in ((32, 64, 64, 16), (32, 32, 16), (18, 32, 3), (32, 32, 32, 16), (32, 64, 16), (64, 64, 64, 64), (64, 64, 64), (32, 64, 64, 64),
                 (32, 64, 64), (64, 64, 16)):
        d = _mlp.MLPDesc(list(dims), 1, 0)
        t = [tiles(v) for v in dims]
        f32 = sum(layer(a, b)[0] for a, b in zip(t[:-1], t[1:]))
        x3 = sum(layer(a, b)[1] for a, b in zip(t[:-1], t[1:]))
        assert d.packed_floats == f32 + x3, dims
        assert d.backward_floats == 4 and d.backward_fusable, (dims, d.backward_floats)
    # hidden width above 64, three hidden layers wider than 32, output wider than the hidden layers: forward only
    for dims in ((32, 128, 128, 16), (32, 64, 64, 64, 16), (32, 32, 64)):
        d = _mlp.MLPDesc(list(dims), 1, 0)
        assert d.packed_floats > 0 and d.backward_floats == 0 and not d.backward_fusable, dims
