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
