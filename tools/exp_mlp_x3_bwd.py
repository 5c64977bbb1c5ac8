"""tools/exp_mlp_x3_bwd.py -- fp32 decoder forward / forward+backward with the bf16 x3 route on and off (run on the GPU box).
mlp_x3 = 1: csrc/mlp.hip backward_x3() decides whether the backward runs on the bf16 MFMA; 0: f32 MFMA throughout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd import _hip as H
from nr3d_lib_amd.models.blocks import MLP

dev = torch.device("cuda:0")
n = 1 << 22


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for dims in ((32, 64, 64, 16), (32, 32, 16), (18, 32, 3), (32, 32, 32, 16), (32, 64, 16), (64, 64, 64, 64), (64, 64, 64), (32, 64, 64, 64), (32, 64, 64), (64, 64, 16), (64, 64, 64, 16)):
    x = torch.randn(n, dims[0], device=dev)
    gy = torch.randn(n, dims[-1], device=dev)
    row, grads = {}, {}
    for x3 in (1, 0):
        H.set_option("mlp_x3", x3)
        torch.manual_seed(0)
        net = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=torch.float, device=dev)

        def fwd():
            with torch.no_grad():
                return net(x)

        def fwd_bwd():
            xr = x.detach().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            net(xr).backward(gy)
            return xr.grad
        f, fb = timed(fwd), timed(fwd_bwd)
        row[x3] = (round(f, 3), round(fb, 3), round(fb - f, 3))
        gx = fwd_bwd()
        grads[x3] = [gx] + [p.grad.clone() for p in net.parameters()]
    H.set_option("mlp_x3", -1)
    err = max(float((a - b).norm() / b.norm()) for a, b in zip(grads[1], grads[0]))      # (ReLU masks of |pre-activation| < 1 ulp may differ)
    print(dims, "x3 (fwd, fwd+bwd, bwd):", row[1], " f32:", row[0], " max relative L2 difference of the gradients %.1e" % err, flush=True)
