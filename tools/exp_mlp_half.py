"""tools/exp_mlp_half.py -- half decoder forward / forward+backward at 2^22 samples (run on the GPU box); prints one line per shape.
For an A/B of two builds on ONE box: cp ab/libA.so nr3d_lib_amd/libnr3d_hip.so; python tools/exp_mlp_half.py; cp ab/libB.so ...; again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd.models.blocks import MLP

dev = torch.device("cuda:0")
n = 1 << 22
dtype = torch.half if (len(sys.argv) < 2 or sys.argv[1] == "half") else torch.float


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = []
for dims in ((32, 64, 64, 16), (32, 32, 16), (32, 32, 32, 16), (32, 64, 16), (64, 64, 64, 64)):
    torch.manual_seed(0)
    net = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=dtype, device=dev)
    x = torch.randn(n, dims[0], device=dev, dtype=dtype)
    gy = torch.randn(n, dims[-1], device=dev, dtype=dtype)

    def fwd():
        with torch.no_grad():
            return net(x)

    def fwd_bwd():
        xr = x.detach().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        net(xr).backward(gy)
    f, fb = timed(fwd), timed(fwd_bwd)
    out.append("%s %.3f/%.3f" % ("-".join(map(str, dims)), f, fb))
print(" | ".join(out), flush=True)
