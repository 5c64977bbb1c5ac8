"""tools/exp_mlp_wide_half.py [float] -- forward / forward + backward of the half fused decoder on its 64-wide shapes (the backward
instantiations that still spill), 2^22 samples, against the bytes each sample moves"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from nr3d_lib_amd.models.blocks import MLP
dev = torch.device("cuda:0")
n = 1 << 22
DT = torch.float if (len(sys.argv) > 1 and sys.argv[1] == "float") else torch.half
BPE = 4 if DT == torch.float else 2


def timed(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for dims in ([32, 64, 64, 16], [32, 64, 64, 64], [64, 64, 64, 32], [64, 64, 64, 64], [64, 64, 64], [32, 32, 32, 32, 32]):
    torch.manual_seed(0)
    net = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=DT, device=dev)
    x = torch.randn(n, dims[0], device=dev).to(DT)
    gy = torch.randn(n, dims[-1], device=dev).to(DT)

    def fwd():
        with torch.no_grad():
            return net(x)

    def fwd_bwd():
        xr = x.detach().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        net(xr).backward(gy)
    f, fb = timed(fwd), timed(fwd_bwd)
    byt = n * BPE * (dims[0] + dims[-1]) + n * BPE * (2 * dims[0] + 2 * dims[-1])        # fwd: x + y; bwd: x + gy + gx (+ y of the fwd in front)
    mac = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    print(f"{dims}: fwd {f:.3f} ms, fwd + bwd {fb:.3f} ms   (HBM floor of both at 5.3 TB/s: {byt / 5.3e9:.3f} ms; "
          f"{6 * mac * n / (fb * 1e-3) / 1e12:.1f} TFLOP/s)")
