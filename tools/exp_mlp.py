"""scratch check of the fused MLP kernels against torch (fp64 reference) + timing"""
import sys, time; sys.path.insert(0, "/root/repo")
import torch
from nr3d_lib_amd.bindings import _mlp
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [([32, 32, 16], 100000), ([32, 64, 64, 16], 100003), ([18, 64, 3], 5000), ([16, 32, 32, 32, 7], 4097), ([40, 48, 33], 33),
         ([32, 64, 64, 16], 1670000), ([32, 32, 16], 1670000)]
for dims, n in cases:
    desc = _mlp.MLPDesc(dims)
    Ws = [(torch.randn(dims[i + 1], dims[i], device=dev) / dims[i] ** 0.5).requires_grad_(True) for i in range(len(dims) - 1)]
    bs = [(torch.randn(dims[i + 1], device=dev) * 0.1).requires_grad_(True) for i in range(len(dims) - 1)]
    x = torch.randn(n, dims[0], device=dev, requires_grad=True)
    gy = torch.randn(n, dims[-1], device=dev)
    print(dims, n, "fusable", desc.fusable, "bwd", desc.backward_fusable)
    if not desc.backward_fusable: continue
    packed = _mlp.pack(desc, Ws, bs, with_backward=True)
    dx, dWs, dbs = _mlp.backward(desc, x.detach(), gy, packed)
    def ref(dt):
        h = x.detach().to(dt).requires_grad_(True); h0 = h
        ws = [w.detach().to(dt).requires_grad_(True) for w in Ws]; bb = [b.detach().to(dt).requires_grad_(True) for b in bs]
        for i, (W, b) in enumerate(zip(ws, bb)):
            h = torch.nn.functional.linear(h, W, b)
            if i + 1 < len(ws): h = h.relu()
        h.backward(gy.to(dt))
        return h0.grad, [w.grad for w in ws], [b.grad for b in bb]
    rx, rW, rb = ref(torch.float64)
    tx, tW, tb = ref(torch.float32)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print("   dx  fused %.2e torch32 %.2e" % (rel(dx, rx), rel(tx, rx)))
    for l in range(len(Ws)):
        print("   dW%d fused %.2e torch32 %.2e   db%d fused %.2e torch32 %.2e" % (l, rel(dWs[l], rW[l]), rel(tW[l], rW[l]), l, rel(dbs[l], rb[l]), rel(tb[l], rb[l])))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): _mlp.backward(desc, x.detach(), gy, packed)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5): ref(torch.float32)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("   backward fused %.3f ms   torch fwd+bwd %.3f ms" % ((t1 - t0) * 200, (t2 - t1) * 200))
