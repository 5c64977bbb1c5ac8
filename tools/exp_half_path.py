"""timing of the LoTD module with dtype=half (the reference's default) vs float, fwd + bwd through autograd"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding, gen_ngp_cfg
dev = torch.device("cuda", 0)
N = 1 << 20
x = (torch.rand(N, 3, device=dev) * 2 - 1).clamp(-0.999, 0.999)
for dt in (torch.float, torch.half):
    enc = LoTDEncoding(3, lotd_cfg=gen_ngp_cfg(), dtype=dt, device=dev)
    g = torch.randn(N, enc.out_features, device=dev, dtype=dt) * 1e-4
    def step():
        enc.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_(False)
        y = enc(xx)
        y.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(dt, round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per fwd+bwd(param)")
