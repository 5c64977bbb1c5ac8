"""experiment: does overlapping stage A of one half batch with stage B of the other (two HIP streams) pay?"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
dev = torch.device("cuda", 0)
cfg = gen_ngp_cfg()
m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
N = 1 << 20
g = torch.Generator().manual_seed(1)
p = torch.empty(m.n_params).uniform_(-1e-4, 1e-4, generator=g).to(dev)
x = torch.rand(N, 3, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
dy = (torch.randn(N, m.n_encoded_dims, generator=g) / 1e4).to(dev)
def full():
    return _lotd.lod_bwd(m, dy, x, p, None, need_input_grad=False, need_param_grad=True)[1]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = N // 2
def halves():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = _lotd.lod_bwd(m, dy[:h], x[:h], p, None, need_input_grad=False, need_param_grad=True)[1]
    with torch.cuda.stream(s2):
        b = _lotd.lod_bwd(m, dy[h:], x[h:], p, None, need_input_grad=False, need_param_grad=True)[1]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
for fn, name in ((full, "one stream, full batch"), (halves, "two streams, half batches")):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 20 * 1e3, 4), "ms")
a, b = halves(); torch.cuda.synchronize()
print("max rel diff", float(((a + b) - full()).abs().max() / full().abs().max()))
