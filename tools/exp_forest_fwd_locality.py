"""tools/exp_forest_fwd_locality.py -- how much of the forest / large-table forward is the RANDOM order of the benchmark's points?
the reference's forest workload (6 blocks, Dense x2 + VM x7 up to 1600^3, 3.65 M uniformly random points), forward and forward + dy/dx,
with the points as drawn and with the same points ordered by (block, 32^3 cell)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.spatial import ForestBlockSpace
dev = torch.device("cuda:0")
res = [34, 55, 90, 140, 230, 370, 600, 1000, 1600]
meta = _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["VM"] * 7)
space = ForestBlockSpace(device=dev)
space.populate(mode="from_corners", corners=[[1, 1, 0], [1, 1, 1], [1, 1, 2], [2, 2, 2], [3, 2, 2], [4, 2, 2]], level=3)
metas = (meta, space.meta)
gen = torch.Generator(device="cpu").manual_seed(42)
n = 3653653
params = (torch.randn(meta.n_params * space.n_trees, generator=gen) / 1.0e2).to(dev).half()
x = torch.rand(n, 3, generator=gen).to(dev)
blidx = torch.randint(space.n_trees, (n,), generator=gen).to(dev)
c = (x * 32).long().clamp_(0, 31)
key = ((blidx * 32 + c[:, 0]) * 32 + c[:, 1]) * 32 + c[:, 2]
order = torch.argsort(key)
xs, bs = x[order].contiguous(), blidx[order].contiguous()


def timed(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for name, (xx, bb) in (("as drawn", (x, blidx)), ("ordered by (block, 32^3 cell)", (xs, bs))):
    f = timed(lambda: _lotd.lod_fwd(metas, xx, params, bb, None, None, None, False))
    fj = timed(lambda: _lotd.lod_fwd(metas, xx, params, bb, None, None, None, True))
    print(f"forest forward, points {name}: {f:.3f} ms, with dy/dx {fj:.3f} ms")
m1 = _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["VM"] * 7, None, False)
p1 = (torch.randn(m1.n_params, generator=gen) / 1.0e2).to(dev).half()
o1 = torch.argsort((c[:, 0] * 32 + c[:, 1]) * 32 + c[:, 2])
x1 = x[o1].contiguous()
for name, xx in (("as drawn", x), ("ordered by 32^3 cell", x1)):
    f = timed(lambda: _lotd.lod_fwd(m1, xx, p1, None, None, None, None, False))
    print(f"single-table Dense x2 + VM x7 forward, points {name}: {f:.3f} ms")
