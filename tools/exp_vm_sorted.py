"""tools/exp_vm_sorted.py -- time of k_vm_sorted on the reference's forest workload.
   python tools/exp_vm_sorted.py levels     : production build; the kernel's time with the levels up to max_level = 2 .. 8 (differences = per level)
   NR3D_VS_DBG=<bits> python tools/exp_vm_sorted.py   : experiments build only (1 no boundary pass, 2 no write-out, 4 no LDS adds,
                                                       8 no own points; results wrong by design)"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from nr3d_lib_amd import _hip
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
grad = (torch.randn(n, meta.n_encoded_dims, generator=gen) / 1.0e4).to(dev).half()


def run(max_level=None, tag=""):
    for _ in range(2):
        _lotd.lod_bwd(metas, grad, x, params, None, blidx, None, None, max_level, False, True)
    _hip.prof_read("lotd_direct")
    _hip.prof_enable("lotd_direct")
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        _lotd.lod_bwd(metas, grad, x, params, None, blidx, None, None, max_level, False, True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    _hip.prof_enable()
    ms, k = _hip.prof_read("lotd_direct")
    print(f"{tag}NR3D_VS_DBG={os.environ.get('NR3D_VS_DBG', '0')} max_level={max_level}: k_vm_sorted {ms / max(k, 1):.3f} ms ({k} launches), pass {wall:.3f} ms", flush=True)
    return ms / max(k, 1)


if len(sys.argv) > 1 and sys.argv[1] == "levels":
    _hip.set_option("vm_sorted", 2)
    prev = 0.0
    for ml in range(2, 9):
        t = run(ml)
        print(f"   level {ml} (R = {res[ml]}): {t - prev:.3f} ms")
        prev = t
else:
    run()
