"""host-time profile of the launch-bound pack ops (4096 packs x randint(32, 96), the reference's unit-test workload)"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nr3d_lib_amd.graphics.pack_ops as po
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
n = torch.randint(32, 96, [4096], generator=g).to(dev)
pi = po.get_pack_infos_from_n(n)
feats = torch.randn(int(n.sum()), 1, generator=g).to(dev)
for name, fn in (("packed_sum", lambda: po.packed_sum(feats, pi)), ("packed_cumsum", lambda: po.packed_cumsum(feats, pi)),
                 ("packed_diff", lambda: po.packed_diff(feats, pi)), ("interleave_arange_simple", lambda: po.interleave_arange_simple(n))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6 * (t1 - t0) / 2000:.1f} us/call, with the device drained {1e6 * (t2 - t0) / 2000:.1f} us/call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:2600])
