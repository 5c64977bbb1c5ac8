"""tools/bench_rsort.py -- device time of the library's radix sort (csrc/rsort.hip) at the two sizes the library uses it at"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nr3d_lib_amd import _hip as H
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for n, bits, pair in ((3653653, 15, True), (1668733, 24, False), (1 << 20, 32, False)):
    k0 = torch.randint(0, 1 << min(bits, 31), (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    k1 = torch.randint(0, 1 << min(bits, 31), (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    keys = [k0, k1] if pair else k0
    for _ in range(3):
        H.sort_pairs_u32(keys, None, bits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.sort_pairs_u32(keys, None, bits)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} bits={bits} sorts_per_call={2 if pair else 1}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
