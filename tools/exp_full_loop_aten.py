#!/usr/bin/env python
"""tools/exp_full_loop_aten.py -- which ATen operators (and from which Python lines) launch the non-nr3d kernels of one
full-loop iteration (configs[4] on one GPU, bench._full_loop_setup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
model, n, fwd_bwd = bench._full_loop_setup(dev, 512)
for _ in range(3):
    model.zero_grad(set_to_none=True); fwd_bwd()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.zero_grad(set_to_none=True); fwd_bwd()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    ct = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if ct <= 0 or not e.key.startswith("aten::"):
        continue
    stack = [s for s in e.stack if "nr3d_lib_amd" in s or "demo_field" in s or "bench.py" in s]
    rows.append((ct, e.count, e.key, stack[0].strip() if stack else "(autograd engine / other)"))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {sum(r[1] for r in rows)} calls, {tot:.0f} us")
for ct, cnt, key, where in rows[:45]:
    print(f"{ct:8.1f} us {cnt:3d}x {key:32s} {where[-110:]}")
