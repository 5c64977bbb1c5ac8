#!/usr/bin/env python
"""tools/exp_full_loop_aten_shapes.py -- the ATen operators with device time in one full-loop iteration, grouped by input shapes
(which tensors the remaining copy_ / fill_ / add_ / index launches move)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "float"
model, n, fwd_bwd = bench._full_loop_setup(dev, 512, precision=prec)
for _ in range(3):
    model.zero_grad(set_to_none=True); fwd_bwd()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    model.zero_grad(set_to_none=True); fwd_bwd()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    ct = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if ct <= 0 or not e.key.startswith("aten::"):
        continue
    rows.append((ct, e.count, e.key, str(e.input_shapes)[:120]))
rows.sort(reverse=True)
print(f"aten ops with device time ({prec}): {sum(r[1] for r in rows)} calls, {sum(r[0] for r in rows):.0f} us")
for ct, cnt, key, shp in rows[:40]:
    print(f"{ct:8.1f} us {cnt:3d}x {key:28s} {shp}")
