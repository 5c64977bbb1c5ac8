#!/usr/bin/env python
"""tools/prof_full_loop.py -- bench.py's full_loop_rate alone (for rocprofv3 --kernel-trace)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
if "SPATIAL_ORDER_BITS" in os.environ:              # A/B of the driver's Morton order (0 = ray order)
    from nr3d_lib_amd.graphics.nerf import nerf_ray_query as _drv
    _drv.SPATIAL_ORDER_BITS = int(os.environ["SPATIAL_ORDER_BITS"])
print(json.dumps(bench.full_loop_rate(torch.device("cuda", 0), precision=os.environ.get("FULL_LOOP_PRECISION", "float"))))
