"""tools/exp_c3_group.py -- configs[2] (4096 rays) with the marcher's lanes per ray forced (NR3D_OPT_MARCH_GROUP): 16 / 32 / 64"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nr3d_lib_amd import _hip as H
dev = torch.device("cuda", 0)
for occ in ("random", "shell"):
    for g in (None, 16, 32, 64):
        H.set_option("march_group", g)
        r = bench.march_composite_rate(dev, iters=200, occupancy=occ)
        print(occ, "group", g, r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
H.set_option("march_group", None)
