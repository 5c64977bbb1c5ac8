#!/usr/bin/env python
"""tools/prof_march.py [side] -- bench.py's march_composite_rate alone (for rocprofv3 --kernel-trace)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.march_composite_rate(torch.device("cuda", 0), iters=20, side=int(sys.argv[1]) if len(sys.argv) > 1 else 64)))
