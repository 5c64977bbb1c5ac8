#!/usr/bin/env python
"""tools/prof_full_loop.py -- bench.py's full_loop_rate alone (for rocprofv3 --kernel-trace)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.full_loop_rate(torch.device("cuda", 0), precision=os.environ.get("FULL_LOOP_PRECISION", "float"))))
