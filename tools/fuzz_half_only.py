"""tools/fuzz_half_only.py -- tools/fuzz_next_rows.py's half-decoder case alone, in a loop (run on the GPU box): usage fuzz_half_only.py [seconds] [seed].
Prints every failure instead of stopping at the first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import fuzz_next_rows as F
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed); torch.manual_seed(seed)
counts, fails = {}, 0
t0 = time.time()
while time.time() - t0 < budget:
    try:
        r = F.fuzz_mlp_half(rng)
    except AssertionError as e:
        r = "FAIL"; fails += 1
        print(str(e)[:300], flush=True)
    counts[r] = counts.get(r, 0) + 1
print(counts, "failures:", fails)
