#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import torch, bench, json
dev = torch.device("cuda", 0)
print(json.dumps(bench.lotd_half_rate(dev)))
from nr3d_lib_amd import _hip as H
names = ["lotd_fwd", "lotd_fwd_lds", "lotd_contract_dx", "lotd_bin", "lotd_accum"]
H.prof_enable(*names)
r = bench.lotd_half_rate(dev, iters=10)
for n in names: print(n, H.prof_read(n))
PY
