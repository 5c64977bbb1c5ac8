#!/bin/bash
# Run on the GPU box: configs[4]'s loop on one GPU, fp32 end to end vs the reference's default storage (half tables + half decoders)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for prec in ("float", "half"):
    print(json.dumps(bench.full_loop_rate(dev, precision=prec)))
PY
