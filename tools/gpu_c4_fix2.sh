#!/bin/bash
# Run on the GPU box: configs[3] with k_vm_direct in fp64 (direct_fixed = 1) and in fixed point (direct_fixed = 2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for MODE in 1 2; do
python - <<PY 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('direct_fixed=$MODE', d['ms'], d['ms_total'])"
import sys
sys.path.insert(0, '.')
from nr3d_lib_amd import _hip
_hip.set_option("direct_fixed", $MODE)
sys.argv = ["bench_c4.py", "--iters", "10"]
exec(open("tools/bench_c4.py").read())
PY
done
