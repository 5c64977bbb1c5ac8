#!/bin/bash
# Run on the GPU box: configs[3] parity tests of the VM / CP paths, then tools/bench_c4.py with the VM line tables in LDS and as records
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "vm or c4 or mixed or half_tables or binned or atomic" 2>&1 | tail -4
python tools/bench_c4.py --iters 20 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('lines in LDS:', d['ms'], d['ms_total'], {k: v['frac'] for k, v in d['roofline']['per_pass'].items()})"
python - <<'PY' 2>/dev/null | tail -1
import subprocess, sys, json, os
sys.path.insert(0, '.')
from nr3d_lib_amd import _hip
_hip.set_option("vm_lines_direct", 0)
sys.argv = ["bench_c4.py", "--iters", "10"]
exec(open("tools/bench_c4.py").read().replace('if __name__ == "__main__":', 'if True:'))
PY
