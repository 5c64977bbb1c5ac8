#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k && rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o p -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /tmp/b.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_k -name '*.db' | head -1)" | head -12 | cut -c1-140
cd $ROOT && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['whole_step_frac'])"
