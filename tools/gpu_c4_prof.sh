#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of tools/bench_c4.py (dL/dparam passes), VM line tables in LDS (default) and as records
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-c4prof}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in 1 0; do
cat > /tmp/c4run.py <<PY
import sys
sys.path.insert(0, "$ROOT")
from nr3d_lib_amd import _hip
_hip.set_option("vm_lines_direct", $MODE)
sys.argv = ["bench_c4.py", "--iters", "5", "--warmup", "3"]
exec(open("$ROOT/tools/bench_c4.py").read())
PY
rm -rf /tmp/prof_c4 && rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o p -- python /tmp/c4run.py > /dev/null 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_c4 -name '*.db' | head -1)" > $OUT/c4_kernel_stats_lines$MODE.txt 2>&1
echo "== vm_lines_direct=$MODE"; head -14 $OUT/c4_kernel_stats_lines$MODE.txt | cut -c1-150
done
