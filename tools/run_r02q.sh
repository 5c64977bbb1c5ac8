#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r02q
export TMPDIR=/tmp PYTHONPATH=$ROOT
cat > /tmp/c3.py <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
r = bench.march_composite_rate(dev, iters=20, side=64, cpu_seconds=0.0)
print(os.environ.get("NR3D_MARCH_GROUP"), r["ms_per_iter"], r["mrays_per_s"], r["kernel_us_per_iter"])
PY
cd /tmp
for g in 1 16 64; do
rm -rf /tmp/prof_$g
NR3D_MARCH_GROUP=$g rocprofv3 --kernel-trace --stats -d /tmp/prof_$g -o c3 -- python /tmp/c3.py > /tmp/log_$g.txt 2>&1
grep -v rocprofv3 /tmp/log_$g.txt | tail -2
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_$g -name '*.db' | head -1)" 2>&1 | head -14 | tee $ROOT/gpurun_out/r02q/c3_group${g}_kernel_stats.txt
done
