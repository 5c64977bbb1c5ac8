#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of the reference's forest workload (tools/bench_reference_workloads.py::forest_rows), dL/dparam only
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-forestprof}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/forest_run.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tools")
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd.models.spatial import ForestBlockSpace
dev = torch.device("cuda:0")
res = [34, 55, 90, 140, 230, 370, 600, 1000, 1600]
meta = _lotd.LoDMeta(3, res, [2] * 9, ["Dense", "Dense"] + ["VM"] * 7)
space = ForestBlockSpace(device=dev)
space.populate(mode="from_corners", corners=[[1, 1, 0], [1, 1, 1], [1, 1, 2], [2, 2, 2], [3, 2, 2], [4, 2, 2]], level=3)
metas = (meta, space.meta)
gen = torch.Generator(device="cpu").manual_seed(42)
n = 3653653
params = (torch.randn(meta.n_params * space.n_trees, generator=gen) / 1.0e2).to(dev).half()
x = torch.rand(n, 3, generator=gen).to(dev)
blidx = torch.randint(space.n_trees, (n,), generator=gen).to(dev)
grad = (torch.randn(n, meta.n_encoded_dims, generator=gen) / 1.0e4).to(dev).half()
for _ in range(${2:-4}):
    _lotd.lod_bwd(metas, grad, x, params, None, blidx, None, None, None, False, True)
torch.cuda.synchronize()
PY
rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o p -- python /tmp/forest_run.py > $OUT/run.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_f -name '*.db' | head -1)" > $OUT/forest_dparam_kernel_stats.txt 2>&1
head -24 $OUT/forest_dparam_kernel_stats.txt | cut -c1-170
