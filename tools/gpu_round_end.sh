#!/bin/bash
# usage: tools/gpu_round_end.sh <tag>   (run on the GPU box through gpurun)
# end-of-round evidence: full GPU suite, smoke, profiles (kernel trace + PMC traffic + counters), then the full bench line
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profile.sh ${TAG} pmc > /dev/null 2>&1; head -16 gpurun_out/${TAG}/bench_kernel_stats.txt
bash tools/gpu_counters.sh ${TAG}_ctr "k_fwd|k_pair_bin|k_pair_accum|k_contract|k_march|k_composite" > /dev/null 2>&1
cp gpurun_out/${TAG}/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null     # bench.py reads the committed file for roofline.traffic
cd $ROOT && python bench.py 2>/dev/null > /tmp/bench_out.txt
tail -1 /tmp/bench_out.txt > gpurun_out/${TAG}_bench.json                  # the compact line the driver parses
cp bench_extra.json gpurun_out/${TAG}_bench_extra.json 2>/dev/null        # the full tables
wc -c gpurun_out/${TAG}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['whole_step_frac'] if 'whole_step_frac' in d else d['roofline'].get('whole_step_frac'))
print(d['cpu_baseline'])
for k,v in d['extra'].items(): print(k, json.dumps(v)[:400])
"
