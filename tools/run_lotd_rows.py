"""tools/run_lotd_rows.py [filter] -- the LoTD rows of tools/bench_reference_workloads.py, one JSON line each (GPU box)"""
import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_reference_workloads as b
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in b.lotd_rows(torch.device('cuda:0')):
    if flt in r["name"]:
        print(json.dumps({k: r.get(k) for k in ('name', 'ref_value_us', 'ours_us', 'device_us_median', 'ref_over_ours', 'error')}))
