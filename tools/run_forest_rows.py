import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_reference_workloads as b
for r in b.forest_rows(torch.device('cuda:0')):
    print(json.dumps({k: r.get(k) for k in ('name', 'ref_value_us', 'ours_us', 'device_us_median', 'ref_over_ours', 'error')}))
