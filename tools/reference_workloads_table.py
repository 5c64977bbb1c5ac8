"""tools/reference_workloads_table.py <bench.json> -- extra.reference_workloads of a bench line as a text table (profiles/r04final_reference_workloads.txt)"""
import json, sys
d = json.load(open(sys.argv[1]))
rw = d["extra"]["reference_workloads"]
print(f"# extra.reference_workloads of {sys.argv[1]}")
print("# " + rw["protocol"])
print(f"# rows with a reference figure: {rw['rows_with_reference_figure']}, faster than it: {rw['rows_faster_than_reference_figure']}")
print(f"{'row':<104s} {'ref_us':>8s} {'ours_us':>10s} {'device_us':>10s} {'iters':>6s} {'ref/ours':>8s}")
for r in rw["rows"]:
    ratio = r.get("ref_over_ours")
    if ratio is None and r.get("ref_value_us") and r.get("ours_us"):
        ratio = round(r["ref_value_us"] / r["ours_us"], 2)
    print(f"{r['name'][:104]:<104s} {str(r.get('ref_value_us')):>8s} {str(r.get('ours_us')):>10s} {str(r.get('device_us_median')):>10s} {str(r.get('iters')):>6s} {str(ratio):>8s}")
