import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernel records
names = collections.OrderedDict()
tail = rows[-30:]
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e - s) / 1e3:8.2f} us  gap {gap:8.2f} us")
    prev_end = e
