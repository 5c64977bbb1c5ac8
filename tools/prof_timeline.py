#!/usr/bin/env python
"""tools/prof_timeline.py <results.db> <anchor-kernel-substring> <occurrence> ... -- kernels between the n-th and (n+1)-th
dispatch of the anchor kernel in a rocprofv3 database, with durations, gaps and grid sizes."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
nc = "display_name" if "display_name" in scol else "kernel_name"
rows = c.execute(f"select s.{nc}, d.start, d.end, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
names = [re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("nr3d::lotd::", "")[:34] for r in rows]
idx = [i for i, n in enumerate(names) if sys.argv[2] in n]
for which in map(int, sys.argv[3:]):
    a, b = idx[which], idx[which + 1]
    print(f"---- occurrence {which}: {(rows[b][1] - rows[a][1]) / 1e3:.1f} us")
    for i in range(a, b):
        r = rows[i]
        print(f"  {names[i]:36s} {(r[2] - r[1]) / 1e3:8.1f} us  gap {(r[1] - rows[i - 1][2]) / 1e3:6.1f}  grid {r[3]}x{r[4]}")
