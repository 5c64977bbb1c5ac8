#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel table:
calls, total / average / min / max duration.  Usage: python tools/prof_summary.py <results.db> [min_calls]"""
import re
import sqlite3
import sys


def main(path, out=sys.stdout):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    rows = c.execute(f"select s.{name_col}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for n, dur in rows:
        n = re.sub(r"\(.*", "", n or "?")
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"{'kernel':78s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}", file=out)
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:78]:78s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f} "
              f"{100 * a[1] / tot:6.2f}", file=out)


def pmc(path, out=sys.stdout):
    """per-kernel average of every collected PMC counter (rocprofv3 --pmc ...)"""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    pe = next(t for t in tabs if t.startswith("rocpd_pmc_event"))
    pi = next(t for t in tabs if t.startswith("rocpd_info_pmc"))
    scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    rows = c.execute(f"select s.{name_col}, i.name, e.value, d.id from {pe} e join {kd} d on e.event_id = d.event_id "
                     f"join {ks} s on d.kernel_id = s.id join {pi} i on e.pmc_id = i.id").fetchall()
    agg = {}
    for n, cn, v, did in rows:
        n = re.sub(r"\(.*", "", n or "?")
        a = agg.setdefault((n, cn), {})
        a[did] = a.get(did, 0.0) + float(v)          # sum over instances (XCDs / channels) of one dispatch
    print(f"{'kernel':70s} {'counter':14s} {'dispatches':>10s} {'avg_per_dispatch':>18s}", file=out)
    for (n, cn), a in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{n[:70]:70s} {cn:14s} {len(a):10d} {sum(a.values()) / len(a):18.1f}", file=out)


def pmc_json(path, out_path):
    """merge the per-kernel averages of one PMC pass into a JSON file {kernel: {counter: avg_per_dispatch}}"""
    import io, json, os
    buf = io.StringIO()
    pmc(path, buf)
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for line in buf.getvalue().splitlines()[1:]:
        name, rest = line[:70].strip(), line[70:].split()
        if len(rest) == 3:
            data.setdefault(name, {})[rest[0]] = float(rest[2])
    # stamp: the counters belong to ONE build of the library; bench.py prints `traffic: null` for any other
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nr3d_lib_amd", "libnr3d_hip.so")
    if os.path.exists(lib):
        data["__lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "pmc-json":
        pmc_json(sys.argv[1], sys.argv[3])
    elif len(sys.argv) > 2 and sys.argv[2] == "pmc":
        pmc(sys.argv[1])
    else:
        main(sys.argv[1])
