#!/usr/bin/env python
"""tests/golden/make_ref_backend_uses.py -- the reference's call sites of its three compiled modules, as DATA.

    python tests/golden/make_ref_backend_uses.py            (needs /root/reference; run in the build container)

INTEGRATION.md section 1 promises a zero-code-change route: `sys.modules["nr3d_lib.bindings._lotd"] = nr3d_lib_amd.bindings._lotd`
(and `_pack_ops`, `_occ_grid`).  That holds iff every attribute the reference's Python takes from those modules exists on the
twin and every CALL it makes binds against the twin's signature.  This script parses (ast, nothing is imported or executed) every
.py under /root/reference/nr3d_lib that does `import nr3d_lib.bindings._lotd|_pack_ops|_occ_grid as <alias>` and records, per
use of `<alias>.<name>`: the module, the name, file:line, and for calls the number of positional arguments, the keyword names and
whether */** unpacking occurs.  Output: tests/golden/ref_backend_uses.json -- names and counts only, no reference source text.
tests/test_boundary_cpu.py::test_reference_call_sites_bind reads it and binds each record with inspect.signature(...).bind.
"""
import ast
import json
import os
import sys

sys.dont_write_bytecode = True     # (nothing of the reference is imported here; like every generator: no __pycache__ under /root/reference)

REF = "/root/reference"
MODS = ("_lotd", "_pack_ops", "_occ_grid")


def scan(path):
    src = open(path, encoding="utf-8", errors="replace").read()
    try:
        tree = ast.parse(src)
    except SyntaxError:
        return []
    alias = {}                                  # local name -> backend module
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                for m in MODS:
                    if a.name == f"nr3d_lib.bindings.{m}" and a.asname:
                        alias[a.asname] = m
        elif isinstance(node, ast.ImportFrom) and node.module == "nr3d_lib.bindings":
            for a in node.names:
                if a.name in MODS:
                    alias[a.asname or a.name] = a.name
    if not alias:
        return []
    calls = {}                                  # id(Attribute node) -> Call node
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
            calls[id(node.func)] = node
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in alias:
            rec = dict(module=alias[node.value.id], name=node.attr, file=os.path.relpath(path, REF), line=node.lineno)
            c = calls.get(id(node))
            if c is not None:
                rec["call"] = dict(n_positional=sum(not isinstance(a, ast.Starred) for a in c.args),
                                   keywords=[k.arg for k in c.keywords if k.arg is not None],
                                   star_args=any(isinstance(a, ast.Starred) for a in c.args),
                                   star_kwargs=any(k.arg is None for k in c.keywords))
            out.append(rec)
    return out


def main():
    uses = []
    for root, _, files in os.walk(os.path.join(REF, "nr3d_lib")):
        for f in sorted(files):
            if f.endswith(".py"):
                uses += scan(os.path.join(root, f))
    uses.sort(key=lambda r: (r["module"], r["name"], r["file"], r["line"]))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_backend_uses.json")
    json.dump(dict(generator="tests/golden/make_ref_backend_uses.py", n_uses=len(uses),
                   n_names=len({(r["module"], r["name"]) for r in uses}), uses=uses), open(dst, "w"), indent=0)
    print(f"{len(uses)} uses of {len({(r['module'], r['name']) for r in uses})} attributes -> {dst}")


if __name__ == "__main__":
    sys.exit(main())
