#!/usr/bin/env python
"""Regenerates tests/golden/ref_scalar_annealers.json.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_scalar_annealers.py

Evaluates the reference's scalar schedules (nr3d_lib/models/annealers.py: get_anneal_val and get_annealer) at iterations
0..1100 step 7 for a list of configurations; stores configuration + values (data only)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference        # noqa: E402

ITS = list(range(0, 1101, 7))
COMMON = dict(start_val=0.9, stop_val=0.25, start_it=10, stop_it=1000, update_every=5)     # the reference's own self-test
FUNCS = [
    dict(type='linear', **COMMON), dict(type='logspace', **COMMON),
    dict(type='linear', stop_it=500), dict(type='logspace', stop_it=300, start_val=10.0, stop_val=800.0, update_every=50),
    dict(type='milestones', milestones=[100, 300, 600], vals=[0.1, 0.2, 0.3, 0.4]),
]
OBJS = FUNCS + [dict(type='constant', val=3.5),
                dict(type='partitions', partition_cfgs=[dict(type='linear', stop_it=200, start_val=0.0, stop_val=1.0),
                                                        dict(type='linear', stop_it=400, start_val=1.0, stop_val=1.0),
                                                        dict(type='logspace', stop_it=900, start_val=1.0, stop_val=64.0),
                                                        dict(type='linear', stop_it=10 ** 9, start_val=64.0, stop_val=64.0,
                                                             update_every=1000)])]


def main():
    mod = import_reference("nr3d_lib.models.annealers")

    class AttrDict(dict):                   # the reference reads `cfg.stop_it`: its partitions are attribute dicts
        __getattr__ = dict.__getitem__
    out = dict(its=ITS, functional=[], objects=[])
    for cfg in FUNCS:
        out["functional"].append(dict(cfg=cfg, vals=[float(mod.get_anneal_val(it=it, **cfg)) for it in ITS]))
    for cfg in OBJS:
        c = json.loads(json.dumps(cfg))
        if c['type'] == 'partitions':
            c['partition_cfgs'] = [AttrDict(p) for p in c['partition_cfgs']]
        try:
            a = mod.get_annealer(**c)
            vals = []
            for it in ITS:
                a.set_iter(it)
                try:
                    vals.append(float(a.get_val()))
                except NotImplementedError:         # the reference's milestones object only has the call form
                    vals.append(float(a(it)))
            out["objects"].append(dict(cfg=cfg, vals=vals))
        except Exception as ex:
            out["objects"].append(dict(cfg=cfg, error=repr(ex)[:200]))
    json.dump(out, open(os.path.join(HERE, "ref_scalar_annealers.json"), "w"))
    print([("err" if "error" in o else "ok") for o in out["objects"]])


if __name__ == "__main__":
    main()
