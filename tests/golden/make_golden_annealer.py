#!/usr/bin/env python
"""Regenerates tests/golden/ref_annealer.json.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_annealer.py

Runs the reference's MultiresAnnealer (nr3d_lib/models/grid_encodings/multires_annealer.py) with the three configurations
of its own self-test (:82-91) plus two more over iterations 0..999 and stores (max_level, window) per iteration.  The
reference keeps its per-level repeat counts in the output dtype, which torch.repeat_interleave rejects, so for the cosine
configurations that buffer is replaced by its integer values before calling (noted in the product's docstring)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference        # noqa: E402

CONFIGS = [
    dict(level_n_feats=[2, 4, 2, 3, 5], type='hardmask', start_it=10, stop_it=433, start_level=1),
    dict(level_n_feats=[2, 4, 2, 3, 5], type='hardmask', start_it=10, stop_it=334, start_level=0),
    dict(level_n_feats=[1] * 13, type='cosine', start_it=100, stop_it=443, start_level=2),
    dict(level_n_feats=[2] * 16, type='hardmask', start_it=0, stop_it=900, update_every=50, start_level=-1),
    dict(level_n_feats=[2, 4, 8, 2], type='cosine', start_it=0, stop_it=600, update_every=7, start_level=-3),
]


def main():
    mod = import_reference("nr3d_lib.models.grid_encodings.multires_annealer")
    out = []
    for cfg in CONFIGS:
        a = mod.MultiresAnnealer(**cfg, device='cpu')
        if cfg['type'] == 'cosine':
            a.level_n_feats = a.level_n_feats.long()
        levels, windows = [], {}
        for it in range(1000):
            ml, w = a(it)
            levels.append(int(ml))
            if w is not None and it % 37 == 0:
                windows[str(it)] = [round(float(v), 7) for v in w]
        ml_default, _ = a()
        out.append(dict(cfg=cfg, max_level=levels, windows=windows, default_max_level=int(ml_default)))
    json.dump(out, open(os.path.join(HERE, "ref_annealer.json"), "w"))
    print("wrote", len(out), "configs")


if __name__ == "__main__":
    main()
