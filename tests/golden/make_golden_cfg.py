#!/usr/bin/env python
"""Regenerates tests/golden/ref_lotd_cfg.json.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cfg.py

Calls the reference's level-layout generators (nr3d_lib/models/grid_encodings/lotd/lotd_cfg.py: get_lotd_cfg with the
types gen_ngp / single_res / ngp / ngp4d) on a list of argument sets and stores arguments + returned dictionaries
(data only; nothing of the reference's source text)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference        # noqa: E402  (stubs the reference's absent third-party modules)

CASES = [
    dict(type="gen_ngp", input_ch=3),
    dict(type="gen_ngp", input_ch=2, min_res=8, n_feats=4, log2_hashmap_size=14, per_level_scale=1.5, num_levels=9),
    dict(type="gen_ngp", input_ch=4, min_res=4, log2_hashmap_size=16, num_levels=10),
    dict(type="single_res", input_ch=3, stretch=[10.0, 7.5, 3.3], voxel_size=0.4),
    dict(type="single_res", input_ch=3, stretch=[2.0, 2.0, 2.0], voxel_size=0.13, n_feats=4, lotd_type="VM"),
    dict(type="ngp", input_ch=3, stretch=1.0, target_num_params=2 ** 24),
    dict(type="ngp", input_ch=3, stretch=[1.0, 1.0, 1.0], target_num_params=32 * 2 ** 20),
    dict(type="ngp", input_ch=3, stretch=[3.0, 2.0, 1.0], target_num_params=2 ** 25, log2_hashmap_size=20),
    dict(type="ngp", input_ch=3, stretch=[120.0, 45.5, 12.25], target_num_params=12_000_000, n_feats=4, min_res=6,
         per_level_scale=1.5),
    dict(type="ngp", input_ch=3, stretch=[1.0, 4.0, 2.5], target_num_params=2 ** 26, max_num_levels=12),
    dict(type="ngp", input_ch=3, stretch=2.0, target_num_params=2 ** 20, log2_hashmap_size=15, per_level_scale=2.0),
    dict(type="ngp", input_ch=2, stretch=[1.0, 2.0], target_num_params=2 ** 23, log2_hashmap_size=17),
    dict(type="ngp4d", input_ch=3, stretch=[1.0, 1.0, 1.0], target_num_params=2 ** 24),
    dict(type="ngp4d", input_ch=3, stretch=[2.0, 1.0, 1.5], target_num_params=2 ** 26, min_dense_layers=3,
         log2_hashmap_size=18, min_res_xyz=6, min_res_w=2, per_level_scale=1.3),
    dict(type="ngp4d", input_ch=3, stretch=[1.0, 3.0, 1.0], target_num_params=5_000_000, max_layers=6, n_feats=4),
]


def main():
    cfg = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_cfg")
    out = []
    for kw in CASES:
        got = cfg.get_lotd_cfg(**kw)
        out.append(dict(args=kw, result=json.loads(json.dumps(got, default=lambda o: o.tolist() if hasattr(o, "tolist") else int(o)))))
    with open(os.path.join(HERE, "ref_lotd_cfg.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
