#!/usr/bin/env python
"""Regenerates tests/golden/ref_table_interpolate.npz.  BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tables.py

Pins the product-type LoTD levels (VM, CP) against the reference's own pure-PyTorch building blocks, the way its
`rescale_volume` uses them (lotd_encoding.py:350-402): every line / plane table of a level is sliced with
`lotd_helpers.level_param_index_shape(meta, l, 'vec' | 'mat', dim)` and sampled with `lotd_helpers.param_interpolate`
(1-D / 2-D `grid_sample`).  What the fixture stores per level type: the level's parameters, the points, and the
sampled value of EVERY table at every point.  The test combines them with the level's definition (VM: sum_d
plane_d * line_d, CP: prod_d line_d -- trilinear interpolation of a product of per-axis factors is the product of the
per-axis interpolations) and compares with the oracle's forward.  Data only; nothing of the reference's text is stored."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import numpy as np
import torch

from make_golden import ROOT, import_reference   # noqa: E402  (also puts the repo and tests/ on sys.path)


def main():
    import oracle
    oracle.build()
    helpers = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_helpers")

    class _LT:   # the reference keys LoDType on its pybind enum (a stub here)
        Dense, VectorMatrix, CP, CPfast, NPlaneMul, NPlaneSum, Hash = 0, 1, 3, 4, 5, 6, 7
        def __new__(cls, v): return v
    helpers.LoDType = _LT

    torch.manual_seed(5)
    R, F, n = 11, 4, 2048
    out = dict(res=np.int64(R), feats=np.int64(F))
    x = torch.rand(n, 3).clamp(1e-6, 1 - 1e-6)
    out["x"] = x.numpy()
    rel = x * 2 - 1
    for name, tp in (("vm", "VM"), ("cp", "CP")):
        m = oracle.lotd_create_meta(3, [R], [F], [tp]).as_dict()
        meta = type("M", (), dict(m))()
        params = torch.randn(m["n_params"])
        out[f"{name}_params"] = params.numpy()
        for d in range(3):
            index, shape = helpers.level_param_index_shape(meta, 0, "vec", d)
            line = params[index].view(shape)                                   # [R, F]
            out[f"{name}_line{d}_start"] = np.int64(index[0].start)
            out[f"{name}_line{d}"] = helpers.param_interpolate(line.view(1, R, F), rel[:, d].reshape(1, n, 1), R)[0].numpy()
            if tp == "VM":
                index, shape = helpers.level_param_index_shape(meta, 0, "mat", d)
                plane = params[index].view(shape)                              # [R, R, F] over the two dims != d
                ab = [k for k in range(3) if k != d]
                out[f"{name}_plane{d}_start"] = np.int64(index[0].start)
                out[f"{name}_plane{d}"] = helpers.param_interpolate(plane.view(1, R, R, F),
                                                                    rel[:, ab].reshape(1, n, 2), R)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "ref_table_interpolate.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
