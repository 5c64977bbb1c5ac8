#!/usr/bin/env python
"""Regenerates tests/golden/ref_lotd_helpers.npz.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_helpers.py

Calls the reference's pure-PyTorch helpers nr3d_lib/models/grid_encodings/lotd/lotd_helpers.py (param_vertices :244-266,
param_interpolate :274-346) on seeded inputs and stores inputs + outputs (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference        # noqa: E402


def main():
    h = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_helpers")
    out = {}
    for tag, res, dim in (("v1", 5, 1), ("v2", 6, 2), ("v3", 4, 3), ("vc", [4, 6, 5], 3)):
        for forest in (False, True):
            out[f"{tag}_{int(forest)}"] = h.param_vertices(res, dim, is_forest=forest, device="cpu").numpy()
    g = torch.Generator().manual_seed(5)
    R, M, B = 6, 3, 2
    for d in (1, 2, 3):
        param = torch.randn(B, *([R] * d), M, generator=g)
        x = torch.rand(B, 7, 5, d, generator=g) * 2.4 - 1.2          # some points outside [-1, 1]: zero padding
        out[f"i{d}_param"], out[f"i{d}_x"] = param.numpy(), x.numpy()
        for forest in (False, True):
            out[f"i{d}_y{int(forest)}"] = h.param_interpolate(param, x, R, forest).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_lotd_helpers.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
