#!/usr/bin/env python
"""Regenerates tests/golden/ref_rescale_volume.npz.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rescale.py

Runs the reference's LoTDEncoding.rescale_volume (nr3d_lib/models/grid_encodings/lotd/lotd_encoding.py:329-401, pure
PyTorch on top of lotd_helpers) on seeded tables.  The reference class cannot be constructed here (its meta comes from the
CUDA extension), so its unbound methods are called on a stand-in object that carries the attributes they read: the level
layout (an nr3d_lib_amd LoDMeta has the same field names and, pinned elsewhere, the same values), the flat parameter
vector, the device and the old AABB.  Stored: layout arguments, parameters before / after, both AABBs (data only)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from make_golden import import_reference        # noqa: E402

RES, FEATS, TYPES = [7, 7, 7, 7, 7], [2, 4, 2, 2, 4], ["Dense", "VM", "NPlaneMul", "CP", "NPlaneSum"]


def main():
    enc = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_encoding")
    helpers = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_helpers")

    class LT:        # the reference keys its LoDType enum on the pybind values, which are stubs here: patch the lookup
        Dense, VectorMatrix, CP, CPfast, NPlaneMul, NPlaneSum, Hash = 0, 1, 3, 4, 5, 6, 7
        def __new__(cls, v): return int(v)
    enc.LoDType = helpers.LoDType = LT
    from nr3d_lib_amd.bindings import _lotd
    meta = _lotd.LoDMeta(3, RES, FEATS, TYPES, None)
    g = torch.Generator().manual_seed(9)
    params = torch.randn(meta.n_params, generator=g)
    old = torch.tensor([[-1.0, -2.0, -0.5], [1.0, 2.0, 1.5]])
    new = torch.tensor([[-0.6, -1.1, 0.0], [0.7, 1.5, 1.2]])
    me = types.SimpleNamespace(lod_meta=meta, flattened_params=params.clone(), device=torch.device("cpu"),
                               space=types.SimpleNamespace(aabb=old))
    cls = enc.LoTDEncoding
    me.get_level_param = types.MethodType(cls.get_level_param, me)
    me.set_level_param = types.MethodType(cls.set_level_param, me)
    cls.rescale_volume(me, new.clone())
    np.savez_compressed(os.path.join(HERE, "ref_rescale_volume.npz"), res=np.array(RES), feats=np.array(FEATS),
                        types=np.array(TYPES), before=params.numpy(), after=me.flattened_params.numpy(),
                        old_aabb=old.numpy(), new_aabb=new.numpy())
    print("changed entries:", int((me.flattened_params != params).sum()), "of", meta.n_params)


if __name__ == "__main__":
    main()
