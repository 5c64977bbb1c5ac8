#!/usr/bin/env python
"""Regenerates tests/golden/*.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Two kinds of fixtures (data only -- inputs and expected outputs):
  ref_*.npz / ref_*.json   produced by IMPORTING the reference's pure-Python code (with stub modules for its
                           compiled extensions and absent pip packages) and calling it:
                             - lotd_helpers.param_interpolate   (Dense LoTD level == config C1)
                             - lotd_cfg.gen_ngp_cfg             (level ladder of configs C2 / C5)
                             - lotd_helpers.level_param_index_shape (parameter layout of every level type)
                             - pack_ops.py pure-torch helpers   (pack_infos builders, intersect, batch merge, matmul)
                             - nerf_utils.tau_to_alpha
  oracle_digest.json       sha256 of the CPU oracle's outputs on seeded inputs: pins the oracle itself against
                           accidental edits (the oracle is what the GPU kernels are compared with).
Nothing from the reference's source text is stored.
"""
import hashlib
import importlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

import numpy as np
import torch


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k == "__path__":
            return []
        if k.startswith("__"):
            raise AttributeError(k)
        return _Stub(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return self

    def __int__(self):
        return 0

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


def import_reference(name):
    sys.path.insert(0, "/root/reference")
    for _ in range(80):
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            parts = e.name.split(".")
            for k in range(1, len(parts) + 1):
                n = ".".join(parts[:k])
                if n == "nr3d_lib" or (n.startswith("nr3d_lib") and k < len(parts)):
                    continue
                if n not in sys.modules or k == len(parts):
                    sys.modules[n] = _Stub(n)
            for k in [k for k in sys.modules if k.startswith("nr3d_lib") and not isinstance(sys.modules[k], _Stub)]:
                del sys.modules[k]
    raise RuntimeError("could not import " + name)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def main():
    import oracle
    from util import LOTD_CASES, lotd_inputs, random_packs
    oracle.build()

    helpers = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_helpers")
    cfgmod = import_reference("nr3d_lib.models.grid_encodings.lotd.lotd_cfg")
    pk = import_reference("nr3d_lib.graphics.pack_ops.pack_ops")
    nerf_utils = import_reference("nr3d_lib.graphics.nerf.nerf_utils")

    # ---- C1: Dense level through the reference's param_interpolate ------------------------------------
    torch.manual_seed(0)
    out = {}
    for tag, R, F, n, D in (("c1", 32, 4, 65536, 3), ("d2", 19, 2, 4096, 2)):
        param = torch.randn(R ** D * F)
        x = torch.rand(n, D).clamp(1e-6, 1 - 1e-6)
        ref = helpers.param_interpolate(param.view(1, *([R] * D), F), (x * 2 - 1).view(1, -1, D), R)[0]
        keep = slice(0, n)      # every point: configs[0] runs at its stated 65 536 points (1.8 MB of x + y)
        out[f"{tag}_res"] = np.int64(R); out[f"{tag}_feats"] = np.int64(F)
        out[f"{tag}_params"] = param.numpy().astype(np.float16 if tag == "c1" else np.float32)
        out[f"{tag}_x"] = x.numpy()[keep]
        if tag == "c1":   # fp16-rounded params so the file stays small; recompute the reference on them
            param = torch.from_numpy(out[f"{tag}_params"].astype(np.float32))
            ref = helpers.param_interpolate(param.view(1, *([R] * D), F), (x * 2 - 1).view(1, -1, D), R)[0]
        out[f"{tag}_y"] = ref.numpy()[keep]
    np.savez_compressed(os.path.join(HERE, "ref_param_interpolate.npz"), **out)

    # ---- gen_ngp_cfg ---------------------------------------------------------------------------------
    cfgs = {}
    for name, kw in {"default": {}, "small": dict(min_res=4, log2_hashmap_size=12, num_levels=8),
                     "dim2": dict(dim=2, n_feats=4, log2_hashmap_size=14, num_levels=10, per_level_scale=1.5),
                     "t22": dict(log2_hashmap_size=22, num_levels=12, per_level_scale=2.0)}.items():
        cfgs[name] = dict(kwargs=kw, cfg=cfgmod.gen_ngp_cfg(**kw))
    # ---- parameter layout (level_param_index_shape) ----------------------------------------------------
    layout = []
    for case in ("mixed", "mixed_cuboid", "nplane", "ngp_small"):
        D, res, nf, types, T, smooth = LOTD_CASES[case]
        m = oracle.lotd_create_meta(D, res, nf, types, T, smooth).as_dict()
        meta = types_ns = type("M", (), dict(m))()
        ref_types = {0: "Dense", 1: "VectorMatrix", 3: "CP", 4: "CPfast", 5: "NPlaneMul", 6: "NPlaneSum", 7: "Hash"}
        for l in range(m["n_levels"]):
            if m["level_types"][l] == 2:
                continue   # VecZMatXoY is not in the reference's Python enum
            # the reference keys its LoDType enum on the pybind values, which are stubs here: patch the lookup
            helpers.LoDType = type("LT", (), {})
            class _LT:
                Dense, VectorMatrix, CP, CPfast, NPlaneMul, NPlaneSum, Hash = 0, 1, 3, 4, 5, 6, 7
                def __new__(cls, v): return v
            helpers.LoDType = _LT
            for op, dims in ((None, [None]), ("line", [None, 0, 1, 2]), ("plane", [None, 0, 1, 2]), ("vol", [None])):
                for dim in dims:
                    try:
                        index, shape = helpers.level_param_index_shape(meta, l, op, dim)
                    except Exception:
                        continue
                    layout.append(dict(case=case, level=l, op=op, dim=dim, start=int(index[0].start),
                                       stop=int(index[0].stop), shape=[int(s) for s in shape]))
    # ---- pack_ops pure-torch helpers -------------------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    boundary = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0], dtype=torch.bool)
    n_per = torch.tensor([4, 1, 5, 3])
    t1 = torch.tensor([1, 3, 4, 7, 9, 12]); t2 = torch.tensor([0, 3, 5, 7, 12, 13, 20])
    inter = [t.tolist() for t in pk.torch_intersect1d_unique(t1, t2)]
    va = torch.rand(6, 5, generator=g).sort(-1)[0]; vb = torch.rand(3, 4, generator=g).sort(-1)[0]
    na = torch.tensor([2, 5, 6, 9, 11, 14]); nb_ = torch.tensor([5, 9, 14])
    pa, pb, pi = pk.merge_two_batch_a_includes_b(va, na, vb, nb_)
    feats = torch.rand(13, 3, generator=g); other = torch.rand(3, 2, 3, generator=g)
    pinfo = pk.get_pack_infos_from_boundary(boundary)
    helpers_out = dict(
        from_boundary=pinfo.tolist(), from_first=pk.get_pack_infos_from_first(torch.tensor([0, 4, 11]), 13).tolist(),
        from_n=pk.get_pack_infos_from_n(n_per).tolist(), from_batch=pk.get_pack_infos_from_batch(3, 5).tolist(),
        expand=pk.expand_pack_boundary(boundary, 3).long().tolist(), intersect=inter,
        merge_batch=dict(va=va.tolist(), vb=vb.tolist(), na=na.tolist(), nb=nb_.tolist(), pidx_a=pa.tolist(),
                         pidx_b=pb.tolist(), pack_infos=pi.tolist()),
        matmul=dict(feats=feats.tolist(), other=other.tolist(), out=pk.packed_matmul(feats, other, pinfo).tolist()),
        tau_to_alpha=dict(tau=[0.0, 0.1, 1.0, 5.0], alpha=nerf_utils.tau_to_alpha(torch.tensor([0.0, 0.1, 1.0, 5.0])).tolist()),
    )
    json.dump(dict(gen_ngp_cfg=cfgs, layout=layout, pack_helpers=helpers_out),
              open(os.path.join(HERE, "ref_python.json"), "w"), indent=1)

    # ---- oracle digests --------------------------------------------------------------------------------
    dig = {}
    for case in LOTD_CASES:
        D, res, nf, types, T, smooth = LOTD_CASES[case]
        m = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
        x, p, g_, v = lotd_inputs(m.as_dict(), 257, 123)
        y, j = oracle.lotd_fwd(m, x, p, need_dydx=True)
        d = dict(y=digest(y), dy_dx=digest(j), dx=digest(oracle.lotd_bwd_dx(m, g_, j)),
                 dparam=digest(oracle.lotd_bwd_dparam(m, g_, x, p)),
                 ddy=digest(oracle.lotd_bwd_bwd_ddLdy(m, v, j)),
                 dparam2=digest(oracle.lotd_bwd_bwd_dparam(m, v, g_, x, p)),
                 dx2=digest(oracle.lotd_bwd_bwd_dx(m, v, g_, x, p)))
        if all(t in (0, 7) for t in m.as_dict()["level_types"]):
            d["grid_index"] = digest(oracle.lotd_grid_index(m, x))
        dig["lotd/" + case] = d
    rng = np.random.default_rng(5)
    o = np.tile(np.array([0.1, -0.2, -3], np.float32), (64, 1))
    dd = np.stack([rng.uniform(-.3, .3, 64), rng.uniform(-.3, .3, 64), np.ones(64)], 1).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    grid = rng.random((16, 20, 12)) > 0.5
    roi = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    for ct in (0, 1, 2):
        r = oracle.ray_marching(o, dd, np.full(64, 1.5, np.float32), np.full(64, 4.5, np.float32), roi, grid, ct, 0.03,
                                1e10, 0.01, 64, True)
        dig[f"march/{ct}"] = digest(*r)
    pi, S = random_packs(rng, 40, 0, 60, 0.1)
    a = (rng.uniform(0, 1, S) ** 2).astype(np.float32); gw = rng.standard_normal(S).astype(np.float32)
    w = oracle.packed_alpha_to_vw_forward(a, pi, 1e-4, 0.0, False)[0]
    dig["pack/alpha"] = digest(w, oracle.packed_alpha_to_vw_backward(w, gw, a, pi, 1e-4, 0.0),
                               *oracle.packed_alpha_to_vw_forward(a, pi, 1e-3, 0.05, True)[1:])
    dig["pack/scan"] = digest(oracle.packed_sum(a, pi), oracle.packed_cumsum(a, pi, True, True),
                              oracle.packed_cumprod(a, pi), oracle.packed_diff(a, pi), oracle.packed_backward_diff(a, pi))
    dig["pack/sample"] = digest(*oracle.interleave_sample_step_wrt_depth_clamped(
        rng.uniform(0.1, 1, 50).astype(np.float32), rng.uniform(1, 4, 50).astype(np.float32), 64, 0.02, 0.01, 0.5))
    json.dump(dig, open(os.path.join(HERE, "oracle_digest.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
