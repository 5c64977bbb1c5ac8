"""Randomised bit-exactness check of the occupancy-grid marcher (single + batched, all contraction types, random ROIs /
resolutions / step laws / ray bundles, sample cache on and off) against the CPU oracle.  usage: fuzz_march.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                   # noqa: E402  (test infrastructure)
from nr3d_lib_amd.bindings import _occ_grid                      # noqa: E402
from nr3d_lib_amd import _hip                                  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["packed_info", "t_starts", "t_ends", "ridx", "bidx", "gidx"]


def one(rng):
    batched = rng.random() < 0.3
    ctype = int(rng.choice([0, 0, 0, 1, 2]))
    res = tuple(int(v) for v in rng.choice([4, 7, 8, 16, 31, 32, 64], 3))
    B = int(rng.integers(1, 4)) if batched else 1
    pow2 = rng.random() < 0.5
    if pow2:
        half = float(rng.choice([0.5, 1.0, 2.0]))
        lo = np.array([-half] * 3, np.float32) + np.float32(rng.choice([0.0, 0.25]))
        hi = lo + 2 * half
    else:
        lo = (rng.random(3) * 0.5 - 1.2).astype(np.float32)
        hi = (lo + 1.5 + rng.random(3)).astype(np.float32)
    roi1 = np.concatenate([lo, hi]).astype(np.float32)
    roi = np.tile(roi1, (B, 1)) if batched else roi1
    grid = rng.random((B, *res) if batched else res) > float(rng.choice([0.2, 0.5, 0.9]))
    n = int(rng.choice([1, 3, 64, 129, 1000, 5000]))
    if batched and rng.random() < 0.5:
        n = B * int(rng.choice([1, 17, 200]))
    o = ((rng.random((n, 3)) - 0.5) * 6).astype(np.float32)
    tgt = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = tgt - o
    d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    if rng.random() < 0.2:
        d[rng.integers(0, n), rng.integers(0, 3)] = 0.0             # axis-parallel component
    near = (rng.random(n) * 0.5).astype(np.float32)
    far = (near + rng.random(n) * 8).astype(np.float32)
    step = float(rng.choice([0.01, 0.02, 0.05, 0.003]))
    gamma = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
    max_step = float(rng.choice([1e10, 0.1, 0.05]))
    max_steps = int(rng.choice([1, 8, 64, 512]))
    kw = {}
    if batched:
        if n % B == 0 and rng.random() < 0.5:
            kw = dict(batch_inds=None, batch_data_size=n // B)
        else:
            bi = rng.integers(-1 if rng.random() < 0.3 else 0, B, n).astype(np.int32)
            kw = dict(batch_inds=bi, batch_data_size=0)
    ref = oracle.ray_marching(o, d, near, far, roi, grid, ctype, np.float32(step), max_step, gamma, max_steps, True, **kw)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    _occ_grid.SAMPLE_CACHE_MAX_BYTES = (2 << 30) if rng.random() < 0.6 else 0
    lanes = int(rng.choice([1, 16, 32, 64]))                                           # lanes per ray (count pass with the cache)
    _hip.set_option("march_group", lanes)
    if batched:
        got = _occ_grid.batched_ray_marching(t(o), t(d), t(near), t(far), t(kw["batch_inds"]), kw["batch_data_size"], t(roi), t(grid),
                                             _occ_grid.ContractionType(ctype), step, max_step, gamma, max_steps, True)
        names = NAMES
    else:
        got = _occ_grid.ray_marching(t(o), t(d), t(near), t(far), t(roi), t(grid), _occ_grid.ContractionType(ctype), step,
                                     max_step, gamma, max_steps, True)
        names = [k for k in NAMES if k != "bidx"]
    if ctype != 0:
        # tanh / sqrt of the contractions come from libm here and from the device library there: a probe that lands
        # within an ulp of a voxel face may be classified differently, so only the sample totals are compared
        a, b = int(got[0][:, 1].sum()), int(np.asarray(ref[0])[:, 1].sum())
        assert abs(a - b) <= max(2, 0.005 * max(a, b)), f"march totals differ: {a} vs {b} (ctype={ctype} res={res} n={n})"
        return b
    for g, r, name in zip(got, ref, names):
        g = g.cpu().numpy()
        r = np.asarray(r)
        ok = g.shape == r.shape and np.array_equal(g, r)
        assert ok, (f"march mismatch in {name}: batched={batched} ctype={ctype} res={res} roi={roi1.tolist()} n={n} step={step} "
                    f"gamma={gamma} max_step={max_step} max_steps={max_steps} cache={_occ_grid.SAMPLE_CACHE_MAX_BYTES} lanes={lanes}")
    return int(ref[1].shape[0])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n, samples, t0 = 0, 0, time.time()
    while time.time() - t0 < budget:
        samples += one(rng)
        n += 1
    print({"march configs ok": n, "samples compared": samples})


if __name__ == "__main__":
    main()
