"""Shared helpers for the parity tests."""
import numpy as np
import torch

# ---- tolerance of the fp32 parity contract (BASELINE.json north_star: "within 1e-5 rel fp32") -------------
# Compared quantity: max |got - want| relative to max |want| (sums with cancellation make an elementwise
# relative error meaningless); elementwise rtol for well-conditioned outputs.
REL_TOL = 1e-5


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def assert_close(got, want, rel=REL_TOL, name="", levels=None):
    """max |got - want| <= rel * scale, the scale taken PER OUTPUT GROUP, not over the whole tensor:
      * >= 2-D outputs ([N, E] values, [N, E, D] Jacobians, [N, D] input gradients, [S, F] packed features): one scale
        per index of axis 1 (per encoded column / per input dim), max |want| over the other axes;
      * parameter-sized 1-D outputs with ``levels=`` (a LoDMeta, the oracle's meta or its as_dict()): one scale per
        level's slice of every table set, so a fine hash level with small gradients is held to 1e-5 of ITS magnitude;
      * other 1-D outputs: one global scale."""
    got, want = _np(got), np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    if want.size == 0:
        return
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    mag = np.abs(want.astype(np.float64))
    if levels is not None and want.ndim == 1:
        d = levels if isinstance(levels, dict) else (levels.as_dict() if hasattr(levels, "as_dict") else
                                                     dict(level_offsets=list(levels.level_offsets), n_params=int(levels.n_params)))
        offs, n_params = list(d["level_offsets"]), int(d["n_params"])
        assert want.size % n_params == 0, f"{name}: {want.size} elements is not a multiple of n_params={n_params}"
        e2, m2 = err.reshape(-1, n_params), mag.reshape(-1, n_params)
        for lvl, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
            scale = max(float(m2[:, a:b].max()) if b > a else 0.0, 1e-30)
            worst = float(e2[:, a:b].max()) if b > a else 0.0
            assert worst <= rel * scale, f"{name}: level {lvl}: max abs err {worst:.3e} > {rel:g} * max|ref| of the level ({scale:.3e})"
        return
    if want.ndim >= 2:
        axes = tuple(a for a in range(want.ndim) if a != 1)
        scale = np.maximum(mag.max(axis=axes), 1e-30)
        worst = err.max(axis=axes)
        bad = np.nonzero(worst > rel * scale)[0]
        assert bad.size == 0, (f"{name}: column {int(bad[0])}: max abs err {worst[bad[0]]:.3e} > {rel:g} * max|ref| of the "
                               f"column ({scale[bad[0]]:.3e}); {bad.size} of {scale.size} columns fail")
        return
    scale = max(float(mag.max()), 1e-30)
    assert float(err.max()) <= rel * scale, f"{name}: max abs err {float(err.max()):.3e} > {rel:g} * max|ref| ({scale:.3e})"


def assert_equal(got, want, name=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} of {want.size} entries differ"


# ---- LoTD test metas ------------------------------------------------------------------------------------
LOTD_CASES = {
    # name: (D, lod_res, n_feats, types, hashmap_size, smoothstep)
    "ngp_small": (3, [8, 11, 15, 21, 29, 40, 55, 76], [2] * 8, ["Dense"] * 4 + ["Hash"] * 4, 2 ** 12, False),
    "ngp_smooth": (3, [8, 13, 21, 34, 55], [2] * 5, ["Dense", "Dense", "Hash", "Hash", "Hash"], 2 ** 11, True),
    # several table buckets per level on the pair-record parameter-gradient path (Dense 40^3: 13 row buckets, Hash 2^16: 8)
    "ngp_pair": (3, [10, 24, 40, 64, 90, 130], [2] * 6, ["Dense"] * 3 + ["Hash"] * 3, 2 ** 16, False),
    # 4-feature levels split into two 2-feature pseudo levels (gcd 2), cuboid Dense, hash table of 2 buckets
    "pair_f4": (3, [[12, 9, 14], [20, 33, 17], [33, 33, 33]], [2, 4, 4], ["Dense", "Dense", "Hash"], 2 ** 14, False),
    "hash_npow2": (3, [9, 17, 33], [4, 4, 4], ["Dense", "Hash", "Hash"], 3001, False),
    "dense_f8": (3, [6, 9], [8, 8], ["Dense", "Dense"], None, False),
    "dense_2d": (2, [16, 33], [2, 4], ["Dense", "Hash"], 257, False),
    "hash_4d": (4, [5, 9], [2, 2], ["Dense", "Hash"], 2 ** 10, False),
    "mixed": (3, [8, 12, 10, 14, 9, 16, 11, 13], [4, 4, 8, 4, 2, 16, 8, 4],
              ["Dense", "Dense", "VM", "VM", "VM", "CP", "CP", "CP"], None, False),
    "mixed_cuboid": (3, [[8, 6, 5], [12, 9, 7], [10, 8, 6], [14, 11, 9], [9, 7, 6], [16, 12, 8]],
                     [4, 4, 8, 4, 2, 16], ["Dense", "Dense", "VM", "VM", "CP", "CP"], None, False),
    "mixed_smooth": (3, [7, 9, 8, 10], [2, 2, 2, 2], ["Dense", "VM", "CP", "Hash"], 509, True),
    "vecz_nplanemul": (3, [9, 8, 7], [2, 2, 4], ["VecZMatXoY", "NPlaneMul", "VecZMatXoY"], None, False),
    "nplane": (3, [8, 9, 7, 10], [2, 4, 2, 2], ["NPlaneMul", "NPlaneSum", "CPfast", "VecZMatXoY"], None, False),
    "nplane_smooth": (3, [8, 9, 7], [2, 2, 4], ["NPlaneSum", "CPfast", "NPlaneMul"], None, True),
    "cp_2d": (2, [9, 12], [2, 2], ["CP", "CPfast"], None, False),
    "cp_only_2d4d": (2, [9, 12], [2, 4], ["CP", "NPlaneMul"], None, False),
    "cp_only_4d": (4, [5, 6], [2, 2], ["CP", "Dense"], None, False),
    "cp_4d": (4, [5, 6, 4], [2, 2, 2], ["CP", "NPlaneMul", "CPfast"], None, False),
    "nplane_4d": (4, [5, 6, 4], [2, 4, 2], ["NPlaneSum", "NPlaneMul", "Dense"], None, True),
    # 4-D with 8-feature pseudo levels: the parameter-gradient stage A runs on the width-4 regrouping (16-corner records x 8 features
    # do not fit the register file: lotd_bin.hip stage_a_meta)
    "nplane_4d_f8": (4, [5, 6, 4], [8, 16, 8], ["NPlaneMul", "Hash", "Dense"], 2 ** 9, False),
}


def lotd_inputs(meta_dict, n_points, seed, param_scale=0.1, n_batch=1):
    """seeded inputs kept away from exact cell boundaries (x*(R-2)+0.5 never within 1e-3 of an integer) so that
    fp32 last-bit differences cannot move a point into another cell"""
    rng = np.random.default_rng(seed)
    D = meta_dict["n_dims_to_encode"]
    x = rng.random((n_points, D)).astype(np.float32).clip(1e-6, 1 - 1e-6)
    for _ in range(8):
        bad = np.zeros(n_points, bool)
        for res in meta_dict["level_res_multidim"]:
            v = x.astype(np.float64) * (np.array(res) - 2) + 0.5
            bad |= (np.abs(v - np.round(v)) < 1e-3).any(1)
        if not bad.any():
            break
        x[bad] = rng.random((int(bad.sum()), D)).astype(np.float32).clip(1e-6, 1 - 1e-6)
    params = (rng.standard_normal(meta_dict["n_params"] * n_batch) * param_scale).astype(np.float32)
    dL_dy = (rng.standard_normal((n_points, meta_dict["n_encoded_dims"])) * 0.1).astype(np.float32)
    dL_ddLdx = rng.standard_normal((n_points, D)).astype(np.float32)
    return x, params, dL_dy, dL_ddLdx


def random_packs(rng, n_packs, lo, hi, empty_frac=0.0):
    n = rng.integers(lo, hi + 1, n_packs).astype(np.int64)
    if empty_frac > 0:
        n[rng.random(n_packs) < empty_frac] = 0
    cs = np.cumsum(n)
    return np.ascontiguousarray(np.stack([cs - n, n], 1)), int(cs[-1])
