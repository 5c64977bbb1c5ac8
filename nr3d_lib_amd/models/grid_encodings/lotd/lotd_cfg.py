"""LoTD level-layout generators: what ``LoTDEncoding(lotd_auto_compute_cfg=...)`` resolves its level ladder with.

Counterpart of the reference's nr3d_lib/models/grid_encodings/lotd/lotd_cfg.py: ``get_lotd_cfg`` (:21-37) with the
generators 'gen_ngp' (:48-57, the instant-ngp ladder of BASELINE configs 2 and 5), 'single_res' (:39-46), 'ngp'
(:59-133, Dense -> Hash ladder sized to a parameter budget for a cuboid of the given aspect) and 'ngp4d' (:135-194).
The reference's own deprecated 'lotd' generator (:196-) is not provided.  Pure host arithmetic; the outputs are pinned
against the reference's functions in tests/golden/ref_lotd_cfg.json (tests/test_golden_cfg_cpu.py).
"""
from numbers import Number

import numpy as np

__all__ = ['get_lotd_cfg', 'gen_ngp_cfg', 'single_res_cfg', 'auto_ngp_cfg', 'auto_ngp4d_cfg']


def gen_ngp_cfg(min_res: int = 16, dim: int = 3, n_feats: int = 2, log2_hashmap_size: int = 19,
                per_level_scale: float = 1.382, num_levels: int = 16) -> dict:
    """Resolutions min_res * per_level_scale**l (truncated); a level is Dense while its full grid fits
    the hash table (res**dim <= 2**log2_hashmap_size), Hash afterwards."""
    table = 2 ** log2_hashmap_size
    res = (min_res * per_level_scale ** np.arange(num_levels)).astype(int)
    kinds = ["Dense" if int(r) ** dim <= table else "Hash" for r in res]
    return dict(lod_res=res.tolist(), lod_n_feats=[n_feats] * num_levels, lod_types=kinds, hashmap_size=table)


def single_res_cfg(stretch, voxel_size: float = 0.4, n_feats: int = 8, lotd_type: str = 'Dense', **kwargs) -> dict:
    """one level whose cells have edge ``voxel_size``: resolution = stretch / voxel_size per axis, truncated"""
    res = (np.asarray(stretch) / voxel_size).astype(int)
    return dict(lod_res=[res.tolist()], lod_n_feats=[n_feats], lod_types=[lotd_type], **kwargs)


def _aspect(stretch, dim):
    return np.asarray([stretch] * dim if isinstance(stretch, Number) else stretch)


def auto_ngp_cfg(stretch, target_num_params: int, *, dim: int = 3, n_feats: int = 2, log2_hashmap_size: int = 19,
                 min_res: int = 4, per_level_scale: float = 1.382, max_num_levels: int = 128) -> dict:
    """Dense -> Hash ladder for a cuboid with side ratios ``stretch`` and about ``target_num_params`` parameters.

    The finest Dense level is the one whose grid has about table / 2.5 cells (so that each level grows the parameter
    count by roughly the same factor up to the first Hash level); coarser Dense levels divide its per-axis resolution by
    per_level_scale**k, the Hash levels multiply it by per_level_scale**(k+1); the Hash level count fills the budget
    (every Hash level costs table * n_feats).  The reference's formulas are kept as they are, including the cube root
    for every ``dim`` and its level-count expression exp(log(r_last / min_res) / per_level_scale)."""
    ratio = _aspect(stretch, dim)
    table = 2 ** log2_hashmap_size
    shortest = ratio.min()
    # shortest-side resolution of the finest Dense level
    r_last = int(((table / 2.5) / (ratio / shortest).prod()) ** (1 / 3))
    n_dense = max(int(np.exp(np.log(r_last / min_res) / per_level_scale) + 1), 1)
    n_hash = max(int(target_num_params / (table * n_feats) - 1 + 0.5), 0)
    n_levels = n_dense + n_hash
    if max_num_levels is not None:
        n_levels = min(n_levels, max_num_levels)
        n_hash = n_levels - n_dense
    finest = ratio / (shortest / r_last)                                    # per-axis, not yet truncated
    down = per_level_scale ** np.arange(n_dense)
    up = per_level_scale ** (np.arange(n_hash) + 1)
    dense_res = (finest[:, None] / down)[:, ::-1].T.astype(int)             # coarse -> fine; truncated last
    hash_res = (finest[:, None] * up).T.astype(int)
    res = np.concatenate([dense_res, hash_res], axis=0)
    return dict(lod_res=res.tolist(), lod_n_feats=[n_feats] * n_levels,
                lod_types=["Dense"] * n_dense + ["Hash"] * n_hash, hashmap_size=table)


def auto_ngp4d_cfg(dim: int = 4, n_feats: int = 2, stretch=1.0, target_num_params: int = 2 ** 32, max_layers=128,
                   min_dense_layers: int = 0, log2_hashmap_size: int = 19, min_res_xyz: int = 4, min_res_w: int = 4,
                   per_level_scale: float = 1.382) -> dict:
    """xyz + w ladder: start from min_res_xyz * aspect (xyz) and min_res_w (w), multiply by per_level_scale per level
    (resolutions rounded up), Dense while the grid fits the table (or for the first ``min_dense_layers`` levels), Hash
    afterwards, until the next level would exceed the parameter budget."""
    table = 2 ** log2_hashmap_size
    ratio = _aspect(stretch, dim)
    cur = np.concatenate([min_res_xyz * ratio / ratio.min(), np.array([min_res_w], dtype=np.float32)])
    total, res, kinds = 0, [], []
    for i in range(max_layers):
        r = np.ceil(cur).astype(np.int64)
        cells = r.prod()
        is_hash = cells > table and i >= min_dense_layers
        cost = table * n_feats if is_hash else cells * n_feats
        if total + cost > target_num_params:
            break
        res.append(r.tolist())
        kinds.append("Hash" if is_hash else "Dense")
        total += cost
        cur *= per_level_scale
    return dict(lod_res=res, lod_n_feats=[n_feats] * len(res), lod_types=kinds, hashmap_size=table)


def get_lotd_cfg(type: str, input_ch: int = 3, stretch=None, **kwargs) -> dict:
    if type == 'gen_ngp':
        return gen_ngp_cfg(dim=input_ch, **kwargs)
    if type == 'single_res':
        return single_res_cfg(stretch, **kwargs)
    if type == 'ngp':
        return auto_ngp_cfg(stretch, dim=input_ch, **kwargs)
    if type == 'ngp4d':
        return auto_ngp4d_cfg(dim=input_ch, stretch=stretch, **kwargs)
    if type == 'lotd':
        raise RuntimeError("type='lotd' (the reference's deprecated Dense -> VM generator) is not provided by nr3d_lib_amd")
    raise RuntimeError(f"Invalid type={type}")
