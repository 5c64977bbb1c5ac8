"""LoTD level-layout generators needed by the hot path's benchmark configs.

Counterpart of the reference's nr3d_lib/models/grid_encodings/lotd/lotd_cfg.py: only ``gen_ngp_cfg``
(:48-57, the instant-ngp style geometric ladder used by BASELINE configs 2 and 5) and the ``get_lotd_cfg``
dispatcher entry for it (:21-37) are in scope; the auto-sizing generators are model tooling.
"""
import numpy as np

__all__ = ['get_lotd_cfg', 'gen_ngp_cfg']


def gen_ngp_cfg(min_res: int = 16, dim: int = 3, n_feats: int = 2, log2_hashmap_size: int = 19,
                per_level_scale: float = 1.382, num_levels: int = 16) -> dict:
    """Resolutions min_res * per_level_scale**l (truncated); a level is Dense while its full grid fits
    the hash table (res**dim <= 2**log2_hashmap_size), Hash afterwards."""
    table = 2 ** log2_hashmap_size
    res = (min_res * per_level_scale ** np.arange(num_levels)).astype(int)
    kinds = ["Dense" if int(r) ** dim <= table else "Hash" for r in res]
    return dict(lod_res=res.tolist(), lod_n_feats=[n_feats] * num_levels, lod_types=kinds, hashmap_size=table)


def get_lotd_cfg(type: str, input_ch: int = 3, stretch=None, **kwargs) -> dict:
    if type == 'gen_ngp':
        return gen_ngp_cfg(dim=input_ch, **kwargs)
    raise RuntimeError(f"Invalid type={type} (only 'gen_ngp' is provided by nr3d_lib_amd)")
