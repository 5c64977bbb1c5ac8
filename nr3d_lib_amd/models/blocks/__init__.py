from .mlp import *  # noqa: F401,F403


def get_blocks(in_features: int, out_features: int, use_tcnn_backend=None, use_lipshitz=False, **params):
    """nr3d_lib/models/blocks/__init__.py:3-15.  `use_tcnn_backend` asked the reference for tiny-cuda-nn's fused
    network; the fused kernels here are picked automatically by ``MLP`` whenever they apply, so the flag only drops
    the options tcnn does not know (as the reference does)."""
    if use_lipshitz:
        raise NotImplementedError("nr3d_lib_amd: LipshitzMLP is outside the path")
    if use_tcnn_backend:
        params.pop('weight_norm', None)
    return MLP(in_features, out_features, **params)


get_mlp = get_blocks
