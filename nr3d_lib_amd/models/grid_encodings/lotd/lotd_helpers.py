"""Parameter-layout helpers of the LoTD encoder: where a level's line / plane / volume tables sit inside the flat
parameter vector, views of them, the grid-vertex coordinates and an N-linear sampler of a table.

Counterpart of the reference's nr3d_lib/models/grid_encodings/lotd/lotd_helpers.py (level_param_index_shape :31-207,
get_level_param :209-223, get_level_param_batched :225-242, param_vertices :244-266, param_interpolate :274-346).
Host-side tooling around the hot path (parameter initialisation, inspection, re-gridding); everything is plain PyTorch
and works on any device.  The layout is the one the HIP kernels index (csrc/lotd_device.h) and is pinned against the
reference's function on every level type (tests/golden/ref_python.json -> tests/test_lotd_helpers_cpu.py).
"""
from math import prod
from numbers import Number
from typing import List, Tuple, Union

import torch

from .lotd import LoDType

__all__ = ['level_param_index_shape', 'get_level_param', 'get_level_param_batched', 'param_vertices', 'param_interpolate']

_LINE_OPS, _PLANE_OPS = ('line', 'vec'), ('plane', 'mat')
# which factor tables a level type stores, in storage order
_SECTIONS = {
    LoDType.VectorMatrix: ('line', 'plane'),
    LoDType.NPlaneMul: ('plane',), LoDType.NPlaneSum: ('plane',),
    LoDType.CP: ('line',), LoDType.CPfast: ('line',),
}
# level types the extension builds per input dimension
_TYPES_BY_DIM = {
    1: (LoDType.Dense, LoDType.Hash),
    2: (LoDType.Dense, LoDType.Hash, LoDType.CP, LoDType.CPfast),
    3: tuple(LoDType),
    4: (LoDType.Dense, LoDType.Hash),
}


def level_param_index_shape(lod_meta, l: int, op: str = None, dim: int = None) -> Tuple[tuple, tuple]:
    """(index, shape) of level ``l``'s parameters in the flat vector: ``params[index].view(shape)``.

    op None: the whole level as [entries, M].  'vol' (Dense): [*res, M].  'line' | 'vec' (VM, CP, CPfast) and 'plane' |
    'mat' (VM, NPlaneMul, NPlaneSum): the level's line / plane tables -- of axis ``dim`` ([res[dim], M] resp.
    [cells of the plane orthogonal to dim, M]) or, with dim None, all of them ([D, R, M] / [D, R, R, M] for a cubic
    level, [total entries, M] otherwise).  A VM level stores its lines first, then its planes.  Hash levels ignore op."""
    D, L = lod_meta.n_dims_to_encode, lod_meta.n_levels
    assert 0 <= l < L
    if dim is not None:
        assert 0 <= dim < D
    if D not in _TYPES_BY_DIM:
        raise NotImplementedError
    res = list(lod_meta.level_res_multidim[l])
    M = lod_meta.level_n_feats[l]
    kind = LoDType(int(lod_meta.level_types[l]))
    begin, end = lod_meta.level_offsets[l], lod_meta.level_offsets[l + 1]
    if kind not in _TYPES_BY_DIM[D]:
        if D == 3:
            raise RuntimeError(f"Invalid tp={kind}")
        raise NotImplementedError
    whole = (slice(begin, end),), (lod_meta.level_sizes[l], M)
    if kind == LoDType.Hash or op is None:
        return whole
    if kind == LoDType.Dense:
        if op == 'vol':
            return whole[0], (*res, M)
        raise RuntimeError(f"Invalid op={op}")
    want = 'line' if op in _LINE_OPS else 'plane' if op in _PLANE_OPS else None
    if want is None or want not in _SECTIONS[kind]:
        raise RuntimeError(f"Invalid op={op}")
    cubic = all(r == res[0] for r in res)
    # entries per axis of each kind of table: a line of axis d has res[d] entries, the plane orthogonal to d the rest
    counts = {'line': res, 'plane': [prod(res) // r for r in res]}
    start = begin
    for sec in _SECTIONS[kind]:
        n = counts[sec]
        if sec == want:
            if dim is None:
                stop = start + sum(n) * M
                if cubic:
                    shape = (D, res[0], M) if sec == 'line' else (D, res[0], res[0], M)
                else:
                    shape = (sum(n), M)
                # the last section is addressed as running to the end of the level, like the reference does
                last = sec == _SECTIONS[kind][-1]
                return (slice(start, end if last else stop),), shape
            a = start + sum(n[:dim]) * M
            return (slice(a, a + n[dim] * M),), (n[dim], M)
        start += sum(n) * M
    raise RuntimeError(f"Invalid op={op}")


def get_level_param(params: torch.Tensor, lod_meta, l: int, op: str = None, dim: int = None) -> torch.Tensor:
    """view of level ``l`` (or one of its tables) in the flat ``params``"""
    index, shape = level_param_index_shape(lod_meta, l, op, dim)
    return params[index].view(shape)


def get_level_param_batched(params: torch.Tensor, lod_meta, bid: Union[int, List[int], slice, torch.Tensor] = slice(None),
                            l: int = ..., op: str = None, dim: int = None) -> torch.Tensor:
    """the same for batched params [B, n_params]; ``bid`` selects batch entries and contributes the leading shape"""
    index, shape = level_param_index_shape(lod_meta, l, op, dim)
    lead = params.data[bid, 0].shape
    return params[(bid, *index)].view(*lead, *shape)


def param_vertices(res: Union[int, List[int]], dim: int = 3, is_forest=False, dtype=torch.float, device=None) -> torch.Tensor:
    """[r_0, .., r_{dim-1}, dim]: positions of the table's vertices in the [-1, 1] coordinates of its cube.  A plain
    level keeps one vertex outside the cube on either side (vertex i at (i - 0.5) / (r - 2)), a forest block's vertices
    sit at its cell centres ((i + 0.5) / r)."""
    res = [res] * dim if isinstance(res, Number) else res
    axes = [torch.linspace(-1, 1, r, device=device, dtype=dtype) * ((1. - 1. / r) if is_forest else (1. + 1. / (r - 2.)))
            for r in res]
    return torch.stack(torch.meshgrid(axes, indexing='ij'), dim=-1)


def param_interpolate(param: torch.Tensor, rel_x: torch.Tensor, res: int, is_forest=False) -> torch.Tensor:
    """N-linear sample of cubic tables: param [B, R, (R, (R,)) M], rel_x [B, ..., d] in [-1, 1] (coordinate k runs along
    the table's k-th axis, vertices where ``param_vertices`` puts them) -> [B, ..., M].  Vertices outside the table
    count as zeros."""
    d = rel_x.shape[-1]
    assert 1 <= d <= 3 and param.dim() == d + 2 and param.shape[0] == rel_x.shape[0] and list(param.shape[1:-1]) == [res] * d
    B, M = param.shape[0], param.shape[-1]
    lead = rel_x.shape[1:-1]
    pts = rel_x.reshape(B, -1, d).to(param.dtype)
    # continuous vertex index
    u = ((pts + 1.) * res - 1.) * 0.5 if is_forest else (pts * ((res - 2.) / (res - 1.)) + 1.) * (0.5 * (res - 1.))
    low = torch.floor(u)
    frac = u - low
    low = low.long()
    table = param.reshape(B, -1, M)
    out = torch.zeros(B, pts.shape[1], M, dtype=param.dtype, device=param.device)
    for corner in range(1 << d):
        flat = torch.zeros_like(low[..., 0])
        weight = torch.ones_like(frac[..., 0])
        for k in range(d):
            up = (corner >> k) & 1
            i_k = low[..., k] + up
            weight = weight * (frac[..., k] if up else 1. - frac[..., k])
            weight = weight * ((i_k >= 0) & (i_k < res)).to(weight.dtype)
            flat = flat * res + i_k.clamp(0, res - 1)
        out = out + table.gather(1, flat.unsqueeze(-1).expand(-1, -1, M)) * weight.unsqueeze(-1)
    return out.reshape(B, *lead, M)
