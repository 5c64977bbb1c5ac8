"""numpy/ctypes veneer over oracle/liboracle.so.  TEST INFRASTRUCTURE, NOT PRODUCT."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAX_LEVELS, MAX_DIMS, MAX_PSEUDO = 32, 4, 256

LOD_TYPES = {  # csrc/lotd/include/lotd/lotd_types.h:16-25, string aliases :42-62
    "dense": 0, "vectormatrix": 1, "vm": 1, "veczmatxoy": 2, "cp": 3, "cpfast": 4,
    "nplanemul": 5, "nplanesum": 6, "nplane": 6, "hash": 7,
}


def build(force=False):
    """Compile liboracle.so with gcc (no-op if up to date)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("lotd_oracle.c", "occ_grid_oracle.c", "pack_ops_oracle.c", "lotd_oracle.h")]
    if not force and os.path.exists(_LIB_PATH) and all(
            not os.path.exists(s) or os.path.getmtime(s) <= os.path.getmtime(_LIB_PATH) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


class LotdMeta(C.Structure):
    _fields_ = [
        ("level_res", (C.c_uint32 * MAX_DIMS) * MAX_LEVELS),
        ("level_n_feats", C.c_uint32 * MAX_LEVELS),
        ("level_types", C.c_uint32 * MAX_LEVELS),
        ("level_n_params", C.c_uint32 * MAX_LEVELS),
        ("level_offsets", C.c_uint32 * (MAX_LEVELS + 1)),
        ("level_sizes", C.c_uint32 * MAX_LEVELS),
        ("map_levels", C.c_uint32 * MAX_PSEUDO),
        ("map_cnt", C.c_uint32 * MAX_PSEUDO),
        ("n_levels", C.c_uint32),
        ("n_pseudo_levels", C.c_uint32),
        ("n_feat_per_pseudo_lvl", C.c_uint32),
        ("n_dims_to_encode", C.c_uint32),
        ("n_encoded_dims", C.c_uint32),
        ("n_params", C.c_uint32),
        ("interpolation_type", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        L, D, Q = self.n_levels, self.n_dims_to_encode, self.n_pseudo_levels
        return dict(
            level_res_multidim=[[int(self.level_res[l][d]) for d in range(D)] for l in range(L)],
            level_n_feats=[int(v) for v in self.level_n_feats[:L]],
            level_types=[int(v) for v in self.level_types[:L]],
            level_n_params=[int(v) for v in self.level_n_params[:L]],
            level_offsets=[int(v) for v in self.level_offsets[:L + 1]],
            level_sizes=[int(v) for v in self.level_sizes[:L]],
            map_levels=[int(v) for v in self.map_levels[:Q]],
            map_cnt=[int(v) for v in self.map_cnt[:Q]],
            n_levels=int(L), n_pseudo_levels=int(Q),
            n_feat_per_pseudo_lvl=int(self.n_feat_per_pseudo_lvl),
            n_dims_to_encode=int(D), n_encoded_dims=int(self.n_encoded_dims),
            n_params=int(self.n_params), interpolation_type=int(self.interpolation_type),
        )


def _p(a, ctype=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def set_num_threads(n=0):
    """host threads of the oracle's OpenMP loops (n <= 0: only query); returns the count in effect"""
    f = lib().orc_set_num_threads
    f.restype = C.c_int
    return int(f(C.c_int(int(n))))


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup v2 CPU quota of the container"""
    import math, os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lotd_create_meta(n_input_dim, lod_res, lod_n_feats, lod_types, hashmap_size=None, use_smooth_step=False):
    lod_res = list(lod_res)
    L = len(lod_res)
    if isinstance(lod_n_feats, int):
        lod_n_feats = [lod_n_feats] * L
    if isinstance(lod_types, str):
        lod_types = [lod_types] * L
    res = np.zeros((L, n_input_dim), dtype=np.int32)
    for l, r in enumerate(lod_res):
        res[l, :] = r
    types = np.array([LOD_TYPES[t.lower()] if isinstance(t, str) else int(t) for t in lod_types], dtype=np.int32)
    nf = np.array(lod_n_feats, dtype=np.int32)
    m = LotdMeta()
    err = C.create_string_buffer(256)
    rc = lib().orc_lotd_create_meta(C.c_int32(n_input_dim), C.c_uint32(L), _p(res), _p(nf), _p(types),
                                    C.c_uint32(int(hashmap_size or 0)), C.c_int(int(bool(use_smooth_step))),
                                    C.byref(m), err, C.c_int(256))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return m


def _batch_args(batch_inds, batch_offsets, batch_data_size):
    bi, bo = _i64(batch_inds), _i64(batch_offsets)
    return bi, bo, C.c_uint32(int(batch_data_size or 0))


def _ml(meta, max_level):
    return C.c_int32(int(meta.n_levels if max_level is None else max_level))


def lotd_fwd(meta, x, params, batch_inds=None, batch_offsets=None, batch_data_size=0, max_level=None,
             need_dydx=False):
    x, params = _f32(x), _f32(params)
    N, D, E = x.shape[0], meta.n_dims_to_encode, meta.n_encoded_dims
    y = np.zeros((N, E), np.float32)
    dydx = np.zeros((N, E, D), np.float32) if need_dydx else None
    bi, bo, bds = _batch_args(batch_inds, batch_offsets, batch_data_size)
    lib().orc_lotd_fwd(C.byref(meta), C.c_uint32(N), _p(x), _p(params), _p(bi), _p(bo), bds,
                       _ml(meta, max_level), _p(y), _p(dydx))
    return y, dydx


def lotd_bwd_dparam(meta, dL_dy, x, params, batch_inds=None, batch_offsets=None, batch_data_size=0,
                    max_level=None, accum_double=False):
    dL_dy, x, params = _f32(dL_dy), _f32(x), _f32(params)
    N = x.shape[0]
    g = np.zeros(params.shape[0], np.float32)
    bi, bo, bds = _batch_args(batch_inds, batch_offsets, batch_data_size)
    lib().orc_lotd_bwd_dparam(C.byref(meta), C.c_uint32(N), _p(dL_dy), _p(x), _p(params), _p(bi), _p(bo), bds,
                              _ml(meta, max_level), C.c_int(int(accum_double)), _p(g), C.c_uint64(g.shape[0]))
    return g


def lotd_bwd_dx(meta, dL_dy, dy_dx):
    dL_dy, dy_dx = _f32(dL_dy), _f32(dy_dx)
    N, D = dL_dy.shape[0], meta.n_dims_to_encode
    out = np.zeros((N, D), np.float32)
    lib().orc_lotd_bwd_dx(C.byref(meta), C.c_uint32(N), _p(dL_dy), _p(dy_dx), _p(out))
    return out


def lotd_bwd_bwd_ddLdy(meta, dL_ddLdx, dy_dx):
    dL_ddLdx, dy_dx = _f32(dL_ddLdx), _f32(dy_dx)
    N, E = dL_ddLdx.shape[0], meta.n_encoded_dims
    out = np.zeros((N, E), np.float32)
    lib().orc_lotd_bwd_bwd_ddLdy(C.byref(meta), C.c_uint32(N), _p(dL_ddLdx), _p(dy_dx), _p(out))
    return out


def lotd_bwd_bwd_dparam(meta, dL_ddLdx, dL_dy, x, params, batch_inds=None, batch_offsets=None,
                        batch_data_size=0, max_level=None, accum_double=False):
    dL_ddLdx, dL_dy, x, params = _f32(dL_ddLdx), _f32(dL_dy), _f32(x), _f32(params)
    N = x.shape[0]
    g = np.zeros(params.shape[0], np.float32)
    bi, bo, bds = _batch_args(batch_inds, batch_offsets, batch_data_size)
    lib().orc_lotd_bwd_bwd_dparam(C.byref(meta), C.c_uint32(N), _p(dL_ddLdx), _p(dL_dy), _p(x), _p(params),
                                  _p(bi), _p(bo), bds, _ml(meta, max_level), C.c_int(int(accum_double)),
                                  _p(g), C.c_uint64(g.shape[0]))
    return g


def lotd_bwd_bwd_dx(meta, dL_ddLdx, dL_dy, x, params, batch_inds=None, batch_offsets=None,
                    batch_data_size=0, max_level=None):
    dL_ddLdx, dL_dy, x, params = _f32(dL_ddLdx), _f32(dL_dy), _f32(x), _f32(params)
    N, D = x.shape[0], meta.n_dims_to_encode
    out = np.zeros((N, D), np.float32)
    bi, bo, bds = _batch_args(batch_inds, batch_offsets, batch_data_size)
    lib().orc_lotd_bwd_bwd_dx(C.byref(meta), C.c_uint32(N), _p(dL_ddLdx), _p(dL_dy), _p(x), _p(params),
                              _p(bi), _p(bo), bds, _ml(meta, max_level), _p(out))
    return out


def lotd_grid_index(meta, x, batch_inds=None, batch_offsets=None, batch_data_size=0, max_level=None):
    x = _f32(x)
    N, D, E = x.shape[0], meta.n_dims_to_encode, meta.n_encoded_dims
    out = np.zeros((N, E, 1 << D), np.int64)
    bi, bo, bds = _batch_args(batch_inds, batch_offsets, batch_data_size)
    rc = lib().orc_lotd_grid_index(C.byref(meta), C.c_uint32(N), _p(x), _p(bi), _p(bo), bds,
                                   _ml(meta, max_level), _p(out))
    if rc != 0:
        raise RuntimeError("LoTDEncoding::get_grid_index: Only support Dense/Hash type.")
    return out


# ------------------------------------------------------------------------------------------------
# occ_grid
# ------------------------------------------------------------------------------------------------
def ray_marching(rays_o, rays_d, t_min, t_max, roi, grid_binary, contraction_type, step_size, max_step_size,
                 dt_gamma, max_steps, return_gidx=True, batch_inds=None, batch_data_size=None,
                 return_probes=False):
    """Single-grid (grid_binary.ndim == 3) or batched (ndim == 4, roi [B,6]) marcher.
    Returns [packed_info i32 [n,2], t_starts f32 [S,1], t_ends f32 [S,1], ridx i32 [S], (bidx i32 [S]),
    gidx i32 [S] | None]  (ray_marching.cu:136-244 / batched_marching.cu:153-287)."""
    rays_o, rays_d, t_min, t_max, roi = map(_f32, (rays_o, rays_d, t_min, t_max, roi))
    grid = np.ascontiguousarray(grid_binary).astype(np.uint8)
    batched = grid.ndim == 4
    res = np.array(grid.shape[-3:], dtype=np.int32)
    n = rays_o.shape[0]
    bi = None if batch_inds is None else np.ascontiguousarray(batch_inds, dtype=np.int32)
    num = np.zeros(n, np.int32)
    probes = C.c_uint64(0)
    args = (C.c_uint32(n), _p(rays_o), _p(rays_d), _p(t_min), _p(t_max), _p(roi), _p(res), _p(grid),
            C.c_int(int(contraction_type)), C.c_float(step_size), C.c_float(max_step_size), C.c_float(dt_gamma))
    lib().orc_march_count(*args, C.c_uint32(int(max_steps)), C.c_int(int(batched)), _p(bi),
                          C.c_uint32(int(batch_data_size or 0)), _p(num), C.byref(probes))
    cum = np.cumsum(num, dtype=np.int32)
    packed_info = np.stack([cum - num, num], 1).astype(np.int32)
    S = int(cum[-1]) if n > 0 else 0
    t_starts, t_ends = np.zeros((S, 1), np.float32), np.zeros((S, 1), np.float32)
    ridx = np.zeros(S, np.int32)
    bidx = np.zeros(S, np.int32) if batched else None
    gidx = np.zeros(S, np.int32) if return_gidx else None
    lib().orc_march_emit(*args, C.c_int(int(batched)), _p(bi), C.c_uint32(int(batch_data_size or 0)),
                         _p(packed_info), _p(t_starts), _p(t_ends), _p(ridx), _p(bidx), _p(gidx))
    out = [packed_info, t_starts, t_ends, ridx] + ([bidx] if batched else []) + [gidx]
    if return_probes:
        out.append(int(probes.value))
    return out


# ------------------------------------------------------------------------------------------------
# pack_ops
# ------------------------------------------------------------------------------------------------
_SFX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64", np.dtype(np.int64): "i64",
        np.dtype(np.int32): "i32"}
_CT = {"f32": C.c_float, "f64": C.c_double, "i64": C.c_int64, "i32": C.c_int32}


def _typed(name, arr):
    sfx = _SFX[np.dtype(arr.dtype)]
    return getattr(lib(), f"orc_{name}_{sfx}"), sfx


def _fd(feats):
    return 1 if feats.ndim == 1 else feats.shape[1]


def interleave_linstep(start, num_steps, step_size, return_idx=True):
    start = np.ascontiguousarray(start)
    num_steps = _i64(num_steps)
    fn, sfx = _typed("interleave_linstep", start)
    total = int(num_steps.sum())
    out = np.zeros(total, start.dtype)
    nidx = np.zeros(total, np.int64) if return_idx else None
    if isinstance(step_size, np.ndarray):
        ss = np.ascontiguousarray(step_size, dtype=start.dtype)
        fn(C.c_uint32(len(start)), _p(num_steps), _p(start), _p(ss), _CT[sfx](0), _CT[sfx](0), _p(out), _p(nidx))
    else:
        fn(C.c_uint32(len(start)), _p(num_steps), _p(start), None, _CT[sfx](0),
           _CT[sfx](start.dtype.type(step_size)), _p(out), _p(nidx))
    return out, nidx


def interleave_arange(stop, return_idx=True):
    stop = _i64(stop)
    total = int(stop.sum())
    out = np.zeros(total, np.int64)
    nidx = np.zeros(total, np.int64) if return_idx else None
    lib().orc_interleave_linstep_i64(C.c_uint32(len(stop)), _p(stop), None, None, C.c_int64(0), C.c_int64(1),
                                     _p(out), _p(nidx))
    return out, nidx


def packed_sum(feats, pack_infos):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    fn, _ = _typed("packed_sum", feats)
    P, fd = pack_infos.shape[0], _fd(feats)
    out = np.zeros((P,) + feats.shape[1:], feats.dtype)
    fn(C.c_uint32(P), C.c_uint32(fd), _p(feats), _p(pack_infos), _p(out))
    return out


def _scan(feats, pack_infos, is_prod, exclusive, reverse):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    fn, _ = _typed("packed_scan", feats)
    out = np.zeros_like(feats)
    fn(C.c_uint32(pack_infos.shape[0]), C.c_uint32(_fd(feats)), _p(feats), _p(pack_infos),
       C.c_int(is_prod), C.c_int(int(exclusive)), C.c_int(int(reverse)), _p(out))
    return out


def packed_cumsum(feats, pack_infos, exclusive=False, reverse=False):
    return _scan(feats, pack_infos, 0, exclusive, reverse)


def packed_cumprod(feats, pack_infos, exclusive=False, reverse=False):
    return _scan(feats, pack_infos, 1, exclusive, reverse)


def packed_diff(feats, pack_infos, pack_appends=None, pack_last_fill=None):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    fn, _ = _typed("packed_diff", feats)
    a = None if pack_appends is None else np.ascontiguousarray(pack_appends, dtype=feats.dtype)
    l = None if pack_last_fill is None else np.ascontiguousarray(pack_last_fill, dtype=feats.dtype)
    out = np.zeros_like(feats)
    fn(C.c_uint32(pack_infos.shape[0]), C.c_uint32(_fd(feats)), _p(feats), _p(a), _p(l), _p(pack_infos), _p(out))
    return out


def packed_backward_diff(feats, pack_infos, pack_prepends=None, pack_first_fill=None):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    fn, _ = _typed("packed_backward_diff", feats)
    a = None if pack_prepends is None else np.ascontiguousarray(pack_prepends, dtype=feats.dtype)
    l = None if pack_first_fill is None else np.ascontiguousarray(pack_first_fill, dtype=feats.dtype)
    out = np.zeros_like(feats)
    fn(C.c_uint32(pack_infos.shape[0]), C.c_uint32(_fd(feats)), _p(feats), _p(a), _p(l), _p(pack_infos), _p(out))
    return out


_BIN = {"add": 0, "sub": 1, "mul": 2, "div": 3}
_CMP = {"gt": 5, "geq": 6, "lt": 7, "leq": 8, "eq": 9, "neq": 10}


def packed_binary(op, feats, other, pack_infos):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    other = np.ascontiguousarray(other, dtype=feats.dtype)
    P, fd = pack_infos.shape[0], _fd(feats)
    if op in _BIN:
        fn, _ = _typed("packed_binary", feats)
        out = np.zeros_like(feats)
        fn(C.c_uint32(P), C.c_uint32(fd), _p(feats), _p(other), _p(pack_infos), C.c_int(_BIN[op]), _p(out))
        return out
    fn, _ = _typed("packed_compare", feats)
    out = np.zeros(feats.shape, np.uint8)
    fn(C.c_uint32(P), C.c_uint32(fd), _p(feats), _p(other), _p(pack_infos), C.c_int(_CMP[op]), _p(out))
    return out.astype(bool)


def packed_matmul(feats, other, pack_infos):
    feats, pack_infos = np.ascontiguousarray(feats), _i64(pack_infos)
    other = np.ascontiguousarray(other, dtype=feats.dtype)
    fn, _ = _typed("packed_matmul", feats)
    out = np.zeros((feats.shape[0], other.shape[1]), feats.dtype)
    fn(C.c_uint32(pack_infos.shape[0]), C.c_uint32(feats.shape[1]), C.c_uint32(other.shape[1]), _p(feats),
       _p(other), _p(pack_infos), _p(out))
    return out


def packed_searchsorted(bins, vals, pack_infos):
    bins, pack_infos = np.ascontiguousarray(bins), _i64(pack_infos)
    vals = np.ascontiguousarray(vals, dtype=bins.dtype)
    fn, _ = _typed("packed_searchsorted", bins)
    out = np.full(vals.shape, -1, np.int64)
    fn(C.c_uint32(pack_infos.shape[0]), _p(bins), _p(vals), _p(pack_infos), C.c_uint32(vals.shape[1]), None, _p(out))
    return out


def packed_searchsorted_packed_vals(bins, pack_infos, vals, val_pack_infos):
    bins, pack_infos, val_pack_infos = np.ascontiguousarray(bins), _i64(pack_infos), _i64(val_pack_infos)
    vals = np.ascontiguousarray(vals, dtype=bins.dtype)
    fn, _ = _typed("packed_searchsorted", bins)
    out = np.full(vals.shape, -1, np.int64)
    fn(C.c_uint32(pack_infos.shape[0]), _p(bins), _p(vals), _p(pack_infos), C.c_uint32(0), _p(val_pack_infos), _p(out))
    return out


def try_merge_two_packs_sorted_aligned(vals_a, pack_infos_a, vals_b, pack_infos_b, b_sorted=True):
    vals_a = np.ascontiguousarray(vals_a)
    vals_b = np.ascontiguousarray(vals_b, dtype=vals_a.dtype)
    pia, pib = _i64(pack_infos_a), _i64(pack_infos_b)
    n = pia[:, 1] + pib[:, 1]
    cs = np.cumsum(n)
    pim = np.ascontiguousarray(np.stack([cs - n, n], 1))
    fn, _ = _typed("try_merge_two_packs_sorted_aligned", vals_a)
    pa, pb = np.zeros(vals_a.shape[0], np.int64), np.zeros(vals_b.shape[0], np.int64)
    fn(C.c_uint32(pia.shape[0]), _p(vals_a), _p(pia), _p(vals_b), _p(pib), _p(pim), C.c_int(int(b_sorted)),
       _p(pa), _p(pb))
    return pa, pb, pim


def packed_sort_qsort(vals, pack_infos, return_idx=True):
    """Sorts ``vals`` IN PLACE (reference behaviour, pack_ops_cuda.cu:2675); returns idx or None."""
    assert vals.flags["C_CONTIGUOUS"]
    pack_infos = _i64(pack_infos)
    fn, _ = _typed("packed_sort_qsort", vals)
    idx = np.arange(vals.shape[0], dtype=np.int64) if return_idx else None
    stack = np.zeros(vals.shape[0] + 2, np.int64)
    fn(C.c_uint32(pack_infos.shape[0]), _p(vals), _p(stack), _p(idx), _p(pack_infos))
    return idx


def interleave_sample_step_wrt_depth_clamped(near, far, max_steps, dt_gamma, min_step_size, max_step_size):
    near, far = _f32(near), _f32(far)
    P = near.shape[0]
    n = np.zeros(P, np.int64)
    lib().orc_sample_step_count(C.c_uint32(P), C.c_uint32(int(max_steps)), C.c_float(dt_gamma),
                                C.c_float(min_step_size), C.c_float(max_step_size), _p(near), _p(far), _p(n))
    cs = np.cumsum(n)
    pi = np.ascontiguousarray(np.stack([cs - n, n], 1))
    S = int(cs[-1]) if P else 0
    t, dt, nidx = np.zeros(S, np.float32), np.zeros(S, np.float32), np.zeros(S, np.int64)
    lib().orc_sample_step_emit(C.c_uint32(P), C.c_float(dt_gamma), C.c_float(min_step_size),
                               C.c_float(max_step_size), _p(near), _p(pi), _p(t), _p(dt), _p(nidx))
    return t, dt, nidx, pi


def interleave_sample_step_wrt_depth_in_packed_segments(near, far, entry, exit, seg_pack_infos, max_steps,
                                                        dt_gamma, min_step_size, max_step_size):
    near, far, entry, exit = map(_f32, (near, far, entry, exit))
    spi = _i64(seg_pack_infos)
    P = near.shape[0]
    n = np.zeros(P, np.int64)
    common = (C.c_uint32(P), C.c_uint32(int(max_steps)), C.c_float(dt_gamma), C.c_float(min_step_size),
              C.c_float(max_step_size), _p(near), _p(far), _p(entry), _p(exit), _p(spi))
    lib().orc_sample_step_segments(*common, C.c_int(0), _p(n), None, None, None, None, None)
    cs = np.cumsum(n)
    pi = np.ascontiguousarray(np.stack([cs - n, n], 1))
    S = int(cs[-1]) if P else 0
    t, dt = np.zeros(S, np.float32), np.zeros(S, np.float32)
    nidx, sidx = np.zeros(S, np.int64), np.zeros(S, np.int64)
    lib().orc_sample_step_segments(*common, C.c_int(1), None, _p(pi), _p(t), _p(dt), _p(nidx), _p(sidx))
    return t, dt, sidx, nidx, pi


def packed_invert_cdf(bins, cdfs, u_vals, pack_infos):
    bins, cdfs, u_vals, pack_infos = _f32(bins), _f32(cdfs), _f32(u_vals), _i64(pack_infos)
    samples = np.zeros_like(u_vals)
    bin_idx = np.full(u_vals.shape, -1, np.int64)
    lib().orc_packed_invert_cdf(C.c_uint32(pack_infos.shape[0]), _p(bins), _p(cdfs), _p(pack_infos), _p(u_vals),
                                C.c_uint32(u_vals.shape[1]), _p(samples), _p(bin_idx))
    return samples, bin_idx


def packed_alpha_to_vw_forward(alphas, pack_infos, early_stop_eps, alpha_thre, compression):
    """-> (weights | None, compact_pack_infos int64 [P,2] | None, compact_selector bool [S] | None)
    (pack_ops_cuda.cu:1850-1907)."""
    alphas, pack_infos = _f32(alphas), _i64(pack_infos)
    P, S = pack_infos.shape[0], alphas.shape[0]
    if compression:
        num = np.zeros(P, np.int64)
        sel = np.zeros(S, np.uint8)
        lib().orc_alpha_to_vw_fwd(C.c_uint32(P), _p(alphas), C.c_float(early_stop_eps), C.c_float(alpha_thre),
                                  _p(pack_infos), None, _p(num), _p(sel))
        cs = np.cumsum(num, dtype=np.int32).astype(np.int64)
        return None, np.stack([cs - num, num], 1), sel.astype(bool)
    w = np.zeros(S, np.float32)
    lib().orc_alpha_to_vw_fwd(C.c_uint32(P), _p(alphas), C.c_float(early_stop_eps), C.c_float(alpha_thre),
                              _p(pack_infos), _p(w), None, None)
    return w, None, None


def packed_alpha_to_vw_backward(weights, grad_weights, alphas, pack_infos, early_stop_eps, alpha_thre):
    weights, grad_weights, alphas, pack_infos = _f32(weights), _f32(grad_weights), _f32(alphas), _i64(pack_infos)
    g = np.zeros_like(alphas)
    lib().orc_alpha_to_vw_bwd(C.c_uint32(pack_infos.shape[0]), _p(alphas), _p(weights), _p(grad_weights),
                              C.c_float(early_stop_eps), C.c_float(alpha_thre), _p(pack_infos), _p(g))
    return g


def mark_pack_boundaries(pack_ids):
    ids = np.ascontiguousarray(pack_ids)
    out = np.zeros(ids.shape[0], np.int32)
    if ids.dtype == np.int32:
        lib().orc_mark_pack_boundaries_i32(C.c_int64(ids.shape[0]), _p(ids), _p(out))
    else:
        ids = ids.astype(np.int64)
        lib().orc_mark_pack_boundaries_i64(C.c_int64(ids.shape[0]), _p(ids), _p(out))
    return out


def get_pack_infos_from_n(n_per_pack):
    n = _i64(n_per_pack)
    cs = np.cumsum(n)
    return np.ascontiguousarray(np.stack([cs - n, n], 1))


# ---------------------------------------------------------------------------------------------------------------------
# occupancy-value grid maintenance (nr3d_lib/models/accelerations/occgrid/utils.py:80-125): numpy restatement
# ---------------------------------------------------------------------------------------------------------------------
def occ_gidx_from_pts(pts, res):
    """((pts / 2 + 0.5) * resolution).long().clamp(0, resolution - 1) in fp32 (utils.py:98)"""
    p = np.asarray(pts, np.float32)
    r = np.asarray(res, np.int64)
    g = ((p / np.float32(2.0) + np.float32(0.5)) * r.astype(np.float32)).astype(np.float32)
    return np.clip(np.trunc(g).astype(np.int64), 0, r - 1)


def occ_scatter_max(shape, gidx, occ_val, bidx=None):
    """per-voxel maximum of occ_val (-inf where no sample falls), flat over `shape` = [Rx,Ry,Rz] or [B,Rx,Ry,Rz]"""
    res = tuple(shape[-3:])
    vol = int(np.prod(res))
    gi = np.asarray(gidx, np.int64)
    val = np.asarray(occ_val, np.float32)
    if len(shape) == 4 and bidx is None:
        b = np.repeat(np.arange(shape[0]), gi.shape[1])
    else:
        b = np.zeros(gi.reshape(-1, 3).shape[0], np.int64) if bidx is None else np.asarray(bidx, np.int64).reshape(-1)
    gi, val = gi.reshape(-1, 3), val.reshape(-1)
    idx = b * vol + gi[:, 0] * (res[1] * res[2]) + gi[:, 1] * res[2] + gi[:, 2]
    vmax = np.full(int(np.prod(shape)), -np.inf, np.float32)
    np.maximum.at(vmax, idx, val)
    return vmax


def occ_apply_max(grid, vmax, ema_decay=1.0):
    g = np.array(grid, np.float32, copy=True)
    flat = g.reshape(-1)
    touched = vmax > -np.inf
    flat[touched] = np.maximum(np.float32(ema_decay) * flat[touched], vmax[touched])
    return g


def occ_update_grid(grid, gidx, occ_val, ema_decay=1.0, bidx=None):
    """returns the updated copy: touched voxels get max(ema_decay * old, max of their samples), the rest keep old.
    grid [Rx,Ry,Rz] or [B,Rx,Ry,Rz] (then bidx [n] or gidx [B,n,3] / occ_val [B,n])"""
    return occ_apply_max(grid, occ_scatter_max(np.shape(grid), gidx, occ_val, bidx), ema_decay)


def octree_mark_consecutive_segments(pidx, pack_infos, point_hierarchies):
    """pack_ops_cuda.cu:2807-2841 with the pack offset applied to the node list (the reference's kernel omits it)"""
    pidx, pi = np.asarray(pidx, np.int64), np.asarray(pack_infos, np.int64)
    pts = np.asarray(point_hierarchies, np.int64)
    ms, me = np.zeros(pidx.shape[0], bool), np.zeros(pidx.shape[0], bool)
    for b, n in pi:
        if n == 0:
            continue
        ms[b] = True
        for j in range(1, n):
            if np.abs(pts[pidx[b + j]] - pts[pidx[b + j - 1]]).sum() > 1:
                me[b + j - 1] = True
                ms[b + j] = True
        me[b + n - 1] = True
    return ms, me


# ------------------------------------------------------------------------------------------------
# forest of blocks (csrc/forest/forest.h, lotd_forest.h, forest_marching.cu)
# ------------------------------------------------------------------------------------------------
class _ForestC(C.Structure):
    _fields_ = [("octree", C.c_void_p), ("exsum", C.c_void_p), ("block_ks", C.c_void_p), ("n_trees", C.c_uint32),
                ("level", C.c_uint32), ("level_poffset", C.c_uint32), ("continuity_enabled", C.c_int32)]


class Forest:
    """host copy of the reference's ForestMeta (forest_cpp_api.h:18-37)"""

    def __init__(self, octree, exsum, block_ks, level, level_poffset, world_origin=(0, 0, 0), world_block_size=(1, 1, 1),
                 continuity_enabled=True):
        self.octree = np.ascontiguousarray(octree, np.uint8)
        self.exsum = np.ascontiguousarray(exsum, np.int32)
        self.block_ks = np.ascontiguousarray(block_ks, np.int16)
        self.level, self.level_poffset = int(level), int(level_poffset)
        self.n_trees = self.block_ks.shape[0]
        self.world_origin = np.asarray(world_origin, np.float64)
        self.world_block_size = np.asarray(world_block_size, np.float64)
        self.continuity_enabled = bool(continuity_enabled)

    def c(self):
        return _ForestC(self.octree.ctypes.data, self.exsum.ctypes.data, self.block_ks.ctypes.data, self.n_trees,
                        self.level, self.level_poffset, int(self.continuity_enabled))


def forest_from_blocks(block_coords, level, **kw):
    """Octree over the given integer block coordinates on `level`, in the layout kaolin's SPC uses and `identify`
    walks: nodes breadth first, children of a node in child-index order (x<<2 | y<<1 | z), one occupancy byte per
    non-leaf node, exsum = exclusive prefix sum of the bytes' popcounts (one extra trailing entry), node 0 = root.
    Built level by level with python sets -- deliberately naive."""
    coords = sorted({tuple(int(v) for v in c) for c in np.asarray(block_coords).reshape(-1, 3)})
    assert all(0 <= v < (1 << level) for c in coords for v in c)
    levels = [None] * (level + 1)
    levels[level] = set(coords)
    for l in range(level - 1, -1, -1):
        levels[l] = {(x >> 1, y >> 1, z >> 1) for x, y, z in levels[l + 1]}
    order = [[(0, 0, 0)]] if level >= 0 else []
    octree = []
    for l in range(level):
        nxt = []
        for x, y, z in order[l]:
            bits = 0
            for child in range(8):
                k = (2 * x + ((child >> 2) & 1), 2 * y + ((child >> 1) & 1), 2 * z + (child & 1))
                if k in levels[l + 1]:
                    bits |= 1 << child
                    nxt.append(k)
            octree.append(bits)
        order.append(nxt)
    octree = np.array(octree, np.uint8)
    pop = np.array([bin(b).count("1") for b in octree], np.int64)
    exsum = np.concatenate([[0], np.cumsum(pop)]).astype(np.int32)
    poffset = sum(len(o) for o in order[:level])
    return Forest(octree, exsum, np.array(order[level], np.int16).reshape(-1, 3), level, poffset, **kw)


def forest_identify(forest, ks):
    ks = np.ascontiguousarray(ks, np.int16).reshape(-1, 3)
    fc = forest.c()
    lib().orc_forest_identify.restype = C.c_int32
    return np.array([lib().orc_forest_identify(C.byref(fc), _p(ks[i:i + 1])) for i in range(ks.shape[0])], np.int32)


def lotd_forest_fwd(meta, forest, x, params, block_inds=None, block_offsets=None, batch_data_size=0, max_level=None,
                    need_dydx=False):
    x, params = _f32(x), _f32(params)
    N, E = x.shape[0], meta.n_encoded_dims
    y = np.zeros((N, E), np.float32)
    dydx = np.zeros((N, E, 3), np.float32) if need_dydx else None
    bi, bo, bds = _batch_args(block_inds, block_offsets, batch_data_size)
    fc = forest.c()
    rc = lib().orc_lotd_forest_fwd(C.byref(meta), C.byref(fc), C.c_uint32(N), _p(x), _p(params), _p(bi), _p(bo), bds,
                                   _ml(meta, max_level), _p(y), _p(dydx))
    if rc:
        raise RuntimeError("lotd forest: 3D Dense / VM / NPlaneMul / CP / Hash levels only")
    return y, dydx


def lotd_forest_bwd_dparam(meta, forest, dL_dy, x, params, block_inds=None, block_offsets=None, batch_data_size=0,
                           max_level=None, dL_ddLdx=None, accum_double=False):
    """dL_ddLdx None: dL/dparam; else d(dL/dx)/dparam"""
    dL_dy, x, params, g2 = _f32(dL_dy), _f32(x), _f32(params), _f32(dL_ddLdx)
    grad = np.zeros(params.shape[0], np.float32)
    bi, bo, bds = _batch_args(block_inds, block_offsets, batch_data_size)
    fc = forest.c()
    rc = lib().orc_lotd_forest_bwd_dparam(C.byref(meta), C.byref(fc), C.c_uint32(x.shape[0]), _p(g2), _p(dL_dy), _p(x),
                                          _p(params), _p(bi), _p(bo), bds, _ml(meta, max_level),
                                          C.c_int(int(accum_double)), _p(grad), C.c_uint64(grad.shape[0]))
    if rc:
        raise RuntimeError("lotd forest: 3D Dense / VM / NPlaneMul / CP / Hash levels only")
    return grad


def lotd_forest_bwd_bwd_dx(meta, forest, dL_ddLdx, dL_dy, x, params, block_inds=None, block_offsets=None,
                           batch_data_size=0, max_level=None):
    dL_ddLdx, dL_dy, x, params = _f32(dL_ddLdx), _f32(dL_dy), _f32(x), _f32(params)
    out = np.zeros((x.shape[0], 3), np.float32)
    bi, bo, bds = _batch_args(block_inds, block_offsets, batch_data_size)
    fc = forest.c()
    rc = lib().orc_lotd_forest_bwd_bwd_dx(C.byref(meta), C.byref(fc), C.c_uint32(x.shape[0]), _p(dL_ddLdx), _p(dL_dy),
                                          _p(x), _p(params), _p(bi), _p(bo), bds, _ml(meta, max_level), _p(out))
    if rc:
        raise RuntimeError("lotd forest: 3D Dense / VM / NPlaneMul / CP / Hash levels only")
    return out


def forest_ray_marching(forest, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits, seg_pack_infos,
                        grid_binary, step_size, max_step_size, dt_gamma, max_steps, return_gidx=False):
    """-> (packed_info int32 [n_rays,2], t_starts [S,1], t_ends [S,1], ridx, blidx, gidx | None)"""
    o, d, tn, tf = _f32(rays_o), _f32(rays_d), _f32(t_min), _f32(t_max)
    sb = np.ascontiguousarray(seg_block_inds, np.int32)
    se, sx = _f32(seg_entries), _f32(seg_exits)
    sp = np.ascontiguousarray(seg_pack_infos, np.int32)
    grid = np.ascontiguousarray(grid_binary).astype(np.uint8)
    res = np.array(grid.shape[1:], np.int32)
    wo, wb = forest.world_origin.astype(np.float32), forest.world_block_size.astype(np.float32)
    n = o.shape[0]
    num = np.zeros(n, np.int32)

    def run(pi, ts, te, ridx, bl, gi):
        lib().orc_forest_march(C.c_uint32(n), _p(o), _p(d), _p(tn), _p(tf), _p(sb), _p(se), _p(sx), _p(sp), _p(res),
                               _p(grid), _p(forest.block_ks), _p(wo), _p(wb), C.c_float(step_size),
                               C.c_float(max_step_size), C.c_float(dt_gamma), C.c_uint32(max_steps), _p(pi), _p(num),
                               _p(ts), _p(te), _p(ridx), _p(bl), _p(gi))
    run(None, None, None, None, None, None)
    cs = np.cumsum(num, dtype=np.int32)
    pi = np.ascontiguousarray(np.stack([cs - num, num], 1).astype(np.int32))
    S = int(cs[-1]) if n else 0
    ts, te = np.zeros((S, 1), np.float32), np.zeros((S, 1), np.float32)
    ridx, bl = np.zeros(S, np.int32), np.zeros(S, np.int32)
    gi = np.zeros(S, np.int32) if return_gidx else None
    run(pi, ts, te, ridx, bl, gi)
    return pi, ts, te, ridx, bl, gi
