"""nr3d_lib_amd.bindings._lotd -- drop-in for the reference pybind module ``nr3d_lib.bindings._lotd``
(csrc/lotd/src/lotd.cpp:23-110), backed by the HIP kernels in libnr3d_hip.so through the C ABI of
include/nr3d_hip.h.

Same Python-visible names, argument order/defaults and return structure:
  LoDType, InterpolationType, LoDMeta, lod_fwd, lod_bwd, lod_bwd_bwd_input, lod_get_grid_index.

Differences that are deliberate and documented in DESIGN.md:
  * y / dy_dx are stored feature-major ([E, N] / [E, N, D]) and returned as transposed views for EVERY
    meta (the reference does this only on its hash-only path; its generic path returns row-major
    [N, E] / [N, E*D]).  Set ``meta.c_permute_dydx = False`` to get the reference's row-major
    [N, E*D] Jacobian.
  * kernels compute in fp32; fp16 params / inputs are converted on the way in and results cast back
    (the reference accumulates in half on its (half, half) path).
  * the forest overloads (``metas`` = ``(lod_meta, forest_meta)`` tuple, lotd.cpp:44-60) live in
    ``bindings/_forest.py``; the functions here dispatch to them.
"""
import ctypes as C
import enum

import torch

from .. import _hip as H

MAX_LEVELS, MAX_DIMS, MAX_PSEUDO = 32, 4, 256


class LoDType(enum.IntEnum):
    """csrc/lotd/include/lotd/lotd_types.h:16-25 (VecZMatXoY is not exported by the reference's
    pybind enum, lotd.cpp:60-68; kept here because the string type is accepted)."""
    Dense = 0
    VectorMatrix = 1
    VecZMatXoY = 2
    CP = 3
    CPfast = 4
    NPlaneMul = 5
    NPlaneSum = 6
    Hash = 7


class InterpolationType(enum.IntEnum):
    Linear = 0
    Smoothstep = 1


# export_values() of the pybind enums
for _e in (LoDType, InterpolationType):
    for _k, _v in _e.__members__.items():
        globals()[_k] = _v

_TYPE_NAMES = {  # lotd_types.h:42-62 (case-insensitive)
    "dense": LoDType.Dense, "hash": LoDType.Hash, "nplane": LoDType.NPlaneSum, "nplanesum": LoDType.NPlaneSum,
    "nplanemul": LoDType.NPlaneMul, "vectormatrix": LoDType.VectorMatrix, "vm": LoDType.VectorMatrix,
    "veczmatxoy": LoDType.VecZMatXoY, "cpfast": LoDType.CPfast, "cp": LoDType.CP,
}


def string_to_lod_type(s):
    try:
        return _TYPE_NAMES[str(s).lower()]
    except KeyError:
        raise RuntimeError(f"LoTDEncoding: Invalid lod type: {s}")


class _CLevel(C.Structure):
    _fields_ = [("res", C.c_uint32 * MAX_DIMS), ("n_feats", C.c_uint32), ("type", C.c_uint32),
                ("size", C.c_uint32), ("offset", C.c_uint32)]


class _CMeta(C.Structure):
    """nr3d_lotd_meta_t of include/nr3d_hip.h"""
    _fields_ = [
        ("levels", _CLevel * MAX_LEVELS),
        ("map_levels", C.c_uint16 * MAX_PSEUDO),
        ("map_cnt", C.c_uint16 * MAX_PSEUDO),
        ("n_levels", C.c_uint32), ("n_pseudo_levels", C.c_uint32), ("n_feat_per_pseudo_lvl", C.c_uint32),
        ("n_dims_to_encode", C.c_uint32), ("n_encoded_dims", C.c_uint32), ("n_params", C.c_uint32),
        ("interpolation_type", C.c_uint32), ("c_hash_only", C.c_uint32),
        ("map_col", C.c_uint16 * MAX_PSEUDO),          # ABI 2: first output column of every pseudo level
    ]


class LoDMeta:
    """LoDMeta(n_input_dims, lod_res | lod_res_multidim, lod_n_feats, lod_types, hashmap_size=None,
    use_smooth_step=None) -- lotd.cpp:75-110, lotd_torch_api.h:79-142, create_meta lotd_torch_api.cu:29-230."""

    def __init__(self, n_input_dims, lod_res, lod_n_feats, lod_types, hashmap_size=None, use_smooth_step=None):
        lod_res, lod_n_feats, lod_types = list(lod_res), list(lod_n_feats), list(lod_types)
        if not (len(lod_res) == len(lod_n_feats) == len(lod_types)):
            raise RuntimeError("LoTDEncoding: Expect los_res, lod_n_feats, lod_types to have the same length")
        L, D = len(lod_res), int(n_input_dims)
        if D not in (2, 3, 4):
            raise RuntimeError("LoTDEncoding: `n_input_dim` must be 2/3/4.")
        res = (C.c_int32 * (L * D))()
        for l, r in enumerate(lod_res):
            rr = [int(r)] * D if not isinstance(r, (list, tuple)) else [int(v) for v in r]
            if len(rr) != D:
                raise RuntimeError("LoTDEncoding: each multi-dim resolution must have n_input_dims entries")
            for d in range(D):
                res[l * D + d] = rr[d]
        nf = (C.c_int32 * L)(*[int(v) for v in lod_n_feats])
        self.level_types_str = [str(t) for t in lod_types]
        tp = (C.c_int32 * L)(*[int(string_to_lod_type(t)) for t in lod_types])
        self._c = _CMeta()
        H.check(H.lib().nr3d_lotd_meta_create(C.c_int32(D), C.c_uint32(L), res, nf, tp,
                                              C.c_uint32(int(hashmap_size or 0)),
                                              C.c_int(int(bool(use_smooth_step))), C.byref(self._c)))
        c = self._c
        # the per-level lists (level_res_multidim, level_res, level_n_feats, level_types, level_sizes, level_n_params, level_offsets,
        # map_levels, map_cnt) are read out of the C struct on first use (__getattr__ below): building them here was half of the
        # constructor's host time, and most callers only ever ask for n_params / n_encoded_dims
        self.n_levels = int(c.n_levels)
        self.n_pseudo_levels = int(c.n_pseudo_levels)
        self.n_feat_per_pseudo_lvl = int(c.n_feat_per_pseudo_lvl)
        self.n_dims_to_encode = int(c.n_dims_to_encode)
        self.n_encoded_dims = int(c.n_encoded_dims)
        self.n_params = int(c.n_params)
        self.interpolation_type = InterpolationType(int(c.interpolation_type))
        # read-write configuration flags (lotd_torch_api.h:98-103)
        self.c_hash_only = bool(c.c_hash_only)
        self.c_profile = False
        self.c_bmm_backend = True
        self.c_prefetch = True
        self.c_permute_dydx = True
        self._all_dense_hash = bool(c.c_hash_only)
        self._dev_cache = {}
        # Levels regrouped by their own feature width (nr3d_lotd_meta_regroup): the reference -- and the meta above -- walk
        # every level in pseudo levels of the GLOBAL gcd of the widths, so a meta that mixes 2- and 16-feature levels
        # (configs[3]: 4, 4, 8, 4, 2, 16, 8, 4) repeats the index work of the 16-feature level eight times and emits eight
        # 12-byte records per table entry where two 36-byte ones would do.  The FORWARD of a meta with product-type levels
        # and mixed widths runs as up to three calls (width 8, 4, 2) that write disjoint output columns (configs[3], 2^22
        # points: 4.42 -> 3.62 ms).  The parameter-gradient and second-order passes keep the single call: measured with wide
        # records / lanes they LOSE (dL/dparam 7.95 -> 12.8 ms, d(dL/dx)/dx 3.58 -> 7.17: three times fewer points per
        # stage-A workgroup for 36-byte records, four times more buckets for the accumulators, registers).  Dense / Hash-only
        # metas keep the single call everywhere (their 2-feature pair kernels are faster than wider lanes).
        # REGROUP = False: always one call.
        self._groups = None
        if not self._all_dense_hash and any(int(f) % (2 * self.n_feat_per_pseudo_lvl) == 0 for f in lod_n_feats):
            groups = []
            for width in (8, 4, 2):
                g = _CMeta()
                H.check(H.lib().nr3d_lotd_meta_regroup(C.byref(self._c), C.c_uint32(width), C.c_uint32(8), C.byref(g)))
                if g.n_pseudo_levels:
                    groups.append(g)
            if len(groups) > 1 or (groups and groups[0].n_feat_per_pseudo_lvl != self.n_feat_per_pseudo_lvl):
                self._groups = groups
        self._group_dev_cache = {}

    _LAZY = ("level_res_multidim", "level_res", "level_n_feats", "level_types", "level_sizes", "level_n_params", "level_offsets",
             "map_levels", "map_cnt")

    def __getattr__(self, name):
        # only reached for attributes that are not set yet: the lazily built per-level lists
        if name in LoDMeta._LAZY and "_c" in self.__dict__:
            c, L, D = self._c, int(self._c.n_levels), int(self._c.n_dims_to_encode)
            d = self.__dict__
            d["level_res_multidim"] = [[int(c.levels[l].res[k]) for k in range(D)] for l in range(L)]
            d["level_res"] = [r[0] if all(v == r[0] for v in r) else 0 for r in d["level_res_multidim"]]
            d["level_n_feats"] = [int(c.levels[l].n_feats) for l in range(L)]
            d["level_types"] = [int(c.levels[l].type) for l in range(L)]
            d["level_sizes"] = [int(c.levels[l].size) for l in range(L)]
            d["level_n_params"] = [sz * f for sz, f in zip(d["level_sizes"], d["level_n_feats"])]
            d["level_offsets"] = [int(c.levels[l].offset) for l in range(L)] + [int(c.n_params)]
            d["map_levels"] = [int(c.map_levels[q]) for q in range(c.n_pseudo_levels)]
            d["map_cnt"] = [int(c.map_cnt[q]) for q in range(c.n_pseudo_levels)]
            return d[name]
        raise AttributeError(name)

    # ---- C-ABI views -------------------------------------------------------------------------
    def _cmeta(self):
        """host struct with the current value of the writable c_hash_only flag"""
        want = 1 if (self.c_hash_only and self._all_dense_hash) else 0
        if self._c.c_hash_only != want:
            self._c.c_hash_only = want
            self._dev_cache.clear()
        return self._c

    def _dev(self, device):
        """device-resident byte copy of the struct (uploaded once per device)"""
        c = self._cmeta()
        key = (device.type, device.index)
        t = self._dev_cache.get(key)
        if t is None:
            host = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8)
            t = host.to(device)
            self._dev_cache[key] = t
        return t

    def _calls(self, device):
        """[(byref(host meta), device-copy pointer)] of the calls that together cover every level: the regrouped metas
        (width 8 / 4 / 2) of a mixed-width meta with product-type levels, else the meta itself"""
        if self._groups is None or not REGROUP:
            return [(C.byref(self._cmeta()), H.ptr(self._dev(device)))]
        key = (device.type, device.index)
        devs = self._group_dev_cache.get(key)
        if devs is None:
            devs = [torch.frombuffer(bytearray(bytes(g)), dtype=torch.uint8).to(device) for g in self._groups]
            self._group_dev_cache[key] = devs
        return [(C.byref(g), H.ptr(d)) for g, d in zip(self._groups, devs)]

    def __repr__(self):
        return (f"LoDMeta(D={self.n_dims_to_encode}, levels={self.n_levels}, n_params={self.n_params}, "
                f"n_encoded_dims={self.n_encoded_dims}, hash_only={self.c_hash_only})")


# ------------------------------------------------------------------------------------------------
# argument checks shared by the entry points (lotd_torch_api.cu:244-290 and twins)
# ------------------------------------------------------------------------------------------------
def _is_divisible(a, b):
    return (a - (a // b) * b) == 0


def _check_common(fn, meta, input, params, batch_inds, batch_offsets, batch_data_size):
    if isinstance(meta, tuple):
        raise RuntimeError(f"{fn}: expected a LoDMeta, got a tuple (forest calls go through bindings._forest)")
    if input.dim() != 2:
        raise RuntimeError(f"{fn}: Expected 2-dimensional tensor for argument x, got {input.dim()}")
    H.require_gpu(input, params, batch_inds, batch_offsets)
    if not input.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous tensor for argument x")
    if input.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"{fn}: Expected x to have one of scalar types Half, Float; got {input.dtype}")
    if input.shape[1] != meta.n_dims_to_encode:
        raise RuntimeError(f"{fn}: Expected x to have size {meta.n_dims_to_encode} at dimension 1, "
                           f"but got size {input.shape[1]}")
    N = input.shape[0]
    if params is not None:
        if params.dim() != 1:
            raise RuntimeError(f"{fn}: Expected 1-dimensional tensor for argument grid, got {params.dim()}")
        if not params.is_contiguous():
            raise RuntimeError(f"{fn}: Expected contiguous tensor for argument grid")
        if params.dtype not in (torch.float16, torch.float32):
            raise RuntimeError(f"{fn}: Expected grid to have one of scalar types Half, Float; got {params.dtype}")
        if params.device != input.device:
            raise RuntimeError(f"{fn}: Expected x and grid on the same GPU")
        if not _is_divisible(params.shape[0], meta.n_params):
            raise RuntimeError(f"LoTDEncoding::{fn}: Expect size of `params`={params.shape[0]} to be an integral "
                               f"multiple of `n_param`={meta.n_params}")
        if input.dtype == torch.float16 and params.dtype == torch.float32:
            raise RuntimeError("LoTDEncoding: Input type combination not supported. Supported types are: "
                               "<input,param> -> (half, half), (float, half), (float, float)")
    if batch_inds is not None:
        if batch_inds.dim() != 1 or batch_inds.shape[0] != N or batch_inds.dtype != torch.int64 \
                or not batch_inds.is_contiguous() or batch_inds.device != input.device:
            raise RuntimeError(f"{fn}: batch_inds must be a contiguous int64 [n_points] tensor on the input's GPU")
    if batch_offsets is not None:
        if batch_offsets.dim() != 1 or batch_offsets.dtype != torch.int64 or not batch_offsets.is_contiguous() \
                or batch_offsets.device != input.device:
            raise RuntimeError(f"{fn}: batch_offsets must be a contiguous int64 1-D tensor on the input's GPU")
    bds = int(batch_data_size) if batch_data_size is not None else 0
    if not (bds == 0 or _is_divisible(N, bds)):
        raise RuntimeError(f"LoTDEncoding::{fn}: Expect nonzero `batch_data_size`={bds} to be a divisor of "
                           f"`batch_size`={N}")
    return N, bds


# scratch for the atomic-free parameter-gradient path, grown on demand, one buffer per (device, stream)
_workspaces = {}
USE_BINNED_DPARAM = True      # False forces the hardware-atomic scatter (debug / A-B measurements)


def _n_batches(meta, params, batch_offsets, batched):
    """number of table sets behind `params` (1 when the call is not batched)"""
    if not batched:
        return 1
    return int(batch_offsets.shape[0]) if batch_offsets is not None else int(params.shape[0]) // meta.n_params


def _dparam_workspace(meta, n_points, device, n_batches=1):
    """(tensor | None, nbytes): device scratch for nr3d_lotd_bwd_dparam's binned path; (None, 0) when that path
    does not apply to this meta."""
    if not USE_BINNED_DPARAM:
        return None, 0
    H.lib().nr3d_lotd_dparam_workspace_bytes.restype = C.c_uint64
    need = int(H.lib().nr3d_lotd_dparam_workspace_bytes(C.byref(meta._cmeta()), H.u32(n_points), H.u32(n_batches)))
    if need == 0:
        return None, 0
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = None
        _workspaces.pop(key, None)
        ws = H.empty(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws, need


def _f32c(t):
    """fp32 working copy (no-op for fp32 tensors)"""
    return t if t.dtype == torch.float32 else t.float()


def _p32(params):
    """fp32 working copy of a half table for the A/B route (NATIVE_HALF = False; the kernels read half tables themselves
    otherwise).  Converted on EVERY call: rounds 2-3 cached the copy per (data_ptr, _version), but a ``.data`` view -- what
    ``LoTDEncoding.inference_param`` passes -- carries a fresh version counter that stays 0, so an optimizer step between two
    inference calls went unseen and the stale table was served (round-3 advisor finding).  The route exists for cross-checks
    only; a conversion per call is its honest cost."""
    p = params.detach()
    return p if p.dtype == torch.float32 else p.float()


def _ptab(params):
    """(table, dtype code) as the kernels get it: half tables as they are (every kernel that reads table entries has a
    __half instantiation: values are read as half and used as float, the reference's (float, half, float) dispatch), unless
    NATIVE_HALF is off -- then the fp32 copy of _p32"""
    if NATIVE_HALF and params.dtype == torch.float16:
        return params.detach(), H.F16
    return _p32(params), H.F32


HVP_WORKSPACE_MAX_BYTES = 8 << 30      # largest per-call scratch of the level-parallel d(dL/dx)/dx (beyond it: lane-serial kernel)
NATIVE_HALF = True       # False: half params always go through fp32 copies (A/B measurements)
REGROUP = True           # False: mixed-width metas run as ONE call in pseudo levels of the global gcd (A/B, cross-check)


def _native_half(meta, params, batched):
    """the kernels read __half params / dL_dy and write __half y / dL_dparam themselves ((float, half, float) type
    combination, lotd_encoding.h:1501-1504): no whole-table conversion per call"""
    return (NATIVE_HALF and params.dtype == torch.float16 and not batched and params.data_ptr() % 4 == 0
            and bool(H.lib().nr3d_lotd_half_params_ok(C.byref(meta._cmeta()), C.c_int(0))))


def _strides2(t):
    return t.stride(0), t.stride(1)


def _jac_view(dy_dx, N, E, D):
    """Interpret a Jacobian handed back by the caller: [N,E,D] (any layout with unit inner stride) or
    contiguous [N,E*D].  Returns (fp32 tensor, stride_n, stride_e)."""
    j = _f32c(dy_dx)
    if j.dim() == 2:
        j = j.view(N, E, D) if j.is_contiguous() else j.reshape(N, E, D)
    if tuple(j.shape) != (N, E, D):
        raise RuntimeError(f"LoTDEncoding: dy_dx must have {N * E * D} elements viewable as [{N}, {E}, {D}]")
    if D > 1 and j.stride(2) != 1:
        j = j.contiguous()
    return j, j.stride(0), j.stride(1)


class _Prof:
    """c_profile: event-timed launches, printed like the reference (lotd_hash_only.h:748-760)."""

    def __init__(self, meta, name, n):
        self.on, self.name, self.n = bool(meta.c_profile), name, n

    def __enter__(self):
        if self.on:
            self.t0, self.t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.t1.record()
            self.t1.synchronize()
            ms = self.t0.elapsed_time(self.t1)
            print(f"{self.name}, {self.n}, {ms}, {ms / max(self.n, 1) * 1e6}")


# ------------------------------------------------------------------------------------------------
# lod_fwd  (lotd.cpp:29-31, lotd_torch_api.cu:232-395)
# ------------------------------------------------------------------------------------------------
def lod_fwd(lod_meta, input, params, batch_inds=None, batch_offsets=None, batch_data_size=None, max_level=None,
            need_input_grad=None):
    if isinstance(lod_meta, tuple):
        from . import _forest
        return _forest.lod_fwd(lod_meta, input, params, batch_inds, batch_offsets, batch_data_size, max_level, need_input_grad)
    m = lod_meta
    N, bds = _check_common("fwd", m, input, params, batch_inds, batch_offsets, batch_data_size)
    max_level = m.n_levels if max_level is None else int(max_level)
    need_input_grad = bool(input.requires_grad) if need_input_grad is None else bool(need_input_grad)
    E, D = m.n_encoded_dims, m.n_dims_to_encode
    if max_level <= -1:  # lotd_torch_api.cu:294-297
        return (torch.zeros((N, E), dtype=params.dtype, device=params.device),
                torch.zeros((N, E * D), dtype=input.dtype, device=input.device))
    batched = batch_inds is not None or batch_offsets is not None or bds != 0
    # half tables are read as they are and y leaves as half -- by the two-lane kernels where they apply, by the general kernel
    # for every other meta and for batched tables (k_fwd<..., __half>): no fp32 copy of the table
    native = NATIVE_HALF and params.dtype == torch.float16
    x32 = _f32c(input.detach())
    p32 = params.detach() if native else _p32(params)
    pcode = H.F16 if native else H.F32
    dev = input.device
    with H.on_device(dev):
        y_store = H.empty((E, N), dtype=torch.float16 if native else torch.float32, device=dev)
        y = y_store.t()
        dy_dx = None
        if need_input_grad:
            if m.c_permute_dydx:
                dy_dx = H.empty((E, N, D), dtype=torch.float32, device=dev).permute(1, 0, 2)
                dsn, dse = dy_dx.stride(0), dy_dx.stride(1)
            else:
                dy_dx = H.empty((N, E * D), dtype=torch.float32, device=dev)
                dsn, dse = E * D, D
        else:
            dsn = dse = 0
        with _Prof(m, f"LoTD{D}-fwd" + ("-grad" if need_input_grad else ""), N):
            for cm, md in m._calls(dev):          # one call, or one per feature width (disjoint output columns)
                H.check(H.lib().nr3d_lotd_fwd(
                    cm, md, H.u32(N), C.c_int(H.F32), C.c_int(pcode),
                    H.ptr(x32), H.ptr(p32), H.ptr(batch_inds), H.ptr(batch_offsets), H.u32(bds), H.i32(max_level),
                    H.ptr(y_store), H.i64(y.stride(0)), H.i64(y.stride(1)),
                    H.ptr(dy_dx), H.i64(dsn), H.i64(dse), H.stream_of(input)))
    if y.dtype != params.dtype:
        y = y.to(params.dtype)
    if dy_dx is not None and input.dtype != torch.float32:
        dy_dx = dy_dx.to(input.dtype)
    return y, dy_dx


# ------------------------------------------------------------------------------------------------
# encode + decoder forward in one kernel (csrc/lotd_mlp.hip; no reference counterpart)
# ------------------------------------------------------------------------------------------------
def lod_mlp_fwd_ok(lod_meta, mlp_desc):
    """can ``lod_mlp_fwd`` serve this (meta, decoder) pair?  (3-D Dense / Hash meta, 2-feature pseudo levels, <= 32 encoded dims = the
    decoder's input width; fp32 decoder with hidden width <= 64 and <= 32 outputs)"""
    if isinstance(lod_meta, tuple) or mlp_desc is None or not getattr(mlp_desc, "fusable", False):
        return False
    if lod_meta._groups is not None and REGROUP:
        return False
    return bool(H.lib().nr3d_lotd_mlp_forward_ok(C.byref(lod_meta._cmeta()), C.byref(mlp_desc._c)))


def lod_mlp_fwd(lod_meta, input, params, mlp_desc, mlp_packed, out_cols=None, max_level=None):
    """out [N, out_cols] float32 = the first ``out_cols`` columns of decoder(lod_fwd(input, params)[0]) (all of them when None), computed
    by ONE kernel that never writes the [N, n_encoded_dims] features: the no-grad density query of the ray driver
    (nerf_ray_query.py:105-127).  ``mlp_desc`` / ``mlp_packed``: bindings._mlp.MLPDesc and its packed buffer (``_mlp.pack``).
    No autograd: callers use it under no_grad."""
    m = lod_meta
    if not lod_mlp_fwd_ok(m, mlp_desc):
        raise RuntimeError("lod_mlp_fwd: this meta / decoder pair is outside the fused kernel's range (lod_mlp_fwd_ok)")
    N, _ = _check_common("mlp_fwd", m, input, params, None, None, None)
    H.require_gpu(mlp_packed)
    n_out = mlp_desc.dims[-1]
    out_cols = n_out if out_cols is None else int(out_cols)
    if not 1 <= out_cols <= n_out:
        raise RuntimeError(f"lod_mlp_fwd: out_cols must be in [1, {n_out}]")
    max_level = m.n_levels if max_level is None else int(max_level)
    dev = input.device
    if max_level <= -1:
        raise RuntimeError("lod_mlp_fwd: max_level <= -1 (an all-zero encoding) keeps the two calls")
    native = NATIVE_HALF and params.dtype == torch.float16
    x32 = _f32c(input.detach())
    p32 = params.detach() if native else _p32(params)
    with H.on_device(dev):
        out = H.empty((N, out_cols), dtype=torch.float32, device=dev)
        H.check(H.lib().nr3d_lotd_mlp_forward(C.byref(m._cmeta()), H.ptr(m._dev(dev)), N, H.ptr(x32), H.ptr(p32),
                                              H.F16 if native else H.F32, max_level, C.byref(mlp_desc._c), H.ptr(mlp_packed), H.ptr(out),
                                              out_cols, out_cols, H.stream_of(input)))
    return out


# ------------------------------------------------------------------------------------------------
# lod_bwd  (lotd.cpp:32-35, lotd_torch_api.cu:397-573)
# ------------------------------------------------------------------------------------------------
def lod_bwd(lod_meta, dL_dy, input, params, dy_dx=None, batch_inds=None, batch_offsets=None, batch_data_size=None,
            max_level=None, need_input_grad=None, need_param_grad=None, level_buckets=None, on_bucket=None):
    """``level_buckets`` / ``on_bucket`` (no reference counterpart, non-batched params only): compute dL/dparam in the
    given order of inclusive level ranges ``[(lo, hi), ...]`` and call ``on_bucket(k, grad_slice)`` as soon as bucket k is
    enqueued -- ``grad_slice`` is the contiguous part of dL_dparam that holds those levels, e.g. to start its
    all-reduce while the next bucket is accumulated (nr3d_lotd_bwd_dparam_levels)."""
    if isinstance(lod_meta, tuple):
        from . import _forest
        return _forest.lod_bwd(lod_meta, dL_dy, input, params, dy_dx, batch_inds, batch_offsets, batch_data_size, max_level,
                               need_input_grad, need_param_grad)
    m = lod_meta
    N, bds = _check_common("bwd", m, input, params, batch_inds, batch_offsets, batch_data_size)
    E, D = m.n_encoded_dims, m.n_dims_to_encode
    H.require_gpu(dL_dy, dy_dx)
    if dL_dy.dim() != 2 or tuple(dL_dy.shape) != (N, E):
        raise RuntimeError(f"bwd: Expected dL_dy of size [{N}, {E}], got {list(dL_dy.shape)}")
    if dL_dy.dtype != params.dtype:
        raise RuntimeError("bwd: Expected dL_dy and grid to have the same dtype")
    if dy_dx is not None and dy_dx.dtype != input.dtype:
        raise RuntimeError("bwd: Expected x and dy_dx to have the same dtype")
    max_level = m.n_levels if max_level is None else int(max_level)
    need_input_grad = bool(input.requires_grad) if need_input_grad is None else bool(need_input_grad)
    need_param_grad = bool(params.requires_grad) if need_param_grad is None else bool(need_param_grad)
    dev = input.device
    dL_dx = dL_dparam = None
    with H.on_device(dev):
        if need_input_grad and dy_dx is None:
            raise RuntimeError("LoTDEncoding::bwd: need `dy_dx` to comput `dL_dx`.")
        batched = batch_inds is not None or batch_offsets is not None or bds != 0
        nothing = max_level <= -1 or not (need_input_grad or need_param_grad) or N == 0
        # (float, half, float): half dL_dy is read and half dL_dparam written by the kernels themselves when the meta
        # qualifies and dL_dy is a contiguous [N, E] tensor; dL_dparam then needs no cast and is allocated as half
        native = (_native_half(m, params, batched) and level_buckets is None and USE_BINNED_DPARAM and input.dtype == torch.float32
                  and dL_dy.is_contiguous() and E % 4 == 0 and dL_dy.data_ptr() % 8 == 0 and not nothing
                  and params.shape[0] == m.n_params)
        # the pair-record path defines every element of dL_dparam itself (nr3d_lotd_bwd_dparam_typed, assign): no
        # zero-fill and no read-modify-write of the 46 MiB gradient
        typed = (not nothing and not batched and level_buckets is None and USE_BINNED_DPARAM
                 and (native or params.dtype == torch.float32) and params.shape[0] == m.n_params
                 and bool(H.lib().nr3d_lotd_pair_path_ok(C.byref(m._cmeta()))))
        if need_input_grad:       # fully written by the dL/dx kernel
            dL_dx = (torch.zeros if nothing else H.empty)((N, D), dtype=torch.float32, device=dev)
        if need_param_grad:
            dL_dparam = (H.empty if typed else torch.zeros)((params.shape[0],), dtype=torch.float16 if native else torch.float32,
                                                                device=dev)
        if nothing:
            if need_param_grad and level_buckets is not None and on_bucket is not None:
                for k, (lo, hi) in enumerate(level_buckets):     # every bucket is announced (collectives stay matched)
                    on_bucket(k, dL_dparam[m.level_offsets[int(lo)]:m.level_offsets[min(int(hi), m.n_levels - 1) + 1]])
            return (_cast(dL_dx, input.dtype), _cast(dL_dparam, params.dtype))
        g32 = dL_dy.detach() if native else _f32c(dL_dy.detach())
        gcode = H.F16 if native else H.F32
        gsn, gse = _strides2(g32)
        st = H.stream_of(input)
        tag = ("dx" if need_input_grad else "") + ("dp" if need_param_grad else "")
        with _Prof(m, f"LoTD{D}-bwd-{tag}", N):
            gT = None
            if need_input_grad and N > 0:
                j, jsn, jse = _jac_view(dy_dx.detach(), N, E, D)
                # both gradients wanted: the dL/dx kernel stages dL_dy through LDS anyway and leaves the feature-major
                # copy the atomic-free parameter scatter reads (saves that path's own transposition pass)
                if (need_param_grad and not batched and USE_BINNED_DPARAM and gse == 1 and gsn == E
                        and g32.data_ptr() % 16 == 0):
                    gT = H.empty((E, N), dtype=torch.float32, device=dev)
                H.check(H.lib().nr3d_lotd_bwd_dx(
                    C.byref(m._cmeta()), H.u32(N), C.c_int(H.F32), C.c_int(gcode), H.ptr(g32), H.i64(gsn),
                    H.i64(gse), H.ptr(j), H.i64(jsn), H.i64(jse), H.ptr(dL_dx), H.ptr(gT), st))
            if need_param_grad and N > 0 and typed:
                ws, wsb = _dparam_workspace(m, N, dev, 1)
                if gT is not None:
                    g32, gsn, gse, gcode = gT, 1, N, H.F32
                H.check(H.lib().nr3d_lotd_bwd_dparam_typed(
                    C.byref(m._cmeta()), H.ptr(m._dev(dev)), H.u32(N), C.c_int(gcode), H.ptr(g32), H.i64(gsn), H.i64(gse),
                    H.ptr(_f32c(input.detach())), H.i32(max_level), C.c_int(H.F16 if native else H.F32), C.c_int(1),
                    H.ptr(dL_dparam), H.ptr(ws), C.c_uint64(wsb), st))
            elif need_param_grad and N > 0:
                x32, (p32, pcode) = _f32c(input.detach()), _ptab(params)
                nbat = _n_batches(m, p32, batch_offsets, batched)
                ws, wsb = _dparam_workspace(m, N, dev, nbat)
                if gT is not None:
                    g32, gsn, gse = gT, 1, N
                if level_buckets is None:
                    H.check(H.lib().nr3d_lotd_bwd_dparam(
                        C.byref(m._cmeta()), H.ptr(m._dev(dev)), H.u32(N), C.c_int(H.F32), C.c_int(pcode),
                        H.ptr(g32), H.i64(gsn), H.i64(gse), H.ptr(x32), H.ptr(p32), H.ptr(batch_inds),
                        H.ptr(batch_offsets), H.u32(bds), H.u32(nbat), H.i32(max_level), H.ptr(dL_dparam), H.ptr(ws),
                        C.c_uint64(wsb), st))
                else:
                    if batched:
                        raise RuntimeError("bwd: level_buckets need non-batched params")
                    if params.dtype != torch.float32:
                        raise RuntimeError("bwd: level_buckets need float params (the slices handed out are the result)")
                    seen = set()
                    for k, (lo, hi) in enumerate(level_buckets):
                        lo, hi = int(lo), min(int(hi), m.n_levels - 1)
                        if lo < 0 or lo > hi or seen & set(range(lo, hi + 1)):
                            raise RuntimeError(f"bwd: bad or overlapping level bucket {(lo, hi)}")
                        seen |= set(range(lo, hi + 1))
                        H.check(H.lib().nr3d_lotd_bwd_dparam_levels(
                            C.byref(m._cmeta()), H.ptr(m._dev(dev)), H.u32(N), C.c_int(H.F32), C.c_int(H.F32),
                            H.ptr(g32), H.i64(gsn), H.i64(gse), H.ptr(x32), H.ptr(p32), None, None, H.u32(0),
                            H.u32(nbat), H.i32(lo), H.i32(min(hi, max_level)), H.ptr(dL_dparam), H.ptr(ws),
                            C.c_uint64(wsb), st))
                        if on_bucket is not None:
                            on_bucket(k, dL_dparam[m.level_offsets[lo]:m.level_offsets[hi + 1]])
    return _cast(dL_dx, input.dtype), _cast(dL_dparam, params.dtype)


def _cast(t, dtype):
    if t is None or t.dtype == dtype:
        return t
    return t.to(dtype)


# ------------------------------------------------------------------------------------------------
# lod_bwd_bwd_input  (lotd.cpp:36-40, lotd_torch_api.cu:575-748)
# ------------------------------------------------------------------------------------------------
def lod_bwd_bwd_input(lod_meta, dL_ddLdx, dL_dy, input, params, dy_dx=None, batch_inds=None, batch_offsets=None,
                      batch_data_size=None, max_level=None, need_dLdinput_ddLdoutput=None,
                      need_dLdinput_dparams=None, need_dLdinput_dinput=None):
    if isinstance(lod_meta, tuple):
        from . import _forest
        return _forest.lod_bwd_bwd_input(lod_meta, dL_ddLdx, dL_dy, input, params, dy_dx, batch_inds, batch_offsets,
                                         batch_data_size, max_level, need_dLdinput_ddLdoutput, need_dLdinput_dparams,
                                         need_dLdinput_dinput)
    m = lod_meta
    N, bds = _check_common("bwd_bwd_input", m, input, params, batch_inds, batch_offsets, batch_data_size)
    E, D = m.n_encoded_dims, m.n_dims_to_encode
    H.require_gpu(dL_ddLdx, dL_dy, dy_dx)
    if input.dtype != torch.float32:  # lotd_encoding.h:1805-1810: (half, half) has no second-order path
        raise RuntimeError("LoTDEncoding: Input type combination not supported. Supported types are: "
                           "<input,param> -> (half, half), (float, half), (float, float)")
    if tuple(dL_ddLdx.shape) != (N, D) or not dL_ddLdx.is_contiguous() or dL_ddLdx.dtype != input.dtype:
        raise RuntimeError(f"bwd_bwd_input: Expected contiguous dL_ddLdx of size [{N}, {D}] and the input's dtype")
    if tuple(dL_dy.shape) != (N, E) or dL_dy.dtype != params.dtype:
        raise RuntimeError(f"bwd_bwd_input: Expected dL_dy of size [{N}, {E}] and the params' dtype")
    if dy_dx is not None and dy_dx.dtype != input.dtype:
        raise RuntimeError("bwd_bwd_input: Expected x and dy_dx to have the same dtype")
    max_level = m.n_levels if max_level is None else int(max_level)
    need_dLdy = bool(dL_dy.requires_grad) if need_dLdinput_ddLdoutput is None else bool(need_dLdinput_ddLdoutput)
    need_dx = bool(input.requires_grad) if need_dLdinput_dinput is None else bool(need_dLdinput_dinput)
    need_dp = bool(params.requires_grad) if need_dLdinput_dparams is None else bool(need_dLdinput_dparams)
    dev = input.device
    dL_ddLdy = dL_dparams = dL_dx = None
    with H.on_device(dev):
        if need_dLdy:
            if dy_dx is None:
                raise RuntimeError("LoTDEncoding::bwd_bwd_input: need `dy_dx` to compute `dL_d(dLdy)`.")
            dL_ddLdy = torch.zeros((E, N), dtype=torch.float32, device=dev).t()
        if need_dx:
            dL_dx = torch.zeros((N, D), dtype=torch.float32, device=dev)
        if need_dp:
            dL_dparams = torch.zeros((params.shape[0],), dtype=torch.float32, device=dev)
        if max_level <= -1 or not (need_dLdy or need_dx or need_dp) or N == 0:
            return _cast(dL_ddLdy, dL_dy.dtype), _cast(dL_dparams, params.dtype), _cast(dL_dx, input.dtype)
        st = H.stream_of(input)
        v32 = _f32c(dL_ddLdx.detach())
        g32 = _f32c(dL_dy.detach())
        gsn, gse = _strides2(g32)
        x32, (p32, pcode) = _f32c(input.detach()), _ptab(params)
        cm, md = C.byref(m._cmeta()), H.ptr(m._dev(dev))
        tag = ("dx" if need_dx else "") + ("dp" if need_dp else "") + ("dLdy" if need_dLdy else "")
        with _Prof(m, f"LoTD{D}-bwd2-{tag}", N):
            if need_dLdy:
                j, jsn, jse = _jac_view(dy_dx.detach(), N, E, D)
                H.check(H.lib().nr3d_lotd_bwd_bwd_ddLdy(
                    cm, H.u32(N), C.c_int(H.F32), C.c_int(H.F32), H.ptr(v32), H.ptr(j), H.i64(jsn), H.i64(jse),
                    H.ptr(dL_ddLdy), H.i64(dL_ddLdy.stride(0)), H.i64(dL_ddLdy.stride(1)), st))
            batched = batch_inds is not None or batch_offsets is not None or bds != 0
            wsb = 0
            if need_dx:
                # scratch for the level-parallel form ([n_pseudo, N, D] floats), while it stays below HVP_WORKSPACE_MAX_BYTES
                f = H.lib().nr3d_lotd_bwd_bwd_dx_workspace_bytes
                f.restype = C.c_uint64
                wsb = int(f(cm, H.u32(N)))
            if (need_dx and need_dp and not batched and 0 < wsb <= HVP_WORKSPACE_MAX_BYTES and USE_BINNED_DPARAM
                    and gse == 1 and gsn == E):
                # both level-major passes of this step read dL_dy feature-major: ONE copy for the two of them (each would make
                # its own: the parameter pass a [E, N] transposition, the Hessian pass its by-level pairs)
                gT = H.empty((E, N), dtype=torch.float32, device=dev)
                H.check(H.lib().nr3d_lotd_dLdy_feature_major(H.u32(N), H.u32(E), C.c_int(H.F32), H.ptr(g32), H.i64(gsn), H.i64(gse),
                                                             H.ptr(gT), st))
                g32, gsn, gse = gT, 1, N
            if need_dx:
                ws = (H.empty((wsb + 3) // 4, dtype=torch.float32, device=dev)
                      if 0 < wsb <= HVP_WORKSPACE_MAX_BYTES else None)
                H.check(H.lib().nr3d_lotd_bwd_bwd_dx_ws(
                    cm, md, H.u32(N), C.c_int(H.F32), C.c_int(pcode), H.ptr(v32), H.ptr(g32), H.i64(gsn),
                    H.i64(gse), H.ptr(x32), H.ptr(p32), H.ptr(batch_inds), H.ptr(batch_offsets), H.u32(bds),
                    H.i32(max_level), H.ptr(dL_dx), H.ptr(ws), C.c_uint64(wsb if ws is not None else 0), st))
            if need_dp:
                nbat = _n_batches(m, p32, batch_offsets, batched)
                ws, wsb = _dparam_workspace(m, N, dev, nbat)
                H.check(H.lib().nr3d_lotd_bwd_bwd_dparam(
                    cm, md, H.u32(N), C.c_int(H.F32), C.c_int(pcode), H.ptr(v32), H.ptr(g32), H.i64(gsn),
                    H.i64(gse), H.ptr(x32), H.ptr(p32), H.ptr(batch_inds), H.ptr(batch_offsets), H.u32(bds),
                    H.u32(nbat), H.i32(max_level), H.ptr(dL_dparams), H.ptr(ws), C.c_uint64(wsb), st))
    return _cast(dL_ddLdy, dL_dy.dtype), _cast(dL_dparams, params.dtype), _cast(dL_dx, input.dtype)


# ------------------------------------------------------------------------------------------------
# lod_get_grid_index  (lotd.cpp:41-42, lotd_torch_api.cu:771-855)
# ------------------------------------------------------------------------------------------------
def lod_get_grid_index(lod_meta, input, batch_inds=None, batch_offsets=None, batch_data_size=None, max_level=None):
    if isinstance(lod_meta, tuple):
        from . import _forest
        return _forest.lod_get_grid_index(lod_meta, input, batch_inds, batch_offsets, batch_data_size, max_level)
    m = lod_meta
    N, bds = _check_common("get_grid_index", m, input, None, batch_inds, batch_offsets, batch_data_size)
    for tp in m.level_types:
        if tp not in (int(LoDType.Dense), int(LoDType.Hash)):
            raise RuntimeError("LoTDEncoding::get_grid_index: Only support Dense/Hash type.")
    max_level = m.n_levels if max_level is None else int(max_level)
    E, D = m.n_encoded_dims, m.n_dims_to_encode
    dev = input.device
    with H.on_device(dev):
        out = torch.zeros((N, E, 1 << D), dtype=torch.int64, device=dev)
        if max_level <= -1 or N == 0:
            return out
        H.check(H.lib().nr3d_lotd_grid_index(
            C.byref(m._cmeta()), H.ptr(m._dev(dev)), H.u32(N), C.c_int(H.F32), H.ptr(_f32c(input.detach())),
            H.ptr(batch_inds), H.ptr(batch_offsets), H.u32(bds), H.i32(max_level), H.ptr(out), H.stream_of(input)))
    return out
