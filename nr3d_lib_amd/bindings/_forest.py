"""nr3d_lib_amd.bindings._forest -- drop-in for the reference pybind module ``nr3d_lib.bindings._forest``
(csrc/forest/forest.cpp:24-41): the ``ForestMeta`` record, plus the forest overloads of the LoTD ops that
``nr3d_lib.bindings._lotd`` exposes under the same names when ``metas`` is a ``(lod_meta, forest_meta)`` tuple
(csrc/lotd/src/lotd.cpp:44-60) -- ``_lotd.lod_fwd / lod_bwd / lod_bwd_bwd_input`` dispatch here.

``raytrace_cuda_fixed`` (the reference's patched copy of kaolin's SPC ray trace, forest.cpp:40) is third-party code
outside the path and is not provided; ``forest_identify`` is the octree point query the reference reaches through
kaolin's ``unbatched_query``.
"""
import ctypes as C

import torch

from .. import _hip as H

__all__ = ['ForestMeta', 'forest_identify']


class _CForest(C.Structure):
    _fields_ = [("octree", C.c_void_p), ("exsum", C.c_void_p), ("block_ks", C.c_void_p),
                ("world_block_size", C.c_float * 3), ("world_origin", C.c_float * 3), ("resolution", C.c_int32 * 3),
                ("n_trees", C.c_uint32), ("level", C.c_uint32), ("level_poffset", C.c_uint32),
                ("continuity_enabled", C.c_int32)]


class ForestMeta:
    """forest_cpp_api.h:18-37: read/write attributes, filled by the owner of the block space"""

    def __init__(self):
        self.octree = None                 # uint8 [n_nodes]
        self.exsum = None                  # int32 [n_nodes + 1]
        self.block_ks = None               # int16 [n_trees, 3]
        self.world_block_size = [1.0, 1.0, 1.0]
        self.world_origin = [0.0, 0.0, 0.0]
        self.resolution = [1, 1, 1]
        self.n_trees = 0
        self.level = 0
        self.level_poffset = 0
        self.continuity_enabled = True

    def _check(self, fn, like=None):
        """the argument checks of lotd_torch_api.cu:335-347"""
        for name, dt, dim in (("octree", torch.uint8, 1), ("exsum", torch.int32, 1), ("block_ks", torch.int16, 2)):
            t = getattr(self, name)
            if not torch.is_tensor(t):
                raise RuntimeError(f"{fn}: forest.{name} is not set")
            if t.dim() != dim:
                raise RuntimeError(f"{fn}: Expected {dim}-dimensional tensor for argument forest.{name}, got {t.dim()}")
            if t.dtype != dt:
                raise RuntimeError(f"{fn}: Expected forest.{name} to have scalar type {dt}; got {t.dtype}")
            if not t.is_contiguous():
                raise RuntimeError(f"{fn}: Expected contiguous tensor for argument forest.{name}")
            if like is not None and t.device != like.device:
                raise RuntimeError(f"{fn}: Expected forest.{name} on the same GPU as the input")
        H.require_gpu(self.octree, self.exsum, self.block_ks)
        if tuple(self.block_ks.shape) != (int(self.n_trees), 3):
            raise RuntimeError(f"{fn}: Expected forest.block_ks of size [{int(self.n_trees)}, 3], got {list(self.block_ks.shape)}")

    def _c(self):
        c = _CForest()
        c.octree, c.exsum, c.block_ks = self.octree.data_ptr(), self.exsum.data_ptr(), self.block_ks.data_ptr()
        for d in range(3):
            c.world_block_size[d] = float(self.world_block_size[d])
            c.world_origin[d] = float(self.world_origin[d])
            c.resolution[d] = int(self.resolution[d])
        c.n_trees, c.level, c.level_poffset = int(self.n_trees), int(self.level), int(self.level_poffset)
        c.continuity_enabled = int(bool(self.continuity_enabled))
        return c


def forest_identify(forest: ForestMeta, ks: torch.Tensor) -> torch.Tensor:
    """block index (int32, -1 = no such block) of integer block coordinates ks [..., 3]"""
    forest._check("forest_identify", ks)
    H.require_gpu(ks)
    k16 = ks.reshape(-1, 3).to(torch.int16).contiguous()
    out = H.empty(k16.shape[0], dtype=torch.int32, device=ks.device)
    with H.on_device(ks.device):
        c = forest._c()
        H.check(H.lib().nr3d_forest_identify(C.byref(c), C.c_uint64(k16.shape[0]), H.ptr(k16), H.ptr(out), H.stream_of(ks)))
    return out.view(ks.shape[:-1])


# ------------------------------------------------------------------------------------------------
# forest overloads of the LoTD ops (lotd_torch_api.cu:333-361, :514-533, :705-745)
# ------------------------------------------------------------------------------------------------
def _check(fn, metas, input, params, block_inds, block_offsets, batch_data_size):
    from . import _lotd
    if not (isinstance(metas, tuple) and len(metas) == 2):
        raise RuntimeError(f"{fn}: `metas` should be a tuple of (lod_meta, forest_meta)")
    m, fo = metas
    N, bds = _lotd._check_common(fn, m, input, params, block_inds, block_offsets, batch_data_size)
    fo._check(fn, input)
    if m.n_dims_to_encode != 3:
        raise RuntimeError("LoTDEncoding::fwd: lotd-forest only supports `n_dims_to_encode`==3")
    if block_offsets is not None and block_offsets.shape[0] != int(fo.n_trees):
        raise RuntimeError(f"{fn}: Expected batch_offset of size [{int(fo.n_trees)}], got {list(block_offsets.shape)}")
    return m, fo, N, bds


_workspaces = {}


def _workspace(m, fo, N, dev):
    """scratch of the atomic-free parameter-gradient path (blocks play the role of batch entries; every 3-D level type
    of the forest kernels: per-corner records, nr3d_lotd_forest_dparam_workspace_bytes); (None, 0): global atomics"""
    from . import _lotd
    if not _lotd.USE_BINNED_DPARAM:
        return None, 0
    H.lib().nr3d_lotd_forest_dparam_workspace_bytes.restype = C.c_uint64
    need = int(H.lib().nr3d_lotd_forest_dparam_workspace_bytes(C.byref(m._cmeta()), H.u32(N), H.u32(int(fo.n_trees))))
    if need == 0:
        return None, 0
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        _workspaces.pop(key, None)
        ws = H.empty(need, dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws, need


def lod_fwd(metas, input, params, batch_inds=None, batch_offsets=None, batch_data_size=None, max_level=None,
            need_input_grad=None):
    from . import _lotd
    m, fo, N, bds = _check("fwd", metas, input, params, batch_inds, batch_offsets, batch_data_size)
    max_level = m.n_levels if max_level is None else int(max_level)
    need_input_grad = bool(input.requires_grad) if need_input_grad is None else bool(need_input_grad)
    E, dev = m.n_encoded_dims, input.device
    if max_level <= -1:
        return (torch.zeros((N, E), dtype=params.dtype, device=dev), torch.zeros((N, E * 3), dtype=input.dtype, device=dev))
    x32, (p32, pcode) = _lotd._f32c(input.detach()), _lotd._ptab(params)
    with H.on_device(dev):
        # feature-major storage behind [N, E] / [N, E, 3] views, like the single-block path: coalesced stores
        y = H.empty((E, N), dtype=torch.float32, device=dev).t()
        dy_dx, dsn, dse = None, 0, 0
        if need_input_grad:
            if m.c_permute_dydx:
                dy_dx = H.empty((E, N, 3), dtype=torch.float32, device=dev).permute(1, 0, 2)
                dsn, dse = dy_dx.stride(0), dy_dx.stride(1)
            else:
                dy_dx, dsn, dse = H.empty((N, E * 3), dtype=torch.float32, device=dev), E * 3, 3
        c = fo._c()
        H.check(H.lib().nr3d_lotd_forest_fwd(
            C.byref(m._cmeta()), H.ptr(m._dev(dev)), C.byref(c), H.u32(N), H.ptr(x32), H.ptr(p32), C.c_int(pcode), H.ptr(batch_inds),
            H.ptr(batch_offsets), H.u32(bds), H.i32(max_level), H.ptr(y), H.i64(y.stride(0)), H.i64(y.stride(1)),
            H.ptr(dy_dx), H.i64(dsn), H.i64(dse), H.stream_of(input)))
    return _lotd._cast(y, params.dtype), _lotd._cast(dy_dx, input.dtype)


def lod_bwd(metas, dL_dy, input, params, dy_dx=None, batch_inds=None, batch_offsets=None, batch_data_size=None,
            max_level=None, need_input_grad=None, need_param_grad=None):
    from . import _lotd
    m, fo, N, bds = _check("bwd", metas, input, params, batch_inds, batch_offsets, batch_data_size)
    E, dev = m.n_encoded_dims, input.device
    H.require_gpu(dL_dy, dy_dx)
    if tuple(dL_dy.shape) != (N, E):
        raise RuntimeError(f"bwd: Expected dL_dy of size [{N}, {E}], got {list(dL_dy.shape)}")
    if dL_dy.dtype != params.dtype:
        raise RuntimeError("bwd: Expected dL_dy and grid to have the same dtype")
    max_level = m.n_levels if max_level is None else int(max_level)
    need_input_grad = bool(input.requires_grad) if need_input_grad is None else bool(need_input_grad)
    need_param_grad = bool(params.requires_grad) if need_param_grad is None else bool(need_param_grad)
    dL_dx = dL_dparam = None
    with H.on_device(dev):
        if need_input_grad:
            if dy_dx is None:
                raise RuntimeError("LoTDEncoding::bwd: need `dy_dx` to comput `dL_dx`.")
            dL_dx = torch.zeros((N, 3), dtype=torch.float32, device=dev)
        if need_param_grad:
            dL_dparam = torch.zeros((params.shape[0],), dtype=torch.float32, device=dev)
        if max_level <= -1 or N == 0 or not (need_input_grad or need_param_grad):
            return _lotd._cast(dL_dx, input.dtype), _lotd._cast(dL_dparam, params.dtype)
        g32 = _lotd._f32c(dL_dy.detach()).contiguous()
        st = H.stream_of(input)
        if need_input_grad:
            j, jsn, jse = _lotd._jac_view(dy_dx.detach(), N, E, 3)
            H.check(H.lib().nr3d_lotd_bwd_dx(
                C.byref(m._cmeta()), H.u32(N), C.c_int(H.F32), C.c_int(H.F32), H.ptr(g32), H.i64(E), H.i64(1),
                H.ptr(j), H.i64(jsn), H.i64(jse), H.ptr(dL_dx), None, st))
        if need_param_grad:
            x32, (p32, pcode) = _lotd._f32c(input.detach()), _lotd._ptab(params)
            c = fo._c()
            ws, wsb = _workspace(m, fo, N, dev)
            H.check(H.lib().nr3d_lotd_forest_bwd_dparam(
                C.byref(m._cmeta()), H.ptr(m._dev(dev)), C.byref(c), H.u32(N), None, H.ptr(g32), H.ptr(x32), H.ptr(p32), C.c_int(pcode),
                H.ptr(batch_inds), H.ptr(batch_offsets), H.u32(bds), H.i32(max_level), H.ptr(dL_dparam), H.ptr(ws),
                C.c_uint64(wsb), st))
    return _lotd._cast(dL_dx, input.dtype), _lotd._cast(dL_dparam, params.dtype)


def lod_bwd_bwd_input(metas, dL_ddLdx, dL_dy, input, params, dy_dx=None, batch_inds=None, batch_offsets=None,
                      batch_data_size=None, max_level=None, need_dLdinput_ddLdoutput=None, need_dLdinput_dparams=None,
                      need_dLdinput_dinput=None):
    from . import _lotd
    m, fo, N, bds = _check("bwd_bwd_input", metas, input, params, batch_inds, batch_offsets, batch_data_size)
    E, dev = m.n_encoded_dims, input.device
    H.require_gpu(dL_ddLdx, dL_dy, dy_dx)
    if input.dtype != torch.float32:
        raise RuntimeError("LoTDEncoding: Input type combination not supported. Supported types are: "
                           "<input,param> -> (half, half), (float, half), (float, float)")
    if tuple(dL_ddLdx.shape) != (N, 3) or not dL_ddLdx.is_contiguous() or dL_ddLdx.dtype != input.dtype:
        raise RuntimeError(f"bwd_bwd_input: Expected contiguous dL_ddLdx of size [{N}, 3] and the input's dtype")
    if tuple(dL_dy.shape) != (N, E) or dL_dy.dtype != params.dtype:
        raise RuntimeError(f"bwd_bwd_input: Expected dL_dy of size [{N}, {E}] and the params' dtype")
    max_level = m.n_levels if max_level is None else int(max_level)
    need_dLdy = bool(dL_dy.requires_grad) if need_dLdinput_ddLdoutput is None else bool(need_dLdinput_ddLdoutput)
    need_dx = bool(input.requires_grad) if need_dLdinput_dinput is None else bool(need_dLdinput_dinput)
    need_dp = bool(params.requires_grad) if need_dLdinput_dparams is None else bool(need_dLdinput_dparams)
    dL_ddLdy = dL_dparams = dL_dx = None
    with H.on_device(dev):
        if need_dLdy:
            if dy_dx is None:
                raise RuntimeError("LoTDEncoding::bwd_bwd_input: need `dy_dx` to compute `dL_d(dLdy)`.")
            dL_ddLdy = torch.zeros((N, E), dtype=torch.float32, device=dev)
        if need_dx:
            dL_dx = torch.zeros((N, 3), dtype=torch.float32, device=dev)
        if need_dp:
            dL_dparams = torch.zeros((params.shape[0],), dtype=torch.float32, device=dev)
        if max_level <= -1 or N == 0 or not (need_dLdy or need_dx or need_dp):
            return _lotd._cast(dL_ddLdy, dL_dy.dtype), _lotd._cast(dL_dparams, params.dtype), _lotd._cast(dL_dx, input.dtype)
        st = H.stream_of(input)
        v32, g32 = _lotd._f32c(dL_ddLdx.detach()), _lotd._f32c(dL_dy.detach()).contiguous()
        x32, (p32, pcode) = _lotd._f32c(input.detach()), _lotd._ptab(params)
        cm, md, c = C.byref(m._cmeta()), H.ptr(m._dev(dev)), fo._c()
        if need_dLdy:
            j, jsn, jse = _lotd._jac_view(dy_dx.detach(), N, E, 3)
            H.check(H.lib().nr3d_lotd_bwd_bwd_ddLdy(
                cm, H.u32(N), C.c_int(H.F32), C.c_int(H.F32), H.ptr(v32), H.ptr(j), H.i64(jsn), H.i64(jse),
                H.ptr(dL_ddLdy), H.i64(E), H.i64(1), st))
        if need_dx:
            H.check(H.lib().nr3d_lotd_forest_bwd_bwd_dx(
                cm, md, C.byref(c), H.u32(N), H.ptr(v32), H.ptr(g32), H.ptr(x32), H.ptr(p32), C.c_int(pcode), H.ptr(batch_inds),
                H.ptr(batch_offsets), H.u32(bds), H.i32(max_level), H.ptr(dL_dx), st))
        if need_dp:
            ws, wsb = _workspace(m, fo, N, dev)
            H.check(H.lib().nr3d_lotd_forest_bwd_dparam(
                cm, md, C.byref(c), H.u32(N), H.ptr(v32), H.ptr(g32), H.ptr(x32), H.ptr(p32), C.c_int(pcode), H.ptr(batch_inds),
                H.ptr(batch_offsets), H.u32(bds), H.i32(max_level), H.ptr(dL_dparams), H.ptr(ws), C.c_uint64(wsb), st))
    return _lotd._cast(dL_ddLdy, dL_dy.dtype), _lotd._cast(dL_dparams, params.dtype), _lotd._cast(dL_dx, input.dtype)


def lod_get_grid_index(metas, input, batch_inds=None, batch_offsets=None, batch_data_size=None, max_level=None):
    raise RuntimeError("LoTDEncoding::lod_get_grid_index: Not implemented for forest for now")   # lotd_torch_api.cu:840
