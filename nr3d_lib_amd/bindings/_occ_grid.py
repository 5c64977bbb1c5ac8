"""nr3d_lib_amd.bindings._occ_grid -- drop-in for the reference pybind module
``nr3d_lib.bindings._occ_grid`` (csrc/occ_grid/src/occ_grid.cpp:22-33, signatures
csrc/occ_grid/include/occ_grid/cpp_api.h:14-66), backed by libnr3d_hip.so.

ray_marching / batched_ray_marching / forest_ray_marching keep the reference's positional signatures and return lists.
The per-ray counts are scanned on the device; the only host sync is the read-back of the total
sample count (needed to size the outputs -- the reference syncs at the same point).
"""
import ctypes as C
import enum

import torch

from .. import _hip as H


class ContractionType(enum.IntEnum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


AABB = ContractionType.AABB
UN_BOUNDED_TANH = ContractionType.UN_BOUNDED_TANH
UN_BOUNDED_SPHERE = ContractionType.UN_BOUNDED_SPHERE

# largest per-call sample cache ([n_rays, max_steps] x 12 B) the binding allocates; beyond it the op marches twice
SAMPLE_CACHE_MAX_BYTES = 2 << 30


def _chk(name, t, dim=None, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dim is not None and t.dim() != dim:
        raise RuntimeError(f"{name}: expected a {dim}-D tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")


_SCAN_TMP = {}


def _scan_tmp_bytes(n):
    """nr3d_scan_tmp_bytes(n), remembered per n (a binding call costs more than the lookup)"""
    b = _SCAN_TMP.get(n)
    if b is None:
        b = _SCAN_TMP[n] = int(H.lib().nr3d_scan_tmp_bytes(C.c_uint64(max(n, 1))))
    return b


def _march(rays_o, rays_d, t_min, t_max, batch_inds, batch_data_size, roi, grid_binary, contraction_type,
           step_size, max_step_size, dt_gamma, max_steps, return_gidx, batched, finish=False):
    _chk("rays_o", rays_o, 2, torch.float32)
    _chk("rays_d", rays_d, 2, torch.float32)
    _chk("t_min", t_min, 1, torch.float32)
    _chk("t_max", t_max, 1, torch.float32)
    _chk("roi", roi, 2 if batched else 1, torch.float32)
    _chk("grid_binary", grid_binary, 4 if batched else 3)
    if grid_binary.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("grid_binary: expected a bool tensor")
    if rays_o.shape[1] != 3 or rays_d.shape[1] != 3 or rays_d.shape[0] != rays_o.shape[0]:
        raise RuntimeError("rays_o / rays_d: expected shape [n_rays, 3]")
    n = rays_o.shape[0]
    if t_min.shape[0] != n or t_max.shape[0] != n:
        raise RuntimeError("t_min / t_max: expected shape [n_rays]")
    if batched:
        if roi.shape[1] != 6 or roi.shape[0] != grid_binary.shape[0]:
            raise RuntimeError("roi: expected shape [B, 6] matching grid_binary's batch dim")
        if batch_inds is not None:
            _chk("batch_inds", batch_inds, 1, torch.int32)
            if batch_inds.shape[0] != n:
                raise RuntimeError("batch_inds: expected shape [n_rays]")
        bds = int(batch_data_size or 0)
        if not (bds == 0 or n % bds == 0):
            raise RuntimeError(f"batched_ray_marching: Expect nonzero `batch_data_size`={bds} to be a divisor of "
                               f"`n_rays`={n}")
    else:
        if roi.shape[0] != 6:
            raise RuntimeError("roi: expected shape [6]")
        bds = 0
    dev = rays_o.device
    res = (C.c_int32 * 3)(*[int(s) for s in grid_binary.shape[-3:]])
    ctype = C.c_int(int(contraction_type))
    with H.on_device(dev):
        st = H.stream_of(rays_o)
        packed_info = H.empty((n, 2), dtype=torch.int32, device=dev)
        total = None if finish else H.host_i64(1, dev)      # pinned host word the scan writes into
        tmp = H.empty((_scan_tmp_bytes(n) + 7) // 8, dtype=torch.int64, device=dev)
        # sample cache: the count pass keeps every sample, the emit pass only compacts (no second march)
        cache_bytes = n * int(max_steps) * 12
        cache = (H.empty((cache_bytes + 3) // 4, dtype=torch.int32, device=dev)
                 if 0 < cache_bytes <= SAMPLE_CACHE_MAX_BYTES else None)
        if finish:
            # the hit rays' compaction only needs the counts: the scan that turns them into packed_info compacts the hit rays
            # as well, and the readback fetches the number of samples and of hit rays together -- ONE device->host sync
            ridx_hit = H.empty(n, dtype=torch.int64, device=dev)
            pack_infos = H.empty((n, 2), dtype=torch.int64, device=dev)
            totals = H.host_i64(2, dev)
            H.check(H.lib().nr3d_ray_marching_count_finished(
                H.u32(n), H.ptr(rays_o), H.ptr(rays_d), H.ptr(t_min), H.ptr(t_max), H.ptr(roi), res, H.ptr(grid_binary),
                ctype, H.f32(step_size), H.f32(max_step_size), H.f32(dt_gamma), H.u32(max_steps), C.c_int(int(batched)),
                H.ptr(batch_inds), H.u32(bds), H.ptr(packed_info), H.ptr(ridx_hit), H.ptr(pack_infos), H.ptr(totals), H.ptr(tmp),
                H.ptr(cache), C.c_uint64(cache_bytes if cache is not None else 0), st))
            S, n_hit = H.wait_i64(totals, dev)
        else:
            H.check(H.lib().nr3d_ray_marching_count(
                H.u32(n), H.ptr(rays_o), H.ptr(rays_d), H.ptr(t_min), H.ptr(t_max), H.ptr(roi), res, H.ptr(grid_binary),
                ctype, H.f32(step_size), H.f32(max_step_size), H.f32(dt_gamma), H.u32(max_steps), C.c_int(int(batched)),
                H.ptr(batch_inds), H.u32(bds), H.ptr(packed_info), H.ptr(total), H.ptr(tmp), H.ptr(cache),
                C.c_uint64(cache_bytes if cache is not None else 0), st))
            S = H.wait_i64(total, dev)[0]  # the single device->host sync of this op
        t_starts = H.empty((S, 1), dtype=torch.float32, device=dev)
        t_ends = H.empty((S, 1), dtype=torch.float32, device=dev)
        ridx = H.empty(S, dtype=torch.int32, device=dev)
        bidx = H.empty(S, dtype=torch.int32, device=dev) if batched else None
        gidx = H.empty(S, dtype=torch.int32, device=dev) if return_gidx else None
        if finish and cache is not None:
            # the cached emit is a per-ray copy; the per-sample epilogue rides on it (one launch, one op)
            ridx64 = H.empty(S, dtype=torch.int64, device=dev)
            deltas = H.empty(S, dtype=torch.float32, device=dev)
            samples = H.empty((S, 3), dtype=torch.float32, device=dev)
            if S > 0:
                H.check(H.lib().nr3d_ray_marching_emit_finished(
                    H.u32(n), H.ptr(rays_o), H.ptr(rays_d), C.c_int(int(batched)), H.ptr(batch_inds), H.u32(bds),
                    H.ptr(packed_info), H.ptr(cache), H.u32(max_steps), H.ptr(t_starts), H.ptr(t_ends), H.ptr(ridx), H.ptr(bidx),
                    H.ptr(gidx), H.ptr(ridx64), H.ptr(deltas), H.ptr(samples), st))
            return dict(n_hit=n_hit, ridx_hit=ridx_hit[:n_hit], pack_infos=H.mark_ordered(pack_infos[:n_hit], total=S), t_starts=t_starts.view(-1),
                        t_ends=t_ends.view(-1), ridx=ridx64, deltas=deltas, samples=samples, bidx=bidx, gidx=gidx)
        if S > 0:
            H.check(H.lib().nr3d_ray_marching_emit(
                H.u32(n), H.ptr(rays_o), H.ptr(rays_d), H.ptr(t_min), H.ptr(t_max), H.ptr(roi), res,
                H.ptr(grid_binary), ctype, H.f32(step_size), H.f32(max_step_size), H.f32(dt_gamma),
                C.c_int(int(batched)), H.ptr(batch_inds), H.u32(bds), H.ptr(packed_info), H.ptr(t_starts),
                H.ptr(t_ends), H.ptr(ridx), H.ptr(bidx), H.ptr(gidx), H.ptr(cache), H.u32(max_steps), st))
        if finish:
            ridx64 = H.empty(S, dtype=torch.int64, device=dev)
            deltas = H.empty(S, dtype=torch.float32, device=dev)
            samples = H.empty((S, 3), dtype=torch.float32, device=dev)
            H.check(H.lib().nr3d_march_finish_samples(C.c_uint64(S), H.ptr(rays_o), H.ptr(rays_d), H.ptr(ridx), H.ptr(t_starts),
                                                      H.ptr(t_ends), H.ptr(ridx64), H.ptr(deltas), H.ptr(samples), st))
            return dict(n_hit=n_hit, ridx_hit=ridx_hit[:n_hit], pack_infos=H.mark_ordered(pack_infos[:n_hit], total=S), t_starts=t_starts.view(-1),
                        t_ends=t_ends.view(-1), ridx=ridx64, deltas=deltas, samples=samples, bidx=bidx, gidx=gidx)
    if batched:
        return [packed_info, t_starts, t_ends, ridx, bidx, gidx]
    return [packed_info, t_starts, t_ends, ridx, gidx]


def ray_marching(rays_o, rays_d, t_min, t_max, roi, grid_binary, contraction_type, step_size, max_step_size,
                 dt_gamma, max_steps, return_gidx):
    """-> [packed_info i32 [n,2], t_starts f32 [S,1], t_ends f32 [S,1], ridx i32 [S], gidx i32 [S] | None]
    (ray_marching.cu:136-244)"""
    return _march(rays_o, rays_d, t_min, t_max, None, None, roi, grid_binary, contraction_type, step_size,
                  max_step_size, dt_gamma, max_steps, return_gidx, False)


def ray_marching_finished(rays_o, rays_d, t_min, t_max, roi, grid_binary, contraction_type, step_size, max_step_size,
                          dt_gamma, max_steps, return_gidx, batch_inds=None, batch_data_size=None):
    """ray_marching / batched_ray_marching (``roi`` [B, 6]) + the post-processing every caller of the reference applies to
    its outputs (occgrid_raymarch.py:87-112): -> dict(n_hit, ridx_hit int64 [n_hit], pack_infos int64 [n_hit, 2],
    t_starts / t_ends f32 [S], ridx int64 [S], deltas = t_ends - t_starts, samples [S, 3] = rays_o + rays_d * t_starts,
    bidx i32 [S] | None, gidx i32 [S] | None).  One device->host sync (samples and hit rays together) and 2 launches
    instead of ~12 ATen ops and a second sync (nonzero)."""
    batched = roi.dim() == 2
    return _march(rays_o, rays_d, t_min, t_max, batch_inds, batch_data_size, roi, grid_binary, contraction_type, step_size,
                  max_step_size, dt_gamma, max_steps, return_gidx, batched, finish=True)


def batched_ray_marching(rays_o, rays_d, t_min, t_max, batch_inds_, batch_data_size_, roi, grid_binary,
                         contraction_type, step_size, max_step_size, dt_gamma, max_steps, return_gidx):
    """-> [packed_info, t_starts, t_ends, ridx, bidx, gidx | None]  (batched_marching.cu:153-287)"""
    return _march(rays_o, rays_d, t_min, t_max, batch_inds_, batch_data_size_, roi, grid_binary, contraction_type,
                  step_size, max_step_size, dt_gamma, max_steps, return_gidx, True)


def forest_ray_marching(forest, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits, seg_pack_infos,
                        grid_binary, step_size, max_step_size, dt_gamma, max_steps, return_gidx):
    """-> [packed_info, t_starts, t_ends, ridx, blidx, gidx | None]  (forest_marching.cu:145-303; `forest` is a
    bindings._forest.ForestMeta, the segments are the caller's per-ray block crossings, packed by seg_pack_infos)"""
    _chk("rays_o", rays_o, 2, torch.float32)
    _chk("rays_d", rays_d, 2, torch.float32)
    _chk("t_min", t_min, 1, torch.float32)
    _chk("t_max", t_max, 1, torch.float32)
    _chk("seg_block_inds", seg_block_inds, 1, torch.int32)
    _chk("seg_entries", seg_entries, 1, torch.float32)
    _chk("seg_exits", seg_exits, 1, torch.float32)
    _chk("seg_pack_infos", seg_pack_infos, 2, torch.int32)
    _chk("grid_binary", grid_binary, 4)
    if grid_binary.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("grid_binary: expected a bool tensor")
    n = rays_o.shape[0]
    if tuple(rays_o.shape) != (n, 3) or tuple(rays_d.shape) != (n, 3):
        raise RuntimeError("rays_o / rays_d: expected shape [n_rays, 3]")
    if t_min.shape[0] != n or t_max.shape[0] != n or tuple(seg_pack_infos.shape) != (n, 2):
        raise RuntimeError("t_min / t_max / seg_pack_infos: expected n_rays rows")
    if seg_entries.shape != seg_block_inds.shape or seg_exits.shape != seg_block_inds.shape:
        raise RuntimeError("seg_block_inds / seg_entries / seg_exits: expected the same size")
    forest._check("forest_ray_marching", rays_o)
    if grid_binary.shape[0] != int(forest.n_trees):
        raise RuntimeError(f"grid_binary: expected {int(forest.n_trees)} block grids, got {grid_binary.shape[0]}")
    dev = rays_o.device
    res = (C.c_int32 * 3)(*[int(v) for v in grid_binary.shape[-3:]])
    with H.on_device(dev):
        st = H.stream_of(rays_o)
        fc = forest._c()
        packed_info = H.empty((n, 2), dtype=torch.int32, device=dev)
        total = H.host_i64(1, dev)
        nbytes = int(H.lib().nr3d_scan_tmp_bytes(C.c_uint64(max(n, 1))))
        tmp = H.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        common = (H.ptr(rays_o), H.ptr(rays_d), H.ptr(t_min), H.ptr(t_max), H.ptr(seg_block_inds), H.ptr(seg_entries),
                  H.ptr(seg_exits), H.ptr(seg_pack_infos), res, H.ptr(grid_binary), H.f32(step_size),
                  H.f32(max_step_size), H.f32(dt_gamma))
        H.check(H.lib().nr3d_forest_ray_marching_count(C.byref(fc), H.u32(n), *common, H.u32(max_steps),
                                                       H.ptr(packed_info), H.ptr(total), H.ptr(tmp), st))
        S = H.wait_i64(total, dev)[0]  # the single device->host sync of this op
        t_starts = H.empty((S, 1), dtype=torch.float32, device=dev)
        t_ends = H.empty((S, 1), dtype=torch.float32, device=dev)
        ridx = H.empty(S, dtype=torch.int32, device=dev)
        blidx = H.empty(S, dtype=torch.int32, device=dev)
        gidx = H.empty(S, dtype=torch.int32, device=dev) if return_gidx else None
        if S > 0:
            H.check(H.lib().nr3d_forest_ray_marching_emit(C.byref(fc), H.u32(n), *common, H.ptr(packed_info),
                                                          H.ptr(t_starts), H.ptr(t_ends), H.ptr(ridx), H.ptr(blidx),
                                                          H.ptr(gidx), st))
    return [packed_info, t_starts, t_ends, ridx, blidx, gidx]
