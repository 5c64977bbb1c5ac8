"""nr3d_lib_amd.bindings._occ_grid -- drop-in for the reference pybind module
``nr3d_lib.bindings._occ_grid`` (csrc/occ_grid/src/occ_grid.cpp:22-33, signatures
csrc/occ_grid/include/occ_grid/cpp_api.h:14-66), backed by libnr3d_hip.so.

ray_marching / batched_ray_marching / forest_ray_marching keep the reference's positional signatures and return lists.
The per-ray counts are scanned on the device; the only host sync is the read-back of the total
sample count (needed to size the outputs -- the reference syncs at the same point).
"""
import ctypes as C
import enum
import threading

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


# ---- configs[2] in one call (round 6) -------------------------------------------------------------------------------------------
# largest bound (n_rays * max_steps rows x ~72 B) the one-call path allocates; above it the two-phase chain (count -> readback ->
# emit -> alpha -> composite) runs, whose buffers are sized by the actual sample count
FUSED_MARCH_COMPOSITE_MAX_BYTES = 512 << 20


_totals_tls = threading.local()
_TOTALS_LIMBO = []       # pinned words of handles dropped before their totals were read: kept alive (a late kernel may still write them)


def _take_totals():
    """a pinned int64 [2] of this handle's own (two outstanding handles must not share the word the kernels write into)"""
    free = getattr(_totals_tls, "free", None)
    if free is None:
        free = _totals_tls.free = []
    return free.pop() if free else torch.empty(2, dtype=torch.int64, pin_memory=True)


class MarchComposite:
    """What ``ray_marching_composite`` returns: the buffers of ONE march + alpha composite whose kernels are (being) enqueued, and the
    two totals that are read back LAZILY -- ``totals()`` is the op's single device->host sync, and everything that does not need a
    tensor SHAPE (enqueueing ``backward``) can happen before it.  Tensor views are built on demand (``view(name)``: exactly-sized;
    the launch-bound caller pays only for what it looks at):
      per ray   packed_info int32 [n, 2], mask / depth [n], rgb [n, 3] (zeros for rays without samples), ridx_hit int64 [n_hit],
                pack_infos int64 [n_hit, 2] (tagged ordered, tiling [0, S));
      per sample [S]: t_starts, t_ends, ridx (int64), ridx32, gidx, deltas, samples [S, 3], alpha, vw.
    ``result()``: all of them in a dict (+ n_hit)."""

    _SPEC = dict(packed_info=(torch.int32, 2, "n"), ridx_hit=(torch.int64, 0, "hit"), pack_infos=(torch.int64, 2, "hit"),
                 mask=(torch.float32, 0, "n"), depth=(torch.float32, 0, "n"), rgb_out=(torch.float32, 3, "n"),
                 t_starts=(torch.float32, 0, "S"), t_ends=(torch.float32, 0, "S"), ridx32=(torch.int32, 0, "S"),
                 gidx=(torch.int32, 0, "S"), ridx=(torch.int64, 0, "S"), deltas=(torch.float32, 0, "S"),
                 samples=(torch.float32, 3, "S"), alpha=(torch.float32, 0, "S"), vw=(torch.float32, 0, "S"))

    def __init__(self, dev, n):
        self.dev, self.n, self._tot, self._host, self.g, self._gpool = dev, n, None, None, None, None

    def totals(self):
        """(S, n_hit): THE device->host sync of the op (waits on the stream the launches went to)"""
        if self._tot is None:
            # the scan (second launch) stores both totals into this pinned word pair with system scope: poll it instead of draining
            # the stream -- the host goes on to build its views while the composite kernels still run (everything the caller does
            # with the tensors is stream-ordered behind them anyway).  Timeout (never seen; e.g. a non-coherent pinned pool): drain.
            if H.lib().nr3d_wait_host_words(self._host.data_ptr(), 2, -1, 5000) != 0:
                cur = torch.cuda.current_stream(self.dev)
                if cur.cuda_stream != self._raw_stream:      # the caller changed streams since the launches: wait on THEIR stream
                    cur = torch.cuda.ExternalStream(self._raw_stream, device=self.dev)
                cur.synchronize()
            S, n_hit = self._host.tolist()
            _totals_tls.free.append(self._host)          # read: the word can serve the next handle of this thread
            self._host = None
            self._tot = (int(S), int(n_hit))
            if S > self.sigma_rows:
                raise RuntimeError(f"ray_marching_composite: the march produced {S} samples but sigma has {self.sigma_rows} rows")
        return self._tot

    def __del__(self):
        if getattr(self, "_host", None) is not None:
            _TOTALS_LIMBO.append(self._host)

    def view(self, name):
        if name == "rgb":
            name = "rgb_out"
        dtype, inner, kind = self._SPEC[name]
        o, nb = self._offs[name]
        if nb == 0:
            return None
        rows = self.n if kind == "n" else self.totals()[0 if kind == "S" else 1]
        esz = 8 if dtype == torch.int64 else 4
        t = self._pool[o:o + rows * max(inner, 1) * esz].view(dtype)
        t = t.view(rows, inner) if inner else t
        if name == "pack_infos":
            H.mark_ordered(t, total=self.totals()[0])
        return t

    def result(self):
        out = {k: self.view(k) for k in self._SPEC}
        out["rgb"] = out.pop("rgb_out")
        out["n_hit"] = self.totals()[1]
        return out

    def backward(self, g_mask, g_depth, g_rgb, need_t=True, need_rgb=True, need_sigma=False):
        """enqueue the composite's backward (no sync): afterwards ``grads()`` -> (grad_alpha [S], grad_t [S] | None, grad_rgb [S, 3] |
        None, grad_sigma [S] | None)"""
        n, dev = self.n, self.dev
        H.require_gpu(g_mask, g_depth, g_rgb)
        for name, t, inner in (("g_mask", g_mask, None), ("g_depth", g_depth, None), ("g_rgb", g_rgb, 3)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != ((n,) if inner is None else (n, inner))):
                raise RuntimeError(f"ray_marching_composite backward: {name} must be a contiguous float32 tensor over the {n} rays")
        rows = self.rows
        base = self._pool.data_ptr()
        P = lambda name: (base + self._offs[name][0]) if self._offs[name][1] else None
        with H.on_device(dev):
            has_rgb = self._rgb is not None
            widths = [1, 1 if need_t else 0, 3 if (need_rgb and has_rgb) else 0, 1 if need_sigma else 0]
            self._gpool = pool = H.empty(max(rows, 1) * sum(widths), dtype=torch.float32, device=dev)
            gb, ptrs, off = pool.data_ptr(), [], 0
            for w in widths:
                ptrs.append(gb + 4 * off if w else None)
                off += rows * w
            self._gw = widths
            H.check(H.lib().nr3d_march_composite_bwd(
                n, P("packed_info"), P("alpha"), P("vw"), P("t_starts"), H.ptr(self._rgb), self.eps, self.thre,
                1 if self.normalize else 0, P("mask"), P("depth"), H.ptr(g_mask), H.ptr(g_depth), H.ptr(g_rgb), ptrs[0], ptrs[1], ptrs[2],
                H.ptr(self._sigma), P("deltas"), self.sigma_rows, ptrs[3], self._raw_stream))
        return self

    def grads(self):
        S, _ = self.totals()
        out, off = [], 0
        for i, w in enumerate(self._gw):
            if not w:
                out.append(None)
                continue
            t = self._gpool[off:off + S * w]
            out.append(t.view(S, 3) if w == 3 else t)
            off += self.rows * w
        return tuple(out)


def ray_marching_composite(rays_o, rays_d, t_min, t_max, roi, grid_binary, contraction_type, step_size, max_step_size, dt_gamma,
                           max_steps, sigma, rgb=None, early_stop_eps=1e-4, alpha_thre=0.0, normalize_depth=True, return_gidx=True):
    """ray_marching (ray_marching.cu:136-244) + the post-processing of occgrid_raymarch.py:87-112 + alpha = 1 - exp(-sigma * delta)
    (nerf_utils.py:23-24) + the renderer's alpha composite (renderer_mixin.py:298-311) as ONE library call that enqueues four launches
    and never waits for the device (nr3d_march_composite_fwd).  ``sigma`` [>= S] float32 (``rgb`` [same rows, 3], optional): the
    per-sample density / colour in march order -- e.g. the output of a field evaluated on a previous march of the same scene, or a
    constant medium.  Returns a ``MarchComposite``; nothing has been read back yet.
    Above FUSED_MARCH_COMPOSITE_MAX_BYTES of bound-sized buffers, for batched grids, or when there are no rays, use the two-phase
    chain (``ray_marching_finished`` -> ``tau_to_alpha_forward`` -> ``packed_composite_forward``): raises ValueError then."""
    _chk("rays_o", rays_o, 2, torch.float32); _chk("rays_d", rays_d, 2, torch.float32)
    _chk("t_min", t_min, 1, torch.float32); _chk("t_max", t_max, 1, torch.float32)
    _chk("roi", roi, 1, torch.float32); _chk("grid_binary", grid_binary, 3)
    _chk("sigma", sigma, 1, torch.float32)
    if grid_binary.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("grid_binary: expected a bool tensor")
    n = rays_o.shape[0]
    if tuple(rays_o.shape) != (n, 3) or tuple(rays_d.shape) != (n, 3) or t_min.shape[0] != n or t_max.shape[0] != n or roi.shape[0] != 6:
        raise RuntimeError("ray_marching_composite: rays_o / rays_d [n, 3], t_min / t_max [n], roi [6] expected")
    sigma_rows = sigma.shape[0]
    if rgb is not None:
        _chk("rgb", rgb, 2, torch.float32)
        if tuple(rgb.shape) != (sigma_rows, 3):
            raise RuntimeError("ray_marching_composite: rgb must be [sigma rows, 3]")
    dev = rays_o.device
    rows = n * int(max_steps)
    per_row = 4 * (5 + 3 + 2 + (1 if return_gidx else 0)) + 8 + 12       # t0 t1 ridx deltas + samples + alpha vw (+ gidx) + ridx64 + cache
    if n == 0 or rows * per_row > FUSED_MARCH_COMPOSITE_MAX_BYTES:
        raise ValueError("ray_marching_composite: outside the one-call range (no rays, or bound-sized buffers above "
                         "FUSED_MARCH_COMPOSITE_MAX_BYTES): use the two-phase chain")
    mc = MarchComposite(dev, n)
    mc.rows, mc.sigma_rows, mc.eps, mc.thre, mc.normalize = rows, sigma_rows, float(early_stop_eps), float(alpha_thre), bool(normalize_depth)
    res = (C.c_int32 * 3)(*[int(s) for s in grid_binary.shape])
    with H.on_device(dev):
        # ONE allocation, carved into 16-byte aligned pieces (sizes in bytes)
        scan_b = _scan_tmp_bytes(n)
        pieces = [("packed_info", n * 8), ("ridx_hit", n * 8), ("pack_infos", n * 16), ("scan_tmp", scan_b + 8), ("cache", rows * 12),
                  ("t_starts", rows * 4), ("t_ends", rows * 4), ("ridx32", rows * 4), ("gidx", rows * 4 if return_gidx else 0),
                  ("ridx", rows * 8), ("deltas", rows * 4), ("samples", rows * 12), ("alpha", rows * 4), ("vw", rows * 4),
                  ("mask", n * 4), ("depth", n * 4), ("rgb_out", n * 12 if rgb is not None else 0)]
        offs, tot = {}, 0
        for name, nb in pieces:
            offs[name] = (tot, nb)
            tot += (nb + 15) & ~15
        pool = H.empty(tot, dtype=torch.uint8, device=dev)

        base = pool.data_ptr()
        P = lambda name: (base + offs[name][0]) if offs[name][1] else None       # raw device addresses: no tensor views before the launch
        mc._pool, mc._offs = pool, offs
        mc._sigma, mc._rgb = sigma, rgb
        mc._host = _take_totals()
        mc._host.fill_(-1)                                   # the sentinel totals() polls against
        mc._raw_stream = H.stream_of(rays_o)
        H.check(H.lib().nr3d_march_composite_fwd(
            n, H.ptr(rays_o), H.ptr(rays_d), H.ptr(t_min), H.ptr(t_max), H.ptr(roi), res, H.ptr(grid_binary),
            int(contraction_type), float(step_size), float(max_step_size), float(dt_gamma), int(max_steps),
            P("packed_info"), P("ridx_hit"), P("pack_infos"), H.ptr(mc._host), P("scan_tmp"), P("cache"),
            rows * 12, rows, P("t_starts"), P("t_ends"), P("ridx32"), P("gidx"),
            P("ridx"), P("deltas"), P("samples"), H.ptr(sigma), sigma_rows, H.ptr(rgb),
            float(early_stop_eps), float(alpha_thre), 1 if normalize_depth else 0, P("alpha"), P("vw"),
            P("mask"), P("depth"), P("rgb_out"), mc._raw_stream))
    return mc
