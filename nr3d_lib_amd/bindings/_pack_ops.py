"""nr3d_lib_amd.bindings._pack_ops -- drop-in for the reference pybind module
``nr3d_lib.bindings._pack_ops`` (csrc/pack_ops/pack_ops.cpp:21-58, signatures pack_ops.h:11-65),
backed by libnr3d_hip.so.

Same function names, positional arguments and return structure.  What differs by design:
  * the reference validates ``feats.size(0) == pack_infos[-1].sum()`` with a device->host ``.item()``
    in EVERY op (25 sync sites, e.g. pack_ops_cuda.cu:839); here that check is off by default and
    enabled with ``nr3d_lib_amd.bindings._pack_ops.CHECK_PACK_SIZES = True`` (debug aid);
  * two-phase producers keep their prefix sums on the device and read back ONE scalar (the total);
  * ``packed_sort_thrust`` is an alias of the in-tree sort (no thrust on this platform).
"""
import ctypes as C

import torch

from .. import _hip as H

CHECK_PACK_SIZES = False
# packed_cumprod(exclusive=True): the reference kernel leaves the first element of each pack at 0, which
# zeroes the whole pack (pack_ops_cuda.cu:884-894 vs the docstring at pack_ops.py:149).  False replicates
# the reference; True gives the documented semantics (identity 1 first).
CUMPROD_EXCLUSIVE_DOCUMENTED = False

_SUPPORTED = (torch.float32, torch.float64, torch.int32, torch.int64)


def _chk_pi(fn, pack_infos, ref=None):
    if pack_infos.dim() != 2 or pack_infos.shape[-1] != 2:
        raise RuntimeError(f"{fn}: Expected pack_infos of shape [num_packs, 2]")
    if pack_infos.dtype != torch.int64:
        raise RuntimeError(f"{fn}: Expected pack_infos to have scalar type Long")
    if not pack_infos.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous tensor for argument pack_infos")
    H.require_gpu(pack_infos, ref)
    if ref is not None and ref.device != pack_infos.device:
        raise RuntimeError(f"{fn}: Expected all tensors on the same GPU")


def _chk_feats(fn, feats, pack_infos, dims=(1, 2)):
    if feats.dim() not in dims:
        raise RuntimeError(f"{fn}: Expected {' or '.join(str(d) for d in dims)}-dimensional tensor for argument feats")
    if not feats.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous tensor for argument feats")
    if feats.dtype not in _SUPPORTED:
        raise RuntimeError(f"{fn}: dtype {feats.dtype} not supported (float32/float64/int32/int64)")
    _chk_pi(fn, pack_infos, feats)
    if CHECK_PACK_SIZES and pack_infos.shape[0] > 0:
        want = int(pack_infos[-1, 0].item()) + int(pack_infos[-1, 1].item())
        if feats.shape[0] != want:
            raise RuntimeError(f"{fn}: Expected feats to have size {want} at dimension 0, but got size {feats.shape[0]}")


def _ordered(pack_infos):
    """1 when the kernel may zero the rows outside the packs itself (H.mark_ordered: pack_infos from one of this library's
    producers, unmodified since, at least one pack); 0 -> a zero-filled output, rows outside the packs untouched"""
    return 1 if (pack_infos.shape[0] > 0 and H.is_ordered(pack_infos)) else 0


def _fd(feats):
    return 1 if feats.dim() == 1 else int(feats.shape[1])


def _code(t):
    return C.c_int(H.DTYPE_CODE[t.dtype])


def _scan_tmp(n, device):
    nbytes = int(H.lib().nr3d_scan_tmp_bytes(C.c_uint64(max(int(n), 1))))
    return H.empty((nbytes + 7) // 8, dtype=torch.int64, device=device)


def _pack_infos_from_n(n_per_pack):
    """device-side exclusive scan -> (pack_infos int64 [P,2], total python int): ONE readback."""
    P = n_per_pack.shape[0]
    dev = n_per_pack.device
    pi = H.empty((P, 2), dtype=torch.int64, device=dev)
    total = H.host_i64(1, dev)                  # the scan's last store goes to pinned host memory: no copy launch
    tmp = _scan_tmp(P, dev)
    H.check(H.lib().nr3d_pack_infos_from_n(H.u32(P), H.ptr(n_per_pack), H.ptr(pi), H.ptr(total), H.ptr(tmp),
                                           H.stream_of(n_per_pack)))
    num = H.wait_i64(total, dev)[0]
    return H.mark_ordered(pi, total=num), num


# ------------------------------------------------------------------------------------------------
# interleave producers (pack_ops_cuda.cu:47-218)
# ------------------------------------------------------------------------------------------------
def interleave_arange(stop, return_idx):
    if stop.dim() != 1 or stop.dtype != torch.int64 or not stop.is_contiguous():
        raise RuntimeError("interleave_arange: Expected contiguous 1-D Long tensor for argument stop")
    H.require_gpu(stop)
    with H.on_device(stop.device):
        pi, num = _pack_infos_from_n(stop)
        out = H.empty(num, dtype=torch.int64, device=stop.device)
        nidx = H.empty(num, dtype=torch.int64, device=stop.device) if return_idx else None
        H.check(H.lib().nr3d_interleave_linstep(H.u32(stop.shape[0]), C.c_int(H.I64), H.ptr(pi), None, None,
                                                C.c_double(0), C.c_double(1), H.ptr(out), H.ptr(nidx),
                                                H.stream_of(stop)))
    return out, nidx


def interleave_linstep(start, num_steps, step_size, return_idx):
    if start.dim() != 1 or num_steps.dim() != 1 or start.shape != num_steps.shape:
        raise RuntimeError("interleave_linstep: Expected 1-D start / num_steps of the same size")
    if num_steps.dtype != torch.int64:
        raise RuntimeError("interleave_linstep: Expected num_steps to have scalar type Long")
    H.require_gpu(start, num_steps)
    if start.dtype not in _SUPPORTED:
        raise RuntimeError(f"interleave_linstep: dtype {start.dtype} not supported")
    start, num_steps = start.contiguous(), num_steps.contiguous()
    steps_t = None
    step_s = 0.0
    if isinstance(step_size, torch.Tensor):
        if step_size.dim() != 1 or step_size.dtype != start.dtype:
            raise RuntimeError("interleave_linstep: Expected 1-D step_size with the dtype of start")
        steps_t = step_size.contiguous()
    else:
        step_s = float(step_size)
    with H.on_device(start.device):
        pi, num = _pack_infos_from_n(num_steps)
        out = H.empty(num, dtype=start.dtype, device=start.device)
        nidx = H.empty(num, dtype=torch.int64, device=start.device) if return_idx else None
        H.check(H.lib().nr3d_interleave_linstep(H.u32(start.shape[0]), _code(start), H.ptr(pi), H.ptr(start),
                                                H.ptr(steps_t), C.c_double(0), C.c_double(step_s), H.ptr(out),
                                                H.ptr(nidx), H.stream_of(start)))
    return out, nidx


def arange_num_steps(start, stop, step_size):
    """ceil((stop - start) / step_size) as int64 [P] in one launch (the wrapper's stop.subtract(start).div(step).ceil().long())"""
    fn = "interleave_arange"
    if start.dim() != 1 or stop.shape != start.shape or stop.dtype != start.dtype or start.dtype not in _SUPPORTED:
        raise RuntimeError(f"{fn}: Expected 1-D start / stop of the same size and dtype (float32/float64/int32/int64)")
    H.require_gpu(start, stop)
    steps_t, step_s = None, 0.0
    if isinstance(step_size, torch.Tensor):
        if step_size.shape != start.shape or step_size.dtype != start.dtype:
            raise RuntimeError(f"{fn}: Expected a step_size tensor with the size and dtype of start")
        steps_t = step_size.contiguous()
    else:
        step_s = float(step_size)
    start, stop = start.contiguous(), stop.contiguous()
    with H.on_device(start.device):
        n = H.empty(start.shape[0], dtype=torch.int64, device=start.device)
        H.check(H.lib().nr3d_arange_num_steps(H.u32(start.shape[0]), _code(start), H.ptr(start), H.ptr(stop), H.ptr(steps_t),
                                              C.c_double(step_s), H.ptr(n), H.stream_of(start)))
    return n


def _chk_near_far(fn, near, far):
    if near.dim() != 1 or far.dim() != 1 or near.shape != far.shape:
        raise RuntimeError(f"{fn}: Expected 1-D near / far of the same size")
    if near.dtype != torch.float32 or far.dtype != torch.float32:
        raise RuntimeError(f"{fn}: float32 only on this platform")
    if not (near.is_contiguous() and far.is_contiguous()):
        raise RuntimeError(f"{fn}: Expected contiguous near / far")
    H.require_gpu(near, far)


def interleave_sample_step_wrt_depth_clamped(near, far, max_steps, dt_gamma, min_step_size, max_step_size):
    """-> (t_samples, deltas, nidx int64, pack_infos int64 [P,2])  (pack_ops_cuda.cu:480-604)"""
    _chk_near_far("interleave_sample_step_wrt_depth_clamped", near, far)
    P, dev = near.shape[0], near.device
    with H.on_device(dev):
        n = H.empty(P, dtype=torch.int64, device=dev)
        st = H.stream_of(near)
        H.check(H.lib().nr3d_sample_step_count(H.u32(P), H.ptr(near), H.ptr(far), H.u32(max_steps), H.f32(dt_gamma),
                                               H.f32(min_step_size), H.f32(max_step_size), H.ptr(n), st))
        pi, num = _pack_infos_from_n(n)
        t = H.empty(num, dtype=near.dtype, device=dev)
        dt = H.empty(num, dtype=near.dtype, device=dev)
        nidx = H.empty(num, dtype=torch.int64, device=dev)
        H.check(H.lib().nr3d_sample_step_emit(H.u32(P), H.ptr(near), H.ptr(pi), H.f32(dt_gamma), H.f32(min_step_size),
                                              H.f32(max_step_size), H.ptr(t), H.ptr(dt), H.ptr(nidx), st))
    return t, dt, nidx, pi


# the deprecated v1 sampler produces the same samples (pack_ops_cuda.cu:226-366)
interleave_sample_step_wrt_depth_clamp_deprecated = interleave_sample_step_wrt_depth_clamped


def interleave_sample_step_wrt_depth_in_packed_segments(near, far, entry, exit, seg_pack_infos, max_steps, dt_gamma,
                                                        min_step_size, max_step_size):
    """-> (t_samples, deltas, sidx, nidx, pack_infos)  (pack_ops_cuda.cu:606-795)"""
    fn = "interleave_sample_step_wrt_depth_in_packed_segments"
    _chk_near_far(fn, near, far)
    _chk_pi(fn, seg_pack_infos, near)
    if entry.dim() != 1 or exit.dim() != 1 or entry.shape != exit.shape or entry.dtype != near.dtype \
            or exit.dtype != near.dtype or not entry.is_contiguous() or not exit.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous 1-D entry / exit of the same size and dtype as near")
    if seg_pack_infos.shape[0] != near.shape[0]:
        raise RuntimeError(f"{fn}: Expected seg_pack_infos of size [{near.shape[0]}, 2]")
    P, dev = near.shape[0], near.device
    with H.on_device(dev):
        n = H.empty(P, dtype=torch.int64, device=dev)
        st = H.stream_of(near)
        common = (H.u32(P), H.ptr(near), H.ptr(far), H.ptr(entry), H.ptr(exit), H.ptr(seg_pack_infos),
                  H.u32(max_steps), H.f32(dt_gamma), H.f32(min_step_size), H.f32(max_step_size))
        H.check(H.lib().nr3d_sample_step_segments(*common, C.c_int(0), H.ptr(n), None, None, None, None, None, st))
        pi, num = _pack_infos_from_n(n)
        t = H.empty(num, dtype=near.dtype, device=dev)
        dt = H.empty(num, dtype=near.dtype, device=dev)
        nidx = H.empty(num, dtype=torch.int64, device=dev)
        sidx = H.empty(num, dtype=torch.int64, device=dev)
        H.check(H.lib().nr3d_sample_step_segments(*common, C.c_int(1), None, H.ptr(pi), H.ptr(t), H.ptr(dt),
                                                  H.ptr(nidx), H.ptr(sidx), st))
    return t, dt, sidx, nidx, pi


# ------------------------------------------------------------------------------------------------
# reductions / scans / differences
# ------------------------------------------------------------------------------------------------
def packed_sum(feats, pack_infos):
    _chk_feats("packed_sum", feats, pack_infos)
    P = pack_infos.shape[0]
    with H.on_device(feats.device):
        # every pack's row is written by the kernel (an empty pack gets its zero there): no zero-fill launch in front of it
        out = H.empty((P,) + tuple(feats.shape[1:]), dtype=feats.dtype, device=feats.device)
        H.check(H.lib().nr3d_packed_sum(H.u32(P), C.c_uint64(feats.shape[0]), H.u32(_fd(feats)), _code(feats),
                                        H.ptr(feats), H.ptr(pack_infos), H.ptr(out), H.stream_of(feats)))
    return out


def _scan(fn, feats, pack_infos, mode, exclusive, reverse):
    _chk_feats(fn, feats, pack_infos)
    ordered = _ordered(pack_infos)
    with H.on_device(feats.device):
        out = H.empty_like(feats) if ordered else torch.zeros_like(feats)
        H.check(H.lib().nr3d_packed_scan(H.u32(pack_infos.shape[0]), C.c_uint64(feats.shape[0]), H.u32(_fd(feats)),
                                         _code(feats), H.ptr(feats), H.ptr(pack_infos), C.c_int(mode),
                                         C.c_int(int(bool(exclusive))), C.c_int(int(bool(reverse))), C.c_int(ordered),
                                         H.ptr(out), H.stream_of(feats)))
    return out


def packed_cumsum(feats, pack_infos, exclusive, reverse):
    return _scan("packed_cumsum", feats, pack_infos, 0, exclusive, reverse)


def packed_cumprod(feats, pack_infos, exclusive, reverse):
    return _scan("packed_cumprod", feats, pack_infos, 2 if CUMPROD_EXCLUSIVE_DOCUMENTED else 1, exclusive, reverse)


def _edge(fn, name, e, feats, P):
    if e is None:
        return None
    want = (P,) if feats.dim() == 1 else (P, feats.shape[1])
    if tuple(e.shape) != want or e.dtype != feats.dtype or not e.is_contiguous() or e.device != feats.device:
        raise RuntimeError(f"{fn}: Expected contiguous {name} of size {list(want)} with the dtype/device of feats")
    return e


def _diff(fn, feats, pack_infos, edge_a, edge_fill, backward, names):
    _chk_feats(fn, feats, pack_infos)
    if edge_a is not None and edge_fill is not None:
        raise RuntimeError("You should only specify AT MOST one of [appends, prepends, last_fill, first_fill]")
    P = pack_infos.shape[0]
    edge_a, edge_fill = _edge(fn, names[0], edge_a, feats, P), _edge(fn, names[1], edge_fill, feats, P)
    ordered = _ordered(pack_infos)
    with H.on_device(feats.device):
        out = H.empty_like(feats) if ordered else torch.zeros_like(feats)
        H.check(H.lib().nr3d_packed_diff(H.u32(P), C.c_uint64(feats.shape[0]), H.u32(_fd(feats)), _code(feats),
                                         H.ptr(feats), H.ptr(pack_infos), H.ptr(edge_a), H.ptr(edge_fill),
                                         C.c_int(backward), C.c_int(ordered), H.ptr(out), H.stream_of(feats)))
    return out


def packed_diff(feats, pack_infos, pack_appends_=None, pack_last_fill_=None):
    return _diff("packed_diff", feats, pack_infos, pack_appends_, pack_last_fill_, 0, ("pack_appends", "pack_last_fill"))


def packed_backward_diff(feats, pack_infos, pack_prepends_=None, pack_first_fill_=None):
    return _diff("packed_backward_diff", feats, pack_infos, pack_prepends_, pack_first_fill_, 1,
                 ("pack_prepends", "pack_first_fill"))


# ------------------------------------------------------------------------------------------------
# per-pack broadcast arithmetic / logic (pack_ops.h:25-37 op codes)
# ------------------------------------------------------------------------------------------------
_OPS = dict(add=0, sub=1, mul=2, div=3, matmul=4, gt=5, geq=6, lt=7, leq=8, eq=9, neq=10)


def _binary(name, feats, other, pack_infos):
    fn = f"packed_{name}"
    op = _OPS[name]
    _chk_feats(fn, feats, pack_infos, dims=(2,) if op == 4 else (1, 2))
    H.require_gpu(other)
    if not other.is_contiguous() or other.dtype != feats.dtype or other.device != feats.device:
        raise RuntimeError(f"{fn}: Expected contiguous `other` with the dtype/device of feats")
    P = pack_infos.shape[0]
    if other.shape[0] != P:
        raise RuntimeError(f"{fn}: Expected other to have size {P} at dimension 0, but got size {other.shape[0]}")
    fd = _fd(feats)
    if op == 4:
        if other.dim() != 3 or other.shape[2] != fd:
            raise RuntimeError(f"{fn}: Expected other of size [num_packs, out_feat_dim, {fd}]")
        od = int(other.shape[1])
        out_shape, out_dtype = (feats.shape[0], od), feats.dtype
    else:
        if other.dim() != feats.dim() or (feats.dim() == 2 and other.shape[1] != fd):
            raise RuntimeError(f"{fn}: Expected feats and other to have the same number of dimensions / feature width")
        od = fd
        out_shape, out_dtype = tuple(feats.shape), (torch.bool if op >= 5 else feats.dtype)
    ordered = _ordered(pack_infos)
    with H.on_device(feats.device):
        out = (H.empty if ordered else torch.zeros)(out_shape, dtype=out_dtype, device=feats.device)
        H.check(H.lib().nr3d_packed_binary(H.u32(P), C.c_uint64(feats.shape[0]), H.u32(fd), H.u32(od), _code(feats),
                                           H.ptr(feats), H.ptr(other), H.ptr(pack_infos), C.c_int(op), C.c_int(ordered),
                                           H.ptr(out), H.stream_of(feats)))
    return out


def packed_add(feats, other, pack_infos): return _binary("add", feats, other, pack_infos)
def packed_sub(feats, other, pack_infos): return _binary("sub", feats, other, pack_infos)
def packed_mul(feats, other, pack_infos): return _binary("mul", feats, other, pack_infos)
def packed_div(feats, other, pack_infos): return _binary("div", feats, other, pack_infos)
def packed_matmul(feats, other, pack_infos): return _binary("matmul", feats, other, pack_infos)
def packed_gt(feats, other, pack_infos): return _binary("gt", feats, other, pack_infos)
def packed_geq(feats, other, pack_infos): return _binary("geq", feats, other, pack_infos)
def packed_lt(feats, other, pack_infos): return _binary("lt", feats, other, pack_infos)
def packed_leq(feats, other, pack_infos): return _binary("leq", feats, other, pack_infos)
def packed_eq(feats, other, pack_infos): return _binary("eq", feats, other, pack_infos)
def packed_neq(feats, other, pack_infos): return _binary("neq", feats, other, pack_infos)


# ------------------------------------------------------------------------------------------------
# sort / search / merge / inverse CDF
# ------------------------------------------------------------------------------------------------
def packed_sort_qsort(vals, pack_infos, return_idx):
    """Sorts ``vals`` IN PLACE per pack (ascending); returns the index permutation or None
    (pack_ops_cuda.cu:2634-2763)."""
    _chk_feats("packed_sort_qsort", vals, pack_infos, dims=(1,))
    with H.on_device(vals.device):
        idx = torch.arange(vals.shape[0], dtype=torch.int64, device=vals.device) if return_idx else None
        H.check(H.lib().nr3d_packed_sort(H.u32(pack_infos.shape[0]), C.c_uint64(vals.shape[0]), _code(vals),
                                         H.ptr(vals), H.ptr(idx), H.ptr(pack_infos), H.stream_of(vals)))
    return idx


packed_sort_thrust = packed_sort_qsort


def packed_searchsorted(bins, vals, pack_infos):
    _chk_feats("packed_searchsorted", bins, pack_infos, dims=(1,))
    H.require_gpu(vals)
    if vals.dim() != 2 or vals.dtype != bins.dtype or not vals.is_contiguous() or vals.shape[0] != pack_infos.shape[0]:
        raise RuntimeError("packed_searchsorted: Expected contiguous vals of size [num_packs, num_to_search] "
                           "with the dtype of bins")
    with H.on_device(bins.device):
        pidx = torch.full(vals.shape, -1, dtype=torch.int64, device=bins.device)
        H.check(H.lib().nr3d_packed_searchsorted(H.u32(pack_infos.shape[0]), _code(bins), H.ptr(bins), H.ptr(vals),
                                                 H.ptr(pack_infos), H.u32(vals.shape[1]), None, H.ptr(pidx),
                                                 H.stream_of(bins)))
    return pidx


def packed_searchsorted_packed_vals(bins, pack_infos, vals, val_pack_infos):
    _chk_feats("packed_searchsorted_packed_vals", bins, pack_infos, dims=(1,))
    _chk_feats("packed_searchsorted_packed_vals", vals, val_pack_infos, dims=(1,))
    if vals.dtype != bins.dtype or val_pack_infos.shape[0] != pack_infos.shape[0]:
        raise RuntimeError("packed_searchsorted_packed_vals: vals must have the dtype of bins and one pack per bin pack")
    with H.on_device(bins.device):
        pidx = torch.full(vals.shape, -1, dtype=torch.int64, device=bins.device)
        H.check(H.lib().nr3d_packed_searchsorted(H.u32(pack_infos.shape[0]), _code(bins), H.ptr(bins), H.ptr(vals),
                                                 H.ptr(pack_infos), H.u32(0), H.ptr(val_pack_infos), H.ptr(pidx),
                                                 H.stream_of(bins)))
    return pidx


def try_merge_two_packs_sorted_aligned(vals_a, pack_infos_a, vals_b, pack_infos_b, b_sorted):
    """-> (pidx_a, pidx_b, pack_infos_merged)  (pack_ops_cuda.cu:1505-1631)"""
    fn = "try_merge_two_packs_sorted_aligned"
    _chk_feats(fn, vals_a, pack_infos_a, dims=(1,))
    _chk_feats(fn, vals_b, pack_infos_b, dims=(1,))
    if vals_a.dtype != vals_b.dtype or pack_infos_a.shape != pack_infos_b.shape:
        raise RuntimeError(f"{fn}: the two packs must be aligned and share a dtype")
    dev, P = vals_a.device, pack_infos_a.shape[0]
    with H.on_device(dev):
        # merged lengths -> merged pack_infos through the device scan (its total is len(a) + len(b): not read back)
        n = pack_infos_a[:, 1] + pack_infos_b[:, 1]
        pim = H.empty((P, 2), dtype=torch.int64, device=dev)
        if P > 0:
            tot = H.empty(1, dtype=torch.int64, device=dev)
            H.check(H.lib().nr3d_pack_infos_from_n(H.u32(P), H.ptr(n), H.ptr(pim), H.ptr(tot), H.ptr(_scan_tmp(P, dev)),
                                                   H.stream_of(vals_a)))
            H.mark_ordered(pim)
        pa = torch.zeros(vals_a.shape[0], dtype=torch.int64, device=dev)      # the merge kernel counts into it
        pb = torch.zeros(vals_b.shape[0], dtype=torch.int64, device=dev)
        H.check(H.lib().nr3d_try_merge_two_packs_sorted_aligned(
            H.u32(pack_infos_a.shape[0]), _code(vals_a), H.ptr(vals_a), H.ptr(pack_infos_a), H.ptr(vals_b),
            H.ptr(pack_infos_b), H.ptr(pim), C.c_int(int(bool(b_sorted))), H.ptr(pa), H.ptr(pb), H.stream_of(vals_a)))
    return pa, pb, pim


def merge_two_packs_sorted_general(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b):
    """merge_two_packs_sorted for arbitrary (sorted, unique) pack-id lists -> (pidx_a, pidx_b, pack_infos of the union): the
    reference's torch.unique / nonzero / index chain (graphics/pack_ops/pack_ops.py:611-640, ~40 launches and six syncs) as the
    union of the id lists in aligned form (nr3d_merge_pack_union: three launches), ONE readback (the number of union packs), the
    scan of the merged lengths and the aligned merge kernel over packs that exist in a, in b or in both"""
    fn = "merge_two_packs_sorted"
    _chk_feats(fn, vals_a, pack_infos_a, dims=(1,))
    _chk_feats(fn, vals_b, pack_infos_b, dims=(1,))
    H.require_gpu(nidx_a, nidx_b)
    Pa, Pb = pack_infos_a.shape[0], pack_infos_b.shape[0]
    if vals_a.dtype != vals_b.dtype:
        raise RuntimeError(f"{fn}: the two packs must share a dtype")
    for nm, t, P in (("nidx_a", nidx_a, Pa), ("nidx_b", nidx_b, Pb)):
        if t.dtype != torch.int64 or tuple(t.shape) != (P,) or not t.is_contiguous():
            raise RuntimeError(f"{fn}: Expected a contiguous int64 {nm} of shape [{P}]")
    if Pa == 0 or Pb == 0:
        raise RuntimeError(f"{fn}: both pack lists must be non-empty")
    dev = vals_a.device
    with H.on_device(dev):
        st = H.stream_of(vals_a)
        scratch = H.empty(3 * Pb + 1, dtype=torch.int64, device=dev)          # only_b [Pb] | ob [Pb, 2] | n_only [1]
        cap = Pa + Pb
        out = H.empty(6 * cap, dtype=torch.int64, device=dev)                 # u | pia_u | pib_u | n_u
        u, pia_u, pib_u, n_u = out[:cap], out[cap:3 * cap].view(cap, 2), out[3 * cap:5 * cap].view(cap, 2), out[5 * cap:]
        total = H.host_i64(1, dev)
        H.check(H.lib().nr3d_merge_pack_union(H.u32(Pa), H.ptr(nidx_a), H.ptr(pack_infos_a), H.u32(Pb), H.ptr(nidx_b),
                                              H.ptr(pack_infos_b), H.ptr(scratch), H.ptr(scratch[Pb:]), H.ptr(_scan_tmp(Pb, dev)),
                                              H.ptr(scratch[3 * Pb:]), H.ptr(u), H.ptr(pia_u), H.ptr(pib_u), H.ptr(n_u), H.ptr(total), st))
        Pu = H.wait_i64(total, dev)[0]                                        # the one device->host sync
        pim = H.empty((Pu, 2), dtype=torch.int64, device=dev)
        tot = H.empty(1, dtype=torch.int64, device=dev)                       # = len(vals_a) + len(vals_b): not read back
        H.check(H.lib().nr3d_pack_infos_from_n(H.u32(Pu), H.ptr(n_u), H.ptr(pim), H.ptr(tot), H.ptr(_scan_tmp(Pu, dev)), st))
        # elements outside every pack have no position: -1, as in the reference (the kernel zeroes the rows of its packs itself
        # before it counts into them)
        pa = torch.full((vals_a.shape[0],), -1, dtype=torch.int64, device=dev)
        pb = torch.full((vals_b.shape[0],), -1, dtype=torch.int64, device=dev)
        H.check(H.lib().nr3d_try_merge_two_packs_sorted_aligned(
            H.u32(Pu), _code(vals_a), H.ptr(vals_a), H.ptr(pia_u), H.ptr(vals_b), H.ptr(pib_u), H.ptr(pim), C.c_int(1),
            H.ptr(pa), H.ptr(pb), st))
    return pa, pb, H.mark_ordered(pim, total=vals_a.shape[0] + vals_b.shape[0])


def packed_invert_cdf(bins, cdfs, u, pack_infos):
    fn = "packed_invert_cdf"
    _chk_feats(fn, bins, pack_infos, dims=(1,))
    H.require_gpu(cdfs, u)
    if bins.dtype != torch.float32:
        raise RuntimeError(f"{fn}: float32 only on this platform")
    if cdfs.shape != bins.shape or cdfs.dtype != bins.dtype or not cdfs.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous cdfs with the size and dtype of bins")
    if u.dim() != 2 or u.dtype != bins.dtype or not u.is_contiguous() or u.shape[0] != pack_infos.shape[0]:
        raise RuntimeError(f"{fn}: Expected contiguous u of size [num_packs, num_to_sample]")
    with H.on_device(bins.device):
        bin_idx = torch.full(u.shape, -1, dtype=torch.int64, device=bins.device)
        samples = torch.zeros_like(u)
        H.check(H.lib().nr3d_packed_invert_cdf(H.u32(pack_infos.shape[0]), H.ptr(bins), H.ptr(cdfs), H.ptr(pack_infos),
                                               H.ptr(u), H.u32(u.shape[1]), H.ptr(samples), H.ptr(bin_idx),
                                               H.stream_of(bins)))
    return samples, bin_idx


# ------------------------------------------------------------------------------------------------
# alpha -> volume-rendering weights (pack_ops_cuda.cu:1735-1958)
# ------------------------------------------------------------------------------------------------
def packed_alpha_to_vw_forward(alphas, pack_infos, early_stop_eps, alpha_thre, compression):
    """-> (weights | None, compact_pack_infos int64 [P,2] | None, compact_selector bool [S] | None)"""
    fn = "packed_alpha_to_vw_forward"
    _chk_feats(fn, alphas, pack_infos, dims=(1,))
    if alphas.dtype != torch.float32:
        raise RuntimeError(f"{fn}: float32 only on this platform")
    P, S, dev = pack_infos.shape[0], alphas.shape[0], alphas.device
    with H.on_device(dev):
        st = H.stream_of(alphas)
        if compression:
            num = torch.zeros(P, dtype=torch.int64, device=dev)
            # the kernel writes the rows of the packs; rows outside every pack are zero (at::zeros in the reference): a fill launch
            # unless the packs are known to tile [0, S) (H.tiles: a marcher's / a two-phase op's pack_infos)
            sel = (H.empty if H.tiles(pack_infos, S) else torch.zeros)(S, dtype=torch.bool, device=dev)
            H.check(H.lib().nr3d_alpha_to_vw_forward(H.u32(P), C.c_uint64(S), H.ptr(alphas), H.ptr(pack_infos),
                                                     H.f32(early_stop_eps), H.f32(alpha_thre), None, H.ptr(num),
                                                     H.ptr(sel), st))
            cpi = H.empty((P, 2), dtype=torch.int64, device=dev)
            total = H.empty(1, dtype=torch.int64, device=dev)
            H.check(H.lib().nr3d_pack_infos_from_n(H.u32(P), H.ptr(num), H.ptr(cpi), H.ptr(total),
                                                   H.ptr(_scan_tmp(P, dev)), st))
            return None, H.mark_ordered(cpi), sel
        w = (H.empty if (H.tiles(pack_infos, S) and P > 0) else torch.zeros)(S, dtype=alphas.dtype, device=dev)
        H.check(H.lib().nr3d_alpha_to_vw_forward(H.u32(P), C.c_uint64(S), H.ptr(alphas), H.ptr(pack_infos),
                                                 H.f32(early_stop_eps), H.f32(alpha_thre), H.ptr(w), None, None, st))
    return w, None, None


def packed_compression_compact(alphas, pack_infos, early_stop_eps, alpha_thre, tag=None, f1=None, f2=None, f3=None,
                               l1=None, want_pidx=True):
    """Visibility pruning in four launches and ONE readback: the compaction-mode alpha -> weights kernel (selector + kept
    samples per pack), one scan that yields every pack's new begin AND the packs that keep >= 1 sample, one pass that moves
    the kept samples -- replaces packed_alpha_to_vw_forward(compression) + 2 nonzero() + the index gathers of the caller
    (nr3d_lib/graphics/nerf/nerf_utils.py:64-98, nerf_ray_query.py:128-137).
    tag int64 [P] (e.g. the packs' ray indices) | None; per-sample arrays to carry along: f1, f2 float [S], f3 float [S, 3],
    l1 int64 [S].  -> (idx of the useful packs (tag[idx] when tag is given) int64 [P'], pack_infos int64 [P', 2],
    pidx int64 [S'] | None, f1', f2', f3', l1')"""
    fn = "packed_compression_compact"
    _chk_feats(fn, alphas, pack_infos, dims=(1,))
    if alphas.dtype != torch.float32:
        raise RuntimeError(f"{fn}: float32 only on this platform")
    P, S, dev = pack_infos.shape[0], alphas.shape[0], alphas.device
    H.require_gpu(tag, f1, f2, f3, l1)
    for name, t, shape, dt in (("tag", tag, (P,), torch.int64), ("f1", f1, (S,), torch.float32), ("f2", f2, (S,), torch.float32),
                               ("f3", f3, (S, 3), torch.float32), ("l1", l1, (S,), torch.int64)):
        if t is not None and (t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous()):
            raise RuntimeError(f"{fn}: Expected a contiguous {dt} {name} of shape {list(shape)}, got {t.dtype} {list(t.shape)}")
    with H.on_device(dev):
        st = H.stream_of(alphas)
        num = torch.zeros(P, dtype=torch.int64, device=dev)
        sel = H.empty(S, dtype=torch.bool, device=dev)
        H.check(H.lib().nr3d_alpha_to_vw_forward(H.u32(P), C.c_uint64(S), H.ptr(alphas), H.ptr(pack_infos), H.f32(early_stop_eps),
                                                 H.f32(alpha_thre), None, H.ptr(num), H.ptr(sel), st))
        begin_all = H.empty(P, dtype=torch.int64, device=dev)
        idx = H.empty(P, dtype=torch.int64, device=dev)
        cpi = H.empty((P, 2), dtype=torch.int64, device=dev)
        totals = H.host_i64(2, dev)
        H.check(H.lib().nr3d_prune_compact_packs(H.u32(P), H.ptr(num), H.ptr(tag), H.ptr(begin_all), H.ptr(idx), H.ptr(cpi),
                                                 H.ptr(totals), H.ptr(_scan_tmp(P, dev)), st))
        S2, P2 = H.wait_i64(totals, dev)                    # the one device->host sync
        pidx = H.empty(S2, dtype=torch.int64, device=dev) if want_pidx else None
        o1 = H.empty(S2, dtype=torch.float32, device=dev) if f1 is not None else None
        o2 = H.empty(S2, dtype=torch.float32, device=dev) if f2 is not None else None
        o3 = H.empty((S2, 3), dtype=torch.float32, device=dev) if f3 is not None else None
        ol = H.empty(S2, dtype=torch.int64, device=dev) if l1 is not None else None
        if S2 > 0:
            H.check(H.lib().nr3d_prune_compact_samples(H.u32(P), H.ptr(pack_infos), H.ptr(begin_all), H.ptr(sel), H.ptr(f1),
                                                       H.ptr(f2), H.ptr(f3), H.ptr(l1), H.ptr(pidx), H.ptr(o1), H.ptr(o2),
                                                       H.ptr(o3), H.ptr(ol), st))
    return idx[:P2], H.mark_ordered(cpi[:P2], total=S2), pidx, o1, o2, o3, ol


def tau_to_alpha_forward(sigma, delta):
    """alpha = 1 - exp(-sigma * delta), one launch (nerf_utils.py:23-24 on sigma * deltas)"""
    fn = "tau_to_alpha_forward"
    H.require_gpu(sigma, delta)
    if sigma.dtype != torch.float32 or delta.dtype != torch.float32 or sigma.shape != delta.shape \
            or not sigma.is_contiguous() or not delta.is_contiguous():
        raise RuntimeError(f"{fn}: Expected contiguous float32 sigma / delta of the same shape")
    with H.on_device(sigma.device):
        alpha = H.empty_like(sigma)
        H.check(H.lib().nr3d_tau_to_alpha_fwd(C.c_uint64(sigma.numel()), H.ptr(sigma), H.ptr(delta), H.ptr(alpha),
                                              H.stream_of(sigma)))
    return alpha


def tau_to_alpha_backward(sigma, delta, grad_alpha):
    """grad_sigma = grad_alpha * delta * exp(-sigma * delta), one launch"""
    fn = "tau_to_alpha_backward"
    H.require_gpu(sigma, delta, grad_alpha)
    for t in (delta, grad_alpha):
        if t.dtype != torch.float32 or t.shape != sigma.shape or not t.is_contiguous():
            raise RuntimeError(f"{fn}: Expected contiguous float32 tensors of sigma's shape")
    with H.on_device(sigma.device):
        g = H.empty_like(sigma)
        H.check(H.lib().nr3d_tau_to_alpha_bwd(C.c_uint64(sigma.numel()), H.ptr(sigma), H.ptr(delta), H.ptr(grad_alpha), H.ptr(g),
                                              H.stream_of(sigma)))
    return g


def packed_alpha_to_vw_backward(weights, grad_weights, alphas, pack_infos, early_stop_eps, alpha_thre):
    fn = "packed_alpha_to_vw_backward"
    _chk_feats(fn, weights, pack_infos, dims=(1,))
    H.require_gpu(grad_weights, alphas)
    for t in (grad_weights, alphas):
        if t.shape != weights.shape or t.dtype != weights.dtype or not t.is_contiguous():
            raise RuntimeError(f"{fn}: Expected contiguous weights / grad_weights / alphas of the same size and dtype")
    if weights.dtype != torch.float32:
        raise RuntimeError(f"{fn}: float32 only on this platform")
    with H.on_device(weights.device):
        tiled = H.tiles(pack_infos, weights.shape[0]) and pack_infos.shape[0] > 0
        g = H.empty_like(alphas) if tiled else torch.zeros_like(alphas)
        H.check(H.lib().nr3d_alpha_to_vw_backward(H.u32(pack_infos.shape[0]), C.c_uint64(weights.shape[0]),
                                                  H.ptr(alphas), H.ptr(weights), H.ptr(grad_weights), H.ptr(pack_infos),
                                                  H.f32(early_stop_eps), H.f32(alpha_thre), H.ptr(g),
                                                  H.stream_of(weights)))
    return g


# ------------------------------------------------------------------------------------------------
# fused alpha composite (no single reference binding: the renderer's op chain renderer_mixin.py:298-311 in one launch
# each way, nr3d_pack_composite_fwd / _bwd)
# ------------------------------------------------------------------------------------------------
def _f32_1d(fn, name, t, n, inner=None):
    shape = (n,) if inner is None else (n, inner)
    if t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous():
        raise RuntimeError(f"{fn}: Expected a contiguous float32 {name} of shape {list(shape)}, got {t.dtype} {list(t.shape)}")


def packed_composite_forward(alphas, t, rgb, pack_infos, rays_inds_hit, num_rays, early_stop_eps, alpha_thre,
                             normalize_depth, packs_tile=False):
    """alphas, t [S]; rgb [S,3] | None; pack_infos int64 [P,2]; rays_inds_hit int64 [P] | None (then num_rays == P).
    -> (vw [S], mask [num_rays], depth [num_rays], rgb_out [num_rays,3] | None); rays that are not hit keep zeros.
    The kernels write the samples that belong to a pack; samples outside every pack read zero like the reference ops'
    at::zeros outputs -- unless the caller states ``packs_tile`` (the packs cover [0, S) exactly, as a marcher's or a
    compaction's do), which saves the zero-fill."""
    fn = "packed_composite_forward"
    _chk_feats(fn, alphas, pack_infos, dims=(1,))
    P, S, dev = pack_infos.shape[0], alphas.shape[0], alphas.device
    H.require_gpu(t, rgb, rays_inds_hit)
    _f32_1d(fn, "alphas", alphas, S); _f32_1d(fn, "t", t, S)
    if rgb is not None:
        _f32_1d(fn, "rgb", rgb, S, 3)
    if rays_inds_hit is not None:
        if rays_inds_hit.dtype != torch.int64 or tuple(rays_inds_hit.shape) != (P,) or not rays_inds_hit.is_contiguous():
            raise RuntimeError(f"{fn}: Expected a contiguous int64 rays_inds_hit of shape [{P}]")
    elif int(num_rays) != P:
        raise RuntimeError(f"{fn}: num_rays must equal the number of packs when rays_inds_hit is None")
    with H.on_device(dev):
        vw = (H.empty if (packs_tile and P > 0) else torch.zeros)(S, dtype=torch.float32, device=dev)
        # per-ray outputs: [mask | depth | rgb] views of one buffer (rays that are not hit keep zeros: one fill, not three)
        nr = int(num_rays) if rays_inds_hit is not None else P
        pool = (torch.zeros if rays_inds_hit is not None else H.empty)(nr * (5 if rgb is not None else 2), dtype=torch.float32, device=dev)
        mask, depth = pool[:nr], pool[nr:2 * nr]
        rgb_out = pool[2 * nr:].view(nr, 3) if rgb is not None else None
        H.check(H.lib().nr3d_pack_composite_fwd(H.u32(P), H.ptr(alphas), H.ptr(t), H.ptr(rgb), H.ptr(pack_infos),
                                                H.ptr(rays_inds_hit), H.f32(early_stop_eps), H.f32(alpha_thre),
                                                C.c_int(1 if normalize_depth else 0), H.ptr(vw), H.ptr(mask), H.ptr(depth),
                                                H.ptr(rgb_out), H.stream_of(alphas)))
    return vw, mask, depth, rgb_out


def packed_composite_backward(alphas, vw, t, rgb, pack_infos, rays_inds_hit, early_stop_eps, alpha_thre, normalize_depth,
                              mask, depth, g_mask, g_depth, g_rgb, g_vw, need_t=True, need_rgb=True, packs_tile=False):
    """-> (grad_alphas [S], grad_t [S] | None, grad_rgb [S,3] | None); g_* may be None (zero); ``packs_tile`` as in the
    forward (gradients of samples outside every pack are zero otherwise)"""
    fn = "packed_composite_backward"
    _chk_feats(fn, alphas, pack_infos, dims=(1,))
    P, S, dev = pack_infos.shape[0], alphas.shape[0], alphas.device
    H.require_gpu(vw, t, rgb, rays_inds_hit, mask, depth, g_mask, g_depth, g_rgb, g_vw)
    n_out = mask.shape[0]
    for name, tt, inner in (("vw", vw, None), ("t", t, None), ("g_vw", g_vw, None)):
        if tt is not None:
            _f32_1d(fn, name, tt, S, inner)
    for name, tt, inner in (("mask", mask, None), ("depth", depth, None), ("g_mask", g_mask, None), ("g_depth", g_depth, None),
                            ("g_rgb", g_rgb, 3)):
        if tt is not None:
            _f32_1d(fn, name, tt, n_out, inner)
    with H.on_device(dev):
        alloc = H.empty if (packs_tile and P > 0) else torch.zeros
        ga = alloc(S, dtype=torch.float32, device=dev)
        gt = alloc(S, dtype=torch.float32, device=dev) if need_t else None
        gr = alloc((S, 3), dtype=torch.float32, device=dev) if (need_rgb and rgb is not None) else None
        H.check(H.lib().nr3d_pack_composite_bwd(H.u32(P), H.ptr(alphas), H.ptr(vw), H.ptr(t), H.ptr(rgb), H.ptr(pack_infos),
                                                H.ptr(rays_inds_hit), H.f32(early_stop_eps), H.f32(alpha_thre),
                                                C.c_int(1 if normalize_depth else 0), H.ptr(mask), H.ptr(depth), H.ptr(g_mask),
                                                H.ptr(g_depth), H.ptr(g_rgb), H.ptr(g_vw), H.ptr(ga), H.ptr(gt), H.ptr(gr),
                                                H.stream_of(alphas)))
    return ga, gt, gr


# ------------------------------------------------------------------------------------------------
# misc
# ------------------------------------------------------------------------------------------------
def mark_pack_boundaries_cuda(pack_ids):
    if pack_ids.dim() != 1 or not pack_ids.is_contiguous():
        raise RuntimeError("mark_pack_boundaries_cuda: Expected contiguous 1-D tensor for argument pack_ids")
    if pack_ids.dtype not in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
        raise RuntimeError("mark_pack_boundaries_cuda: Expected pack_ids to have one of scalar types Byte, Char, Int, "
                           "Long, Short")
    H.require_gpu(pack_ids)
    with H.on_device(pack_ids.device):
        b = H.empty(pack_ids.shape[0], dtype=torch.int32, device=pack_ids.device)
        H.check(H.lib().nr3d_mark_pack_boundaries(C.c_uint64(pack_ids.shape[0]), _code(pack_ids), H.ptr(pack_ids),
                                                  H.ptr(b), H.stream_of(pack_ids)))
    return b


def octree_mark_consecutive_segments(pidx, pack_infos, point_hierarchies):
    """-> (mark_start, mark_end) bool [n]  (pack_ops_cuda.cu:2843-2887)"""
    fn = "octree_mark_consecutive_segments"
    _chk_pi(fn, pack_infos, pidx)
    for name, t, dim, dt in (("pidx", pidx, 1, torch.int32), ("point_hierarchies", point_hierarchies, 2, torch.int16)):
        if t.dim() != dim or t.dtype != dt or not t.is_contiguous() or t.device != pack_infos.device:
            raise RuntimeError(f"{fn}: Expected a contiguous {dim}-dimensional {dt} tensor on the same GPU for argument {name}")
    n = pidx.shape[0]
    mark_start = torch.zeros(n, dtype=torch.bool, device=pidx.device)
    mark_end = torch.zeros(n, dtype=torch.bool, device=pidx.device)
    with H.on_device(pidx.device):
        H.check(H.lib().nr3d_octree_mark_consecutive_segments(H.u32(pack_infos.shape[0]), H.ptr(pidx), H.ptr(pack_infos),
                                                              H.ptr(point_hierarchies), H.ptr(mark_start), H.ptr(mark_end),
                                                              H.stream_of(pidx)))
    return mark_start, mark_end