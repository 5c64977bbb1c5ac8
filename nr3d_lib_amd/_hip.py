"""ctypes loader for libnr3d_hip.so (the C-ABI kernel library, include/nr3d_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a kernel entry point is
called without a GPU tensor, this module raises.  ``build()`` compiles the library in-tree with hipcc
for gfx950 (works without a GPU).
"""
import ctypes as C
import os
import subprocess
import threading

import torch

from nr3d_lib_amd import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnr3d_hip.so")
ABI_VERSION = _abi.ABI_VERSION      # NR3D_ABI_VERSION of include/nr3d_hip.h when _abi.py was generated; the library must report the same
CSRC = os.path.join(_PKG, "csrc")

# dtype codes of include/nr3d_hip.h
F32, F16, F64, I32, I64, U8, I16, I8 = range(8)
DTYPE_CODE = {
    torch.float32: F32, torch.float16: F16, torch.float64: F64, torch.int32: I32, torch.int64: I64,
    torch.uint8: U8, torch.bool: U8, torch.int16: I16, torch.int8: I8,
}

_lib = None
_lock = threading.Lock()


def build(verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950 build of nr3d_lib_amd/libnr3d_hip.so (incremental, via make)."""
    jobs = jobs or os.cpu_count() or 4
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("nr3d_lib_amd: building libnr3d_hip.so failed (see output above)")
    # the signature table the loader uses, regenerated from the header the library was just built against (source checkouts only: a
    # vendored package has no tools/ and keeps the table it came with)
    gen = os.path.join(os.path.dirname(_PKG), "tools", "gen_abi.py")
    if os.path.exists(gen):
        import importlib
        import sys
        r = subprocess.run([sys.executable, gen], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("nr3d_lib_amd: tools/gen_abi.py failed:\n" + r.stdout)
        importlib.reload(_abi)
        globals()["ABI_VERSION"] = _abi.ABI_VERSION
    return LIB_PATH


def lib():
    """The loaded C-ABI library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"nr3d_lib_amd: {LIB_PATH} not found. Build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C nr3d_lib_amd/csrc`. "
                        "There is no CPU fallback for the kernel path.")
                l = C.CDLL(LIB_PATH)
                # a stale build behind a newer binding would take arguments in the wrong places (an int where a pointer is
                # expected: wild writes, not an error) -- refuse it here, not in a test
                have = int(l.nr3d_abi_version())
                if have != ABI_VERSION:
                    raise RuntimeError(f"nr3d_lib_amd: {LIB_PATH} has ABI version {have}, these bindings need {ABI_VERSION}: "
                                       "rebuild it (`make -C nr3d_lib_amd/csrc` or __graft_entry__.build())")
                _declare(l)
                _lib = l
    return _lib


_CTYPE = {"int": C.c_int, "int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
          "float": C.c_float, "double": C.c_double, "void": None, "str": C.c_char_p, "ptr": C.c_void_p}


def _declare(l):
    """argtypes / restype of EVERY entry point, from the table tools/gen_abi.py generated out of include/nr3d_hip.h at build time
    (nr3d_lib_amd/_abi.py -- inside the package: a vendored copy needs no header).  With them the bindings pass plain Python
    ints / floats -- ``ptr()`` is ``tensor.data_ptr()``, no ctypes object per argument (a launch-bound op spent ~10 us per iteration
    wrapping ~50 arguments) -- and a Python int can no longer be taken for a 32-bit C int where a pointer is meant: every pointer
    parameter is declared c_void_p (which also takes None, ctypes arrays and byref() results).  An entry point of the table that
    the library does not export is an error here, not a crash later; that the table matches the header one to one, and the
    library's exports match both, is tests/test_boundary_cpu.py's job (it regenerates the table and runs nm)."""
    missing = []
    for name, (ret, args) in _abi.SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = _CTYPE[ret]
        fn.argtypes = [_CTYPE[a] for a in args]
    if missing:
        raise RuntimeError(f"nr3d_lib_amd: {LIB_PATH} does not export {missing} (stale build? `make -C nr3d_lib_amd/csrc`)")


# ---- debug hook: poison every uninitialised output -----------------------------------------------------------------
# The bindings allocate outputs the kernels are documented to write completely (include/nr3d_hip.h: "fully written")
# with empty().  With NR3D_POISON_EMPTY=1 (the whole `-m gpu` test run sets it, tests/conftest.py) such a buffer is
# filled with NaN / 0xAB bytes before the launch, so an element a kernel forgets shows up in the parity tests instead of
# passing on allocator-reuse luck (a fresh block from the caching allocator is often zero).
POISON = os.environ.get("NR3D_POISON_EMPTY", "0") == "1"


def _poison(t):
    if t.numel() == 0:
        return t
    if t.dtype.is_floating_point:
        t.fill_(float("nan"))
    elif t.dtype == torch.bool:
        t.fill_(True)
    else:
        t.fill_(-0x54545455 if t.dtype in (torch.int32, torch.int64) else 0x2B)     # 0xABABABAB as int32
    return t


def empty(*args, **kwargs):
    t = torch.empty(*args, **kwargs)
    return _poison(t) if POISON else t


def empty_like(x, **kwargs):
    t = torch.empty_like(x, **kwargs)
    return _poison(t) if POISON else t


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().nr3d_last_error().decode())


# ---- optional per-kernel HIP-event timers (include/nr3d_hip.h: NR3D_PROF_*) -------------------------------
PROF_IDS = dict(lotd_fwd=0, lotd_fwd_lds=1, lotd_contract_dx=2, lotd_bin=3, lotd_accum=4, march=5, composite_fwd=6,
                composite_bwd=7, lotd_direct=8)


def prof_enable(*names):
    """time every launch of the named kernels with an event pair on its stream (no names: everything off)"""
    mask = 0
    for n in names:
        mask |= 1 << PROF_IDS[n]
    lib().nr3d_prof_enable(C.c_uint32(mask))


def prof_read(name, reset=True):
    """(total ms, number of intervals) recorded for `name` since the last reset; synchronises on the events"""
    ms, n = C.c_double(0.0), C.c_uint32(0)
    check(lib().nr3d_prof_read(C.c_int(PROF_IDS[name]), C.byref(ms), C.byref(n), C.c_int(1 if reset else 0)))
    return float(ms.value), int(n.value)


# ---- selectable code paths (include/nr3d_hip.h: NR3D_OPT_*) -------------------------------------------------------
OPTION_IDS = dict(lotd_pair=0, pair_quad=1, pair_second=2, pair_direct=3, pair_fixed=4, fwd_pairlane=5, fwd_split=6,
                  fwd_lds_stage=7, hvp_levels=8, hvp_pairlane=9, hvp_split=10, vm_split=11, cp_direct=12, march_group=13,
                  pack_scan=14, vm_lines_direct=15, fwd_cell_major=16, sort_wave=17, vm_direct=18, direct_fixed=19, vm_sorted=20, mlp_x3=21)


def set_option(name, value):
    """choose between two implementations of the same result (A/B measurement, cross-checks in the tests); value < 0 (or None)
    restores the default.  TEST / MEASUREMENT ONLY, process-wide (it has to reach the launches of autograd's device thread):
    include/nr3d_hip.h."""
    l = lib()
    check(l.nr3d_set_option(OPTION_IDS[name], -1 if value is None else int(value)))


def get_option(name):
    l = lib()
    return int(l.nr3d_get_option(OPTION_IDS[name]))


class options:
    """``with options(pair_quad=0): ...`` -- set, run, restore the previous values"""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def ptr(t):
    """Device (or host) address of a tensor (a plain int: every pointer parameter is declared c_void_p, _declare); None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """hipStream_t of torch's current stream on the tensor's device (the raw handle: building a torch.cuda.Stream object
    per call costs ~5 us of host time, which the launch-bound ops notice)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def sort_pairs_u32(keys, values=None, bits=32, n_dev=None):
    """the library's own stable LSD radix sort (csrc/rsort.hip; nr3d_sort_pairs_u32): keys int32 [n] or a pair of them (two
    independent sorts in the same launches), read as uint32 and ordered by their low `bits` bits; values int32 [n] (None: the
    element indices); n_dev int32 [1] on the device: only the first n_dev[0] elements take part (the rest of the outputs is
    left as allocated).  Returns (sorted keys, values in that order) -- lists when a pair was given.  Internal (the sorted-points
    dL/dparam path uses it); exported for tests/test_rsort_gpu.py."""
    pair = isinstance(keys, (list, tuple))
    ks = list(keys) if pair else [keys]
    vs = list(values) if isinstance(values, (list, tuple)) else [values] * len(ks)
    assert len(ks) in (1, 2) and all(k.is_cuda and k.dtype == torch.int32 and k.is_contiguous() and k.shape == ks[0].shape for k in ks)
    assert all(v is None or (v.is_cuda and v.dtype == torch.int32 and v.is_contiguous() and v.shape == ks[0].shape) for v in vs)
    n, dev = ks[0].numel(), ks[0].device
    l = lib()
    l.nr3d_sort_pairs_u32_tmp_bytes.restype = C.c_uint64
    tmp = torch.empty(int(l.nr3d_sort_pairs_u32_tmp_bytes(C.c_uint32(n), C.c_int(len(ks)))) or 1, dtype=torch.uint8, device=dev)
    ko, vo = [empty(n, dtype=torch.int32, device=dev) for _ in ks], [empty(n, dtype=torch.int32, device=dev) for _ in ks]
    a = []
    for i in range(2):
        j = i if i < len(ks) else None
        a += [ptr(ks[j]) if j is not None else None, ptr(vs[j]) if j is not None else None,
              ptr(ko[j]) if j is not None else None, ptr(vo[j]) if j is not None else None]
    with on_device(dev):
        check(l.nr3d_sort_pairs_u32(ptr(tmp), C.c_int(len(ks)), *a, C.c_uint32(n), ptr(n_dev), C.c_int(int(bits)), stream_of(ks[0])))
    return (ko, vo) if pair else (ko[0], vo[0])


def spatial_order(x, bits_per_dim=6):
    """order int32 [n]: order[k] = the row of x [n, 3] (float32, contiguous, CUDA) at position k of a Morton curve through a
    2^bits_per_dim grid over the batch's bounding box, ties in input order (nr3d_spatial_order)"""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3 and x.is_contiguous()
    n, dev = x.shape[0], x.device
    l = lib()
    l.nr3d_spatial_order_tmp_bytes.restype = C.c_uint64
    order = empty(n, dtype=torch.int32, device=dev)
    if n:
        tmp = torch.empty(int(l.nr3d_spatial_order_tmp_bytes(C.c_uint32(n))), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(l.nr3d_spatial_order(C.c_uint32(n), ptr(x), C.c_uint32(int(bits_per_dim)), ptr(order), ptr(tmp), stream_of(x)))
    return order


def order_gather_inputs(order, x, ridx=None, dirs=None):
    """(x[order], ridx[order] | None, dirs[ridx[order]] | None) in one launch (nr3d_order_gather_inputs): x float32 [n, 3], ridx int64
    [n], dirs float32 [n_rays, 3]"""
    n, dev = x.shape[0], x.device
    assert x.dtype == torch.float32 and x.is_contiguous() and order.dtype == torch.int32 and order.shape[0] == n
    assert ridx is None or (ridx.dtype == torch.int64 and ridx.is_contiguous() and ridx.shape[0] == n)
    assert dirs is None or (ridx is not None and dirs.dtype == torch.float32 and dirs.is_contiguous() and dirs.shape[1] == 3)
    x_s = empty((n, 3), dtype=torch.float32, device=dev)
    r_s = empty(n, dtype=torch.int64, device=dev) if ridx is not None else None
    d_s = empty((n, 3), dtype=torch.float32, device=dev) if dirs is not None else None
    with on_device(dev):
        check(lib().nr3d_order_gather_inputs(C.c_uint32(n), ptr(order), ptr(x), ptr(ridx), ptr(dirs), ptr(x_s), ptr(r_s), ptr(d_s), stream_of(x)))
    return x_s, r_s, d_s


def order_move_rows(order, a, b=None, scatter=True):
    """rows of one or two float32 arrays [n, ...] between the spatial order and the samples' own order (nr3d_order_move_rows):
    scatter: out[order[k]] = in[k]; else out[k] = in[order[k]]"""
    n, dev = a.shape[0], a.device
    a = a.contiguous()
    wa = a.numel() // max(n, 1)
    a_out = empty(a.shape, dtype=torch.float32, device=dev)
    wb, b_out = 0, None
    if b is not None:
        b = b.contiguous()
        wb = b.numel() // max(n, 1)
        b_out = empty(b.shape, dtype=torch.float32, device=dev)
    with on_device(dev):
        check(lib().nr3d_order_move_rows(C.c_uint32(n), ptr(order), C.c_int(1 if scatter else 0), ptr(a), C.c_uint32(wa), ptr(a_out),
                                         ptr(b), C.c_uint32(wb), ptr(b_out), stream_of(a)))
    return a_out, b_out


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def on_device(dev):
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager is ~3 us per use)"""
    idx = dev.index if isinstance(dev, torch.device) else int(dev)
    if idx is None or idx == torch.cuda.current_device():
        return _NOCTX
    return torch.cuda.device(idx)


_tls = threading.local()


def read_i64(t):
    """the (few) int64 values of device tensor ``t`` as python ints: THE device->host sync of a two-phase op.  Goes through
    a pinned staging buffer (async copy + stream sync) instead of ``.item()`` / ``.tolist()``, which stage through
    pageable memory: ~10 us less per op, which is visible where an op is launch bound (configs[2]: 4096 rays).

    The staging buffers are THREAD-LOCAL (keyed by device and size inside the thread): two host threads working on
    different streams of one device never see each other's counts between the copy and the read
    (tests/test_occ_grid_gpu.py::test_marcher_two_threads_two_streams).  Within a thread the copy, the stream
    synchronisation and the read happen back to back, so one buffer per (device, n) is enough there."""
    n = t.numel()
    pinned = getattr(_tls, "pinned", None)
    if pinned is None:
        pinned = _tls.pinned = {}
    key = (t.device.index, n)
    buf = pinned.get(key)
    if buf is None:
        buf = pinned[key] = torch.empty(n, dtype=torch.int64, pin_memory=True)
    buf.copy_(t.view(-1), non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.tolist()


def host_i64(n, device):
    """thread-local pinned int64 [n] for a kernel to write its totals INTO (pinned host memory is device-visible at the same
    address): the count kernel's last store lands in host memory, so the readback is a stream synchronisation and nothing
    else -- no device->host copy launch behind the kernel (~10 us of a two-phase op that is launch bound at 4096 rays).
    Pass ``ptr(buf)`` where include/nr3d_hip.h takes `int64_t *total`, then ``wait_i64(buf, device)``."""
    pinned = getattr(_tls, "host_out", None)
    if pinned is None:
        pinned = _tls.host_out = {}
    key = (device.index, n)
    buf = pinned.get(key)
    if buf is None:
        buf = pinned[key] = torch.empty(n, dtype=torch.int64, pin_memory=True)
    buf.fill_(_SENTINEL)                    # what wait_i64 polls against (totals are counts: never negative)
    return buf


_SENTINEL = -1
POLL_TIMEOUT_US = 5000


def wait_i64(buf, device):
    """the values a kernel wrote into ``host_i64``'s buffer: THE device->host sync of a two-phase op.  The kernels store their totals
    with system scope, so the host polls the pinned words (nr3d_wait_host_words: ~2 us after the store) instead of draining the
    stream (~20 us of wake-up latency, paid on the critical path of every launch-bound op); if the words do not arrive within
    POLL_TIMEOUT_US (not observed; e.g. a pinned pool that is not host-coherent), the stream is drained as before."""
    if lib().nr3d_wait_host_words(buf.data_ptr(), buf.numel(), _SENTINEL, POLL_TIMEOUT_US) != 0:
        torch.cuda.current_stream(device).synchronize()
    return buf.tolist()


def mark_ordered(pack_infos, total=None):
    """tag a pack_infos tensor whose packs are known to be ordered and disjoint (begin[p+1] >= begin[p] + len[p]) BY
    CONSTRUCTION -- the output of a marcher, an interleave_* producer or get_pack_infos_from_n.  The launch-bound pack ops
    (packed_cumsum / cumprod / diff / add ...) then let the kernel zero the rows outside the packs (``ordered_packs`` of
    include/nr3d_hip.h) instead of a zero-fill launch in front of it.  The tag is the tensor's version counter: an
    in-place edit invalidates it, and a tensor from anywhere else (a slice, a clone, the user's own arithmetic) has no
    tag, so those take the zero-filled path.
    ``total``: the producer also knows (on the host) that the packs TILE rows [0, total) without a gap -- a marcher's sample
    count, the scalar a two-phase op read back.  Ops whose kernels write the rows of the packs only (packed_alpha_to_vw)
    allocate `empty` outputs when the tensor they are called on has exactly that many rows (``tiles``), zeros otherwise.
    LIMIT (round-4 advisor): the version counter sees in-place torch ops only.  A write through ``pack_infos.data`` or by a kernel
    that got the raw pointer does not bump it, so the tag survives such an edit and the ops would trust a claim that no longer
    holds (rows of other packs zeroed by the gap fill, unwritten `empty` rows).  Editing a producer's pack_infos that way is NOT
    supported: clone it first (a clone carries no tag) or call ``clear_ordered``.  tests/test_host_logic_cpu.py pins both halves."""
    try:
        pack_infos._nr3d_ordered = pack_infos._version
        pack_infos._nr3d_total = None if total is None else int(total)
    except RuntimeError:                    # inference tensors have no version counter: untagged
        pass
    return pack_infos


def clear_ordered(pack_infos):
    """drop the tag (after an edit the version counter cannot see: ``.data`` writes, raw-pointer kernels)"""
    for a in ("_nr3d_ordered", "_nr3d_total"):
        if hasattr(pack_infos, a):
            delattr(pack_infos, a)
    return pack_infos


def is_ordered(pack_infos):
    try:
        return getattr(pack_infos, "_nr3d_ordered", -1) == pack_infos._version
    except RuntimeError:
        return False


def tiles(pack_infos, n_rows):
    """the packs are known to cover rows [0, n_rows) exactly (see mark_ordered)"""
    return is_ordered(pack_infos) and getattr(pack_infos, "_nr3d_total", None) == int(n_rows)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("nr3d_lib_amd: kernels run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback in the product path")


# scalar arguments: plain Python values (the entry points' argtypes convert them, _declare)
def i64(v):
    return int(v)


def u32(v):
    return int(v)


def i32(v):
    return int(v)


def f32(v):
    return float(v)
