"""GPU parity: pack_ops HIP kernels vs the CPU oracle.  Offsets / indices / selectors bit-exact, values
within REL_TOL (wave-parallel reductions reorder the reference's serial sums)."""
import numpy as np
import pytest
import torch

from util import assert_close, assert_equal, random_packs

pytestmark = pytest.mark.gpu


@pytest.fixture()
def P(hiplib):
    from nr3d_lib_amd.bindings import _pack_ops
    return _pack_ops


def T(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


PACK_SHAPES = [(37, 0, 5, 0.3), (64, 1, 200, 0.1), (5, 300, 700, 0.0)]   # (n_packs, lo, hi, empty_frac)


@pytest.mark.parametrize("shape", PACK_SHAPES)
@pytest.mark.parametrize("fd", [None, 1, 3])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64, np.int32])
def test_sum_scan_diff(oracle, dev, P, shape, fd, dtype):
    rng = np.random.default_rng(hash((shape, fd, dtype.__name__)) % 2 ** 31)
    pi, S = random_packs(rng, *shape)
    shp = (S,) if fd is None else (S, fd)
    if np.issubdtype(dtype, np.floating):
        x = rng.uniform(0.5, 1.5, shp).astype(dtype)
    else:
        x = rng.integers(-3, 4, shp).astype(dtype)
    rel = 1e-5 if dtype == np.float32 else 1e-12
    cmp = assert_close if np.issubdtype(dtype, np.floating) else (lambda a, b, rel=None, name="": assert_equal(a, b, name))
    cmp(P.packed_sum(T(x, dev), T(pi, dev)), oracle.packed_sum(x, pi), rel=rel, name="sum")
    for excl in (False, True):
        for rev in (False, True):
            cmp(P.packed_cumsum(T(x, dev), T(pi, dev), excl, rev), oracle.packed_cumsum(x, pi, excl, rev), rel=rel,
                name=f"cumsum e{excl} r{rev}")
            cmp(P.packed_cumprod(T(x, dev), T(pi, dev), excl, rev), oracle.packed_cumprod(x, pi, excl, rev),
                rel=max(rel, 1e-4 if shape[2] > 100 else rel), name=f"cumprod e{excl} r{rev}")
    npk = pi.shape[0]
    e = (rng.uniform(0.5, 1.5, (npk,) + shp[1:]) if np.issubdtype(dtype, np.floating)
         else rng.integers(-3, 4, (npk,) + shp[1:])).astype(dtype)
    for a, f in ((None, None), (e, None), (None, e)):
        assert_equal(P.packed_diff(T(x, dev), T(pi, dev), T(a, dev), T(f, dev)), oracle.packed_diff(x, pi, a, f), "diff")
        assert_equal(P.packed_backward_diff(T(x, dev), T(pi, dev), T(a, dev), T(f, dev)),
                     oracle.packed_backward_diff(x, pi, a, f), "backward_diff")


@pytest.mark.parametrize("head,tail", [(0, 0), (3, 0), (0, 70), (5, 9)])
@pytest.mark.parametrize("fd", [None, 3])
def test_ordered_packs_kernel_zeroes_rows_outside_packs(oracle, dev, P, head, tail, fd):
    """pack_infos tagged as ordered (_hip.mark_ordered: what the library's own producers emit): no zero-fill launch, the kernel
    writes the rows in front of / between / behind the packs.  Runs with poisoned empty() buffers (conftest), so a row the
    kernel misses is a NaN here; packs with gaps and empty packs, outputs equal to the untagged (zero-filled) path's"""
    from nr3d_lib_amd import _hip as H
    rng = np.random.default_rng(head * 100 + tail + (fd or 0))
    lens = rng.integers(0, 90, 41)
    lens[rng.random(41) < 0.25] = 0
    gaps = rng.integers(0, 4, 41) * (rng.random(41) < 0.5)
    begin = head + np.cumsum(lens + gaps) - lens                        # a gap in FRONT of each pack
    pi = np.stack([begin, lens], 1).astype(np.int64)
    S = int(begin[-1] + lens[-1] + tail)
    shp = (S,) if fd is None else (S, fd)
    x = rng.uniform(0.5, 1.5, shp).astype(np.float32)
    pit = H.mark_ordered(T(pi, dev))
    assert H.is_ordered(pit) and P._ordered(pit) == 1
    for excl in (False, True):
        for rev in (False, True):
            assert_close(P.packed_cumsum(T(x, dev), pit, excl, rev), oracle.packed_cumsum(x, pi, excl, rev), rel=1e-5, name="cumsum")
            assert_close(P.packed_cumprod(T(x, dev), pit, excl, rev), oracle.packed_cumprod(x, pi, excl, rev), rel=1e-4, name="cumprod")
    e = rng.uniform(0.5, 1.5, (41,) + shp[1:]).astype(np.float32)
    for a, f in ((None, None), (e, None), (None, e)):
        assert_equal(P.packed_diff(T(x, dev), pit, T(a, dev), T(f, dev)), oracle.packed_diff(x, pi, a, f), "diff")
        assert_equal(P.packed_backward_diff(T(x, dev), pit, T(a, dev), T(f, dev)), oracle.packed_backward_diff(x, pi, a, f), "bdiff")
    o = rng.uniform(0.5, 2.0, (41,) + shp[1:]).astype(np.float32)
    for op in ("add", "div", "gt", "neq"):
        assert_equal(getattr(P, f"packed_{op}")(T(x, dev), T(o, dev), pit), oracle.packed_binary(op, x, o, pi), op)
    if fd:
        w = rng.standard_normal((41, 2, fd)).astype(np.float32)
        assert_close(P.packed_matmul(T(x, dev), T(w, dev), pit), oracle.packed_matmul(x, w, pi), name="matmul")
    # an in-place edit drops the tag (version counter), a slice never had it
    assert not H.is_ordered(pit[:5])
    pit[0, 1] += 0
    assert not H.is_ordered(pit) and P._ordered(pit) == 0


def test_reference_known_answers(dev, P):
    """literal tensors of the reference's own test (graphics/pack_ops/unit_test.py:536-563, 4 printed digits)"""
    boundary = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0], bool)
    first = np.nonzero(boundary)[0]
    pi = np.stack([first, np.diff(np.append(first, 13))], 1).astype(np.int64)
    feat = np.array([0.8750, 0.0581, 0.9378, 0.9638, 0.9859, 0.4652, 0.9105, 0.5071, 0.0173, 0.6071, 0.7123,
                     0.7371, 0.8094], np.float32)
    incl = np.array([0.8750, 0.9331, 1.8709, 2.8347, 0.9859, 1.4512, 2.3617, 2.8688, 2.8860, 3.4931, 4.2054,
                     0.7371, 1.5465], np.float32)
    excl = np.array([0.0, 0.8750, 0.9331, 1.8709, 0.0, 0.9859, 1.4512, 2.3617, 2.8688, 2.8860, 3.4931, 0.0,
                     0.7371], np.float32)
    np.testing.assert_allclose(P.packed_cumsum(T(feat, dev), T(pi, dev), False, False).cpu().numpy(), incl, atol=2e-4)
    np.testing.assert_allclose(P.packed_cumsum(T(feat, dev), T(pi, dev), True, False).cpu().numpy(), excl, atol=2e-4)
    featp = np.array([-0.8033, 1.2413, 0.1971, 1.2183, 1.3434, 1.7485, 0.0624, -0.3419, -0.1997, -1.4790, 1.1720,
                      0.1686, -0.0704], np.float32)
    inclp = np.array([-0.8033, -0.9972, -0.1965, -0.2394, 1.3434, 2.3489, 0.1465, -0.0501, 0.0100, -0.0148, -0.0173,
                      0.1686, -0.0119], np.float32)
    np.testing.assert_allclose(P.packed_cumprod(T(featp, dev), T(pi, dev), False, False).cpu().numpy(), inclp, atol=2e-4)
    # exclusive cumprod: the reference KERNEL yields zeros (see DESIGN.md quirk list); the documented
    # right-shift-with-1 semantics (the kaolin numbers of the docstring) are available behind the flag
    assert float(P.packed_cumprod(T(featp, dev), T(pi, dev), True, False).abs().max()) == 0.0
    P.CUMPROD_EXCLUSIVE_DOCUMENTED = True
    try:
        exclp = np.array([1.0, -0.8033, -0.9972, -0.1965, 1.0, 1.3434, 2.3489, 0.1465, -0.0501, 0.0100, -0.0148, 1.0,
                          0.1686], np.float32)
        np.testing.assert_allclose(P.packed_cumprod(T(featp, dev), T(pi, dev), True, False).cpu().numpy(), exclp, atol=2e-4)
    finally:
        P.CUMPROD_EXCLUSIVE_DOCUMENTED = False
    assert_equal(P.mark_pack_boundaries_cuda(T(np.repeat(np.arange(3), [4, 7, 2]).astype(np.int64), dev)),
                 boundary.astype(np.int32), "boundaries")
    # docs/pack_ops.md:21
    from nr3d_lib_amd.graphics.pack_ops import get_pack_infos_from_n
    assert get_pack_infos_from_n(torch.tensor([4, 1, 5, 3])).tolist() == [[0, 4], [4, 1], [5, 5], [10, 3]]


@pytest.mark.parametrize("shape", PACK_SHAPES)
@pytest.mark.parametrize("fd", [None, 4])
def test_binary_ops(oracle, dev, P, shape, fd):
    rng = np.random.default_rng(11)
    pi, S = random_packs(rng, *shape)
    npk = pi.shape[0]
    x = rng.uniform(0.5, 2.0, (S,) if fd is None else (S, fd)).astype(np.float32)
    o = rng.uniform(0.5, 2.0, (npk,) if fd is None else (npk, fd)).astype(np.float32)
    o[::3] = x[np.minimum(pi[::3, 0], max(S - 1, 0))] if S else o[::3]      # force some equalities
    for op in ("add", "sub", "mul", "div"):
        assert_equal(getattr(P, f"packed_{op}")(T(x, dev), T(o, dev), T(pi, dev)), oracle.packed_binary(op, x, o, pi), op)
    for op in ("gt", "geq", "lt", "leq", "eq", "neq"):
        got = getattr(P, f"packed_{op}")(T(x, dev), T(o, dev), T(pi, dev))
        assert got.dtype == torch.bool
        assert_equal(got, oracle.packed_binary(op, x, o, pi), op)
    xi = rng.integers(0, 100, (S,)).astype(np.int64)
    oi = rng.integers(0, 100, (npk,)).astype(np.int64)
    assert_equal(P.packed_add(T(xi, dev), T(oi, dev), T(pi, dev)), oracle.packed_binary("add", xi, oi, pi), "add i64")
    if fd:
        w = rng.standard_normal((npk, 3, fd)).astype(np.float32)
        assert_close(P.packed_matmul(T(x, dev), T(w, dev), T(pi, dev)), oracle.packed_matmul(x, w, pi), name="matmul")


def test_interleave(oracle, dev, P):
    rng = np.random.default_rng(12)
    n = rng.integers(0, 300, 97).astype(np.int64)
    out, nidx = P.interleave_arange(T(n, dev), True)
    ro, rn = oracle.interleave_arange(n, True)
    assert_equal(out, ro, "arange"); assert_equal(nidx, rn, "arange nidx")
    assert P.interleave_arange(T(n, dev), False)[1] is None
    start = rng.uniform(0, 10, 97).astype(np.float32)
    step = rng.uniform(0.1, 1, 97).astype(np.float32)
    for ss, rs in ((T(step, dev), step), (0.25, 0.25)):
        out, nidx = P.interleave_linstep(T(start, dev), T(n, dev), ss, True)
        ro, rn = oracle.interleave_linstep(start, n, rs, True)
        assert_equal(out, ro, "linstep"); assert_equal(nidx, rn, "linstep nidx")
    starti = rng.integers(0, 1000, 97).astype(np.int64)
    out, _ = P.interleave_linstep(T(starti, dev), T(n, dev), 1, False)
    assert_equal(out, oracle.interleave_linstep(starti, n, 1, False)[0], "linstep i64")
    # literal inputs of the reference test (unit_test.py:697-701) through the arange wrapper
    from nr3d_lib_amd.graphics.pack_ops import interleave_arange
    y = interleave_arange(torch.tensor([1.1, 2.2, 3.3], device=dev), torch.tensor([9.9, 5.5, 7.7], device=dev),
                          torch.tensor([0.5, 1.3, 0.9], device=dev), False)
    want = np.concatenate([np.float32(1.1) + np.arange(18, dtype=np.float32) * np.float32(0.5),
                           np.float32(2.2) + np.arange(3, dtype=np.float32) * np.float32(1.3),
                           np.float32(3.3) + np.arange(5, dtype=np.float32) * np.float32(0.9)])
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-6)


def test_interleave_arange_step_counts(dev):
    """interleave_arange's step counts in one launch = the wrapper's ATen expression stop.subtract(start).div(step).ceil().long(),
    bit for bit: int64 / int32 (quotient in float32), float32, float64; tensor and scalar steps"""
    from nr3d_lib_amd.bindings import _pack_ops as B
    import nr3d_lib_amd.graphics.pack_ops as po
    rng = np.random.default_rng(3)
    n = 5000
    for dt in (torch.int64, torch.int32, torch.float32, torch.float64):
        if dt.is_floating_point:
            a = torch.from_numpy(rng.uniform(-3, 3, n)).to(dt).to(dev)
            b = a + torch.from_numpy(rng.uniform(0, 40, n)).to(dt).to(dev)
            st = torch.from_numpy(rng.uniform(0.05, 3, n)).to(dt).to(dev)
            scalars = (0.37, 1, 2.5)
        else:
            a = torch.from_numpy(rng.integers(-5, 5, n)).to(dt).to(dev)
            b = a + torch.from_numpy(rng.integers(0, 100, n)).to(dt).to(dev)
            st = torch.from_numpy(rng.integers(1, 7, n)).to(dt).to(dev)
            scalars = (1, 3, 0.5)
        for step in (st,) + scalars:
            want = b.subtract(a).div(step).ceil().long()
            assert torch.equal(B.arange_num_steps(a, b, step), want), f"{dt} step {type(step).__name__}"
    z, cnt, one = torch.zeros(64, dtype=torch.int64, device=dev), torch.arange(64, device=dev), torch.ones(64, dtype=torch.int64, device=dev)
    v, idx = po.interleave_arange(z, cnt, one)
    assert v.numel() == int(cnt.sum()) and torch.equal(idx, torch.repeat_interleave(torch.arange(64, device=dev), cnt))


@pytest.mark.parametrize("gamma,lo,hi", [(0.01, 0.01, 1.0), (0.0, 0.02, 1e10), (0.05, 0.001, 0.04)])
def test_sample_step(oracle, dev, P, gamma, lo, hi):
    rng = np.random.default_rng(13)
    near = rng.uniform(0.1, 2, 301).astype(np.float32)
    far = (near + rng.uniform(-0.5, 6, 301)).astype(np.float32)     # some far < near -> empty packs
    got = P.interleave_sample_step_wrt_depth_clamped(T(near, dev), T(far, dev), 200, gamma, lo, hi)
    ref = oracle.interleave_sample_step_wrt_depth_clamped(near, far, 200, gamma, lo, hi)
    for g, r, n in zip(got, ref, ["t", "dt", "nidx", "pack_infos"]):
        assert_equal(g, r, n)                                       # serial recurrence replayed exactly
    # segments
    nseg = rng.integers(0, 4, 301).astype(np.int64)
    spi = oracle.get_pack_infos_from_n(nseg)
    entry, exit = [], []
    for i in range(301):
        cuts = np.sort(rng.uniform(near[i], max(far[i], near[i] + 0.1), 2 * nseg[i])).astype(np.float32)
        entry += list(cuts[0::2]); exit += list(cuts[1::2])
    entry, exit = np.array(entry, np.float32), np.array(exit, np.float32)
    got = P.interleave_sample_step_wrt_depth_in_packed_segments(T(near, dev), T(far, dev), T(entry, dev), T(exit, dev),
                                                                T(spi, dev), 64, gamma, lo, hi)
    ref = oracle.interleave_sample_step_wrt_depth_in_packed_segments(near, far, entry, exit, spi, 64, gamma, lo, hi)
    for g, r, n in zip(got, ref, ["t", "dt", "sidx", "nidx", "pack_infos"]):
        assert_equal(g, r, "seg " + n)


def test_search_and_invert_cdf(oracle, dev, P):
    # literal inputs of the reference test (unit_test.py:1125-1143)
    bins = np.array([2, 3, 7, 1, 0, 4, 8, 3, 6], np.float32)
    u = np.array([4, 0, 2, 9, 7], np.float32)[:, None]
    pi = oracle.get_pack_infos_from_n(np.array([3, 1, 2, 1, 2]))
    assert_equal(P.packed_searchsorted(T(bins, dev), T(u, dev), T(pi, dev)), oracle.packed_searchsorted(bins, u, pi), "ss")
    assert P.packed_searchsorted(T(bins, dev), T(u, dev), T(pi, dev)).flatten().tolist() == [2, 3, 5, 6, 8]
    up = np.array([3.5, 1.2, 2.2, -1, 3.2, 6.0, 7.0, 5.0], np.float32)
    upi = oracle.get_pack_infos_from_n(np.array([2, 0, 3, 2, 1]))
    assert_equal(P.packed_searchsorted_packed_vals(T(bins, dev), T(pi, dev), T(up, dev), T(upi, dev)),
                 oracle.packed_searchsorted_packed_vals(bins, pi, up, upi), "ss packed")
    bins = np.array([2, 3, 7, 1, 2, 3, 4, 6, 8], np.float32)
    cdfs = np.array([0.0, 0.4, 1.0, 0.0, 1.0, 0.0, 0.1, 0.8, 1.0], np.float32)
    pi = oracle.get_pack_infos_from_n(np.array([3, 2, 4]))
    uu = np.tile(np.array([0.5, 0.9], np.float32), (3, 1))
    s, b = P.packed_invert_cdf(T(bins, dev), T(cdfs, dev), T(uu, dev), T(pi, dev))
    rs, rb = oracle.packed_invert_cdf(bins, cdfs, uu, pi)
    assert_equal(b, rb, "bin_idx"); assert_equal(s, rs, "samples")
    # random, larger
    rng = np.random.default_rng(14)
    pi, S = random_packs(rng, 150, 1, 90)
    bins = np.concatenate([np.sort(rng.uniform(0, 5, int(n))) for n in pi[:, 1]]).astype(np.float32)
    pdf = rng.uniform(0, 1, S) * (rng.random(S) > 0.2)
    cdfs = np.concatenate([np.cumsum(pdf[b:b + n]) / max(pdf[b:b + n].sum(), 1e-9) for b, n in pi]).astype(np.float32)
    uu = rng.uniform(0, 1, (150, 70)).astype(np.float32)
    s, b = P.packed_invert_cdf(T(bins, dev), T(cdfs, dev), T(uu, dev), T(pi, dev))
    rs, rb = oracle.packed_invert_cdf(bins, cdfs, uu, pi)
    assert_equal(b, rb, "bin_idx rnd"); assert_equal(s, rs, "samples rnd")
    vals = rng.uniform(-1, 6, (150, 70)).astype(np.float32)
    assert_equal(P.packed_searchsorted(T(bins, dev), T(vals, dev), T(pi, dev)), oracle.packed_searchsorted(bins, vals, pi), "ss rnd")


@pytest.mark.parametrize("b_sorted", [True, False])
def test_merge_sorted(oracle, dev, P, b_sorted):
    rng = np.random.default_rng(15)
    pia, Sa = random_packs(rng, 80, 0, 150, 0.1)
    pib, Sb = random_packs(rng, 80, 0, 90, 0.2)
    va = np.concatenate([np.sort(rng.integers(0, 60, int(n))) for n in pia[:, 1]] + [np.zeros(0)]).astype(np.float32)
    # unsorted b: the reference's run-rank only separates CONSECUTIVE equal lower bounds; kernel and oracle
    # implement the same formula, so equality holds either way
    vb = [rng.integers(0, 60, int(n)) + 0.5 * (rng.random(int(n)) > 0.5) for n in pib[:, 1]]
    vb = np.concatenate([np.sort(v) if b_sorted else v for v in vb] + [np.zeros(0)]).astype(np.float32)
    pa, pb, pim = P.try_merge_two_packs_sorted_aligned(T(va, dev), T(pia, dev), T(vb, dev), T(pib, dev), b_sorted)
    ra, rb, rm = oracle.try_merge_two_packs_sorted_aligned(va, pia, vb, pib, b_sorted)
    assert_equal(pim, rm, "merged pack_infos"); assert_equal(pa, ra, "pidx_a"); assert_equal(pb, rb, "pidx_b")
    if b_sorted:   # the merged array really is sorted and a permutation
        merged = np.full(Sa + Sb, np.nan, np.float32)
        merged[pa.cpu().numpy()] = va; merged[pb.cpu().numpy()] = vb
        assert not np.isnan(merged).any()
        for b, n in rm:
            assert (np.diff(merged[b:b + n]) >= 0).all()


@pytest.mark.parametrize("dtype", [np.float32, np.int64, np.float64, np.int32])
@pytest.mark.parametrize("wave", [1, 0], ids=["wave_per_pack", "lane_per_pack"])
def test_sort(oracle, dev, P, dtype, wave, hip_option):
    """packed_sort_qsort: one wave per pack with a bitonic network in registers (default) and one lane per pack (heapsort,
    option sort_wave = 0) -- packs of 0 ... 2600 elements cover every register size class (64 ... 2048) and the fallback"""
    hip_option("sort_wave", wave)
    # literal example of the reference test (unit_test.py:1086-1092)
    vals = np.array([0.2, 0.1, 0.3, 2.9, 2.3, 2.5, 2.4, 2.1, 1.0, 1.1], np.float32)
    pi = oracle.get_pack_infos_from_n(np.array([3, 5, 2]))
    v = T(vals, dev).clone()
    idx = P.packed_sort_qsort(v, T(pi, dev), True)
    assert v.tolist() == pytest.approx([0.1, 0.2, 0.3, 2.1, 2.3, 2.4, 2.5, 2.9, 1.0, 1.1])
    assert torch.equal(T(vals, dev)[idx], v)
    rng = np.random.default_rng(16)
    pi, S = random_packs(rng, 60, 0, 400, 0.1)
    extra = np.array([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 2600])
    n_all = np.concatenate([pi[:, 1], extra])
    pi = oracle.get_pack_infos_from_n(n_all)
    S = int(n_all.sum())
    x = (rng.standard_normal(S) * 100).astype(dtype)
    if np.issubdtype(dtype, np.integer):
        x = (x // 10).astype(dtype)                                        # many equal keys
    else:
        x[rng.random(S) < 0.05] = 0.0                                      # equal keys among floats, and signed zeros
        x[rng.random(S) < 0.02] = -0.0
        x[rng.random(S) < 0.01] = np.inf
        x[rng.random(S) < 0.01] = -np.inf
    v = T(x, dev).clone()
    idx = P.packed_sort_qsort(v, T(pi, dev), True)
    want = np.concatenate([np.sort(x[b:b + n]) for b, n in pi] + [np.zeros(0, dtype)])
    assert_equal(v, want, "sorted values")                                 # (-0.0 == 0.0 compare equal: any order of the two passes)
    assert_equal(T(x, dev)[idx], want, "vals[idx]")
    ii = idx.cpu().numpy()
    for b, n in pi:
        assert sorted(ii[b:b + n].tolist()) == list(range(b, b + n))      # a permutation inside every pack
    if wave:
        # the register network orders (key, position): the STABLE ascending order, for packs it holds in registers
        lim = 2048 if np.dtype(dtype).itemsize == 4 else 1024
        for b, n in pi:
            if 0 < n <= lim:
                seg = x[b:b + n]
                key = seg if np.issubdtype(dtype, np.integer) else np.where(np.signbit(seg) & (seg == 0), -np.finfo(dtype).tiny, seg)   # total order: -0 < +0
                assert np.array_equal(ii[b:b + n] - b, np.argsort(key, kind="stable")), f"pack of {n}: not the stable order"
    # ids need not be an arange: any int64 payload is permuted like the values
    v2 = T(x, dev).clone()
    payload = torch.from_numpy(rng.integers(-2 ** 40, 2 ** 40, S)).to(dev)
    from nr3d_lib_amd import _hip as H
    import ctypes as Cc
    pit = T(pi, dev)
    p0 = payload.clone()
    H.check(H.lib().nr3d_packed_sort(H.u32(pit.shape[0]), Cc.c_uint64(S), Cc.c_int(H.DTYPE_CODE[v2.dtype]), H.ptr(v2), H.ptr(payload),
                                     H.ptr(pit), H.stream_of(v2)))
    assert torch.equal(payload, p0[idx]) and torch.equal(v2, v)
    assert P.packed_sort_qsort(T(x, dev).clone(), T(pi, dev), False) is None


@pytest.mark.parametrize("eps,thre", [(1e-4, 0.0), (1e-2, 0.05), (0.0, 0.0)])
def test_alpha_to_vw(oracle, dev, P, eps, thre):
    rng = np.random.default_rng(17)
    pi, S = random_packs(rng, 257, 0, 300, 0.1)
    alpha = (rng.uniform(0, 1, S) ** 3).astype(np.float32)
    alpha[rng.random(S) < 0.1] = 0.0
    alpha[rng.random(S) < 0.02] = 1.0
    if thre > 0:
        alpha[rng.random(S) < 0.05] = np.float32(thre)                  # exactly on the threshold: fwd <=, bwd <
    w, _, _ = P.packed_alpha_to_vw_forward(T(alpha, dev), T(pi, dev), eps, thre, False)
    rw, _, _ = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, False)
    assert_equal(w, rw, "weights")                                      # serial T replayed exactly
    _, cpi, sel = P.packed_alpha_to_vw_forward(T(alpha, dev), T(pi, dev), eps, thre, True)
    _, rcpi, rsel = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, True)
    assert sel.dtype == torch.bool and cpi.dtype == torch.int64
    assert_equal(sel, rsel, "compact_selector"); assert_equal(cpi, rcpi, "compact_pack_infos")
    gw = rng.standard_normal(S).astype(np.float32)
    ga = P.packed_alpha_to_vw_backward(T(rw, dev), T(gw, dev), T(alpha, dev), T(pi, dev), eps, thre)
    # serial fma chain + IEEE division replayed exactly -> bit-exact (matters: /max(1-alpha,1e-10) amplifies)
    assert_equal(ga, oracle.packed_alpha_to_vw_backward(rw, gw, alpha, pi, eps, thre), "grad_alphas")


def test_alpha_to_vw_many_short_packs(oracle, dev, P):
    """>= 65536 packs switch to the lane-per-pack kernels: same bits as the oracle"""
    rng = np.random.default_rng(23)
    pi, S = random_packs(rng, 70000, 0, 24, 0.05)
    alpha = (rng.uniform(0, 1, S) ** 2).astype(np.float32)
    alpha[rng.random(S) < 0.05] = 1.0
    eps, thre = 1e-4, 0.01
    w, _, _ = P.packed_alpha_to_vw_forward(T(alpha, dev), T(pi, dev), eps, thre, False)
    rw, _, _ = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, False)
    assert_equal(w, rw, "weights")
    _, cpi, sel = P.packed_alpha_to_vw_forward(T(alpha, dev), T(pi, dev), eps, thre, True)
    _, rcpi, rsel = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, True)
    assert_equal(sel, rsel, "compact_selector"); assert_equal(cpi, rcpi, "compact_pack_infos")
    gw = rng.standard_normal(S).astype(np.float32)
    ga = P.packed_alpha_to_vw_backward(T(rw, dev), T(gw, dev), T(alpha, dev), T(pi, dev), eps, thre)
    assert_equal(ga, oracle.packed_alpha_to_vw_backward(rw, gw, alpha, pi, eps, thre), "grad_alphas")


def test_autograd_wrappers(oracle, dev):
    """gradients of the pack_ops.py autograd layer against dense torch autograd on equal-length packs
    (the reference's own strategy, unit_test.py:100-131, 203-210)"""
    import nr3d_lib_amd.graphics.pack_ops as po
    torch.manual_seed(0)
    npk, n, F = 12, 9, 3
    pi = po.get_pack_infos_from_batch(npk, n, device=dev)
    x = torch.rand(npk * n, F, device=dev, dtype=torch.float64) + 0.5
    o = torch.rand(npk, F, device=dev, dtype=torch.float64) + 0.5
    g = torch.randn(npk * n, F, device=dev, dtype=torch.float64)

    def check(fn_packed, fn_dense, inputs, gout):
        a = [t.clone().requires_grad_(True) for t in inputs]
        b = [t.clone().requires_grad_(True) for t in inputs]
        ya, yb = fn_packed(*a), fn_dense(*b)
        torch.testing.assert_close(ya, yb.reshape(ya.shape), rtol=1e-10, atol=1e-12)
        ga = torch.autograd.grad(ya, a, gout.reshape(ya.shape))
        gb = torch.autograd.grad(yb, b, gout.reshape(yb.shape))
        for u, v in zip(ga, gb):
            torch.testing.assert_close(u, v, rtol=1e-9, atol=1e-11)

    d = lambda t: t.view(npk, n, F)
    check(lambda x: po.packed_sum(x, pi), lambda x: d(x).sum(1), [x], torch.randn(npk, F, device=dev, dtype=torch.float64))
    check(lambda x: po.packed_cumsum(x, pi), lambda x: d(x).cumsum(1), [x], g)
    check(lambda x: po.packed_cumsum(x, pi, reverse=True), lambda x: d(x).flip(1).cumsum(1).flip(1), [x], g)
    check(lambda x: po.packed_cumprod(x, pi), lambda x: d(x).cumprod(1), [x], g)
    check(lambda x: po.packed_diff(x, pi), lambda x: torch.cat([d(x).diff(dim=1), torch.zeros_like(d(x)[:, :1])], 1), [x], g)
    check(lambda x, a: po.packed_diff(x, pi, pack_appends=a), lambda x, a: torch.cat([d(x), a[:, None]], 1).diff(dim=1), [x, o], g)
    check(lambda x, a: po.packed_backward_diff(x, pi, pack_prepends=a), lambda x, a: torch.cat([a[:, None], d(x)], 1).diff(dim=1), [x, o], g)
    check(lambda x: po.packed_backward_diff(x, pi), lambda x: torch.cat([torch.zeros_like(d(x)[:, :1]), d(x).diff(dim=1)], 1), [x], g)
    for name, f in (("add", torch.add), ("sub", torch.sub), ("mul", torch.mul), ("div", torch.div)):
        check(lambda x, o, name=name: getattr(po, f"packed_{name}")(x, o, pi), lambda x, o, f=f: f(d(x), o[:, None]), [x, o], g)
    # alpha -> weights against the cumprod formulation
    al = (torch.rand(npk * n, device=dev) * 0.8).float()

    def dense_vw(a):
        a = a.view(npk, n)
        Tr = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a[:, :-1]], 1), 1)
        return (a * Tr).reshape(-1)
    a1, a2 = al.clone().requires_grad_(True), al.clone().requires_grad_(True)
    w1, w2 = po.packed_alpha_to_vw(a1, pi, early_stop_eps=0.0), dense_vw(a2)
    torch.testing.assert_close(w1, w2, rtol=1e-5, atol=1e-7)
    gw = torch.randn_like(w1)
    torch.testing.assert_close(torch.autograd.grad(w1, a1, gw)[0], torch.autograd.grad(w2, a2, gw)[0], rtol=2e-4, atol=1e-6)
    # compression helper
    nidx_useful, cpi, pidx = po.packed_volume_render_compression(al, pi, early_stop_eps=0.05, alpha_thre=0.1)
    assert cpi.dtype == torch.int64 and int(cpi[:, 1].sum()) == pidx.numel() and (cpi[:, 1] > 0).all()


def test_merge_wrappers(oracle, dev):
    """merge_two_packs_sorted* (pure-torch orchestration over the kernels): merged values sorted per pack"""
    import nr3d_lib_amd.graphics.pack_ops as po
    rng = np.random.default_rng(18)
    nidx_a = np.sort(rng.choice(40, 25, replace=False))
    nidx_b_sub = np.sort(rng.choice(nidx_a, 11, replace=False))
    nidx_b_any = np.sort(rng.choice(40, 20, replace=False))

    def mk(nidx):
        n = rng.integers(1, 30, len(nidx))
        pi = oracle.get_pack_infos_from_n(n)
        v = np.concatenate([np.sort(rng.uniform(0, 1, int(k))) for k in n]).astype(np.float32)
        return T(v, dev), T(pi, dev), T(nidx.astype(np.int64), dev)
    va, pia, na = mk(nidx_a)
    for nb, fn in ((nidx_b_sub, po.merge_two_packs_sorted_a_includes_b), (nidx_b_any, po.merge_two_packs_sorted),
                   (nidx_a, po.merge_two_packs_sorted)):
        vb, pib, nbt = mk(nb)
        val, pim = fn(va, pia, na, vb, pib, nbt, return_val=True)
        assert val.numel() == va.numel() + vb.numel()
        for b, n in pim.tolist():
            assert (val[b:b + n].diff() >= 0).all()
        assert torch.equal(torch.sort(val)[0], torch.sort(torch.cat([va, vb]))[0])


@pytest.mark.parametrize("dtype", [np.float32, np.int64])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_merge_two_packs_sorted_native_union(oracle, dev, dtype, seed):
    """merge_two_packs_sorted on arbitrary pack-id lists: the native route (nr3d_merge_pack_union: union of the id lists in aligned
    form + scan + the aligned merge kernel, one readback) gives exactly the positions and pack_infos of the reference's
    torch.unique / nonzero formulation (kept as _merge_two_packs_sorted_torch); disjoint lists, one list inside the other, empty
    packs, equal values across a and b"""
    import nr3d_lib_amd.graphics.pack_ops as po
    from nr3d_lib_amd.graphics.pack_ops import pack_ops as pom
    rng = np.random.default_rng(100 + seed)
    ids = np.arange(3000)
    cases = [(np.sort(rng.choice(ids, 900, replace=False)), np.sort(rng.choice(ids, 700, replace=False))),
             (np.sort(rng.choice(ids[:1500], 400, replace=False)), np.sort(rng.choice(ids[1500:], 500, replace=False))),   # disjoint
             (np.array([5]), np.array([2, 5, 9])), (np.array([1, 2, 3]), np.array([0])), (np.array([7]), np.array([8]))]

    def mk(nidx):
        n = rng.integers(0, 40, len(nidx))
        if n.sum() == 0:
            n[0] = 3
        pi = oracle.get_pack_infos_from_n(n)
        if dtype == np.float32:
            v = np.concatenate([np.sort(rng.integers(0, 50, int(k)) / 50.0) for k in n]).astype(np.float32)     # many ties
        else:
            v = np.concatenate([np.sort(rng.integers(0, 50, int(k))) for k in n]).astype(np.int64)
        return T(v, dev), T(pi, dev), T(nidx.astype(np.int64), dev)
    for ida, idb in cases:
        va, pia, na = mk(ida)
        vb, pib, nb = mk(idb)
        got = po.merge_two_packs_sorted(va, pia, na, vb, pib, nb)
        want = pom._merge_two_packs_sorted_torch(va, pia, na, vb, pib, nb)
        for g, w, name in zip(got, want, ("pidx_a", "pidx_b", "pack_infos")):
            assert g.dtype == w.dtype and torch.equal(g, w), f"{name}: {len(ida)} + {len(idb)} packs"
        val, pim = po.merge_two_packs_sorted(va, pia, na, vb, pib, nb, return_val=True)
        assert torch.equal(torch.sort(val)[0], torch.sort(torch.cat([va, vb]))[0])
        for b, n in pim.tolist():
            assert (val[b:b + n].diff() >= 0).all()


def test_octree_mark_consecutive_segments(oracle, dev):
    """the reference's own example (unit_test.py:760-781: nodes x = 0,1,2 | 5,6 on one ray -> two runs) plus ragged packs"""
    from nr3d_lib_amd.graphics.pack_ops import octree_mark_consecutive_segments
    pts = torch.tensor([[0, 3, 3], [1, 3, 3], [2, 3, 3], [5, 3, 3], [6, 3, 3]], dtype=torch.int16, device=dev)
    ms, me = octree_mark_consecutive_segments(torch.arange(5, device=dev), torch.tensor([[0, 5]], device=dev), pts)
    assert ms.tolist() == [True, False, False, True, False] and me.tolist() == [False, False, True, False, True]
    rng = np.random.default_rng(0)
    hier = rng.integers(0, 6, (400, 3)).astype(np.int16)
    pinfo, total = random_packs(rng, 300, 0, 70, empty_frac=0.1)
    # walks: mostly unit steps between consecutive nodes, so that both outcomes occur
    walk = np.cumsum(rng.integers(-1, 2, (400, 3)) * (rng.random((400, 1)) < 0.8), 0).astype(np.int16)
    pidx = rng.integers(0, 400, total).astype(np.int32)
    pidx[1:] = np.where(rng.random(total - 1) < 0.7, np.minimum(pidx[:-1] + 1, 399), pidx[1:])
    want_s, want_e = oracle.octree_mark_consecutive_segments(pidx, pinfo, walk)
    got_s, got_e = octree_mark_consecutive_segments(torch.from_numpy(pidx).to(dev), torch.from_numpy(pinfo).to(dev), torch.from_numpy(walk).to(dev))
    assert_equal(got_s, want_s, "mark_start"); assert_equal(got_e, want_e, "mark_end")
    assert want_s.sum() == want_e.sum() > (pinfo[:, 1] > 0).sum()            # some packs split into several runs
    assert hier.shape == (400, 3)


# ---------------------------------------------------------------------------------------------------------
# fused composite (graphics.pack_ops.packed_composite) against the oracle's op CHAIN
# (renderer_mixin.py:298-311: packed_alpha_to_vw -> packed_sum -> packed_div -> packed_sum(. * t) -> packed_sum(. * rgb))
# ---------------------------------------------------------------------------------------------------------
def _chain_forward(oracle, alpha, t, rgb, pi, eps, thre, normalize):
    """the reference's chain on the oracle's pack ops, fp32 like the reference"""
    vw = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, False)[0]
    mask = oracle.packed_sum(vw, pi)
    if normalize:
        wn = oracle.packed_binary("div", vw, (mask + np.float32(1e-10)).astype(np.float32), pi)
        depth = oracle.packed_sum((wn * t).astype(np.float32), pi)
    else:
        depth = oracle.packed_sum((vw * t).astype(np.float32), pi)
    col = oracle.packed_sum((vw[:, None] * rgb).astype(np.float32), pi) if rgb is not None else None
    return vw, mask, depth, col


def _chain_backward(oracle, alpha, t, rgb, pi, eps, thre, normalize, vw, mask, g_mask, g_depth, g_rgb, g_vw):
    """autograd of the chain written out (pack_ops.py:97-116 PackedSum, :293-392 PackedDiv, :261-283 PackedAlphaToVW),
    in float64 for everything but the alpha_to_vw backward kernel itself"""
    n = pi[:, 1]
    rep = lambda a: np.repeat(a, n, axis=0)
    s = mask.astype(np.float64) + 1e-10
    gw = rep(g_mask.astype(np.float64))
    if normalize:
        wt = np.bincount(np.repeat(np.arange(len(n)), n), weights=vw.astype(np.float64) * t, minlength=len(n))
        gw = gw + rep(g_depth / s) * t - rep(g_depth * wt / s ** 2)
        g_t = rep(g_depth / s) * vw
    else:
        gw = gw + rep(g_depth.astype(np.float64)) * t
        g_t = rep(g_depth.astype(np.float64)) * vw
    g_c = None
    if rgb is not None:
        gw = gw + (rep(g_rgb.astype(np.float64)) * rgb).sum(1)
        g_c = rep(g_rgb.astype(np.float64)) * vw[:, None]
    if g_vw is not None:
        gw = gw + g_vw
    g_a = oracle.packed_alpha_to_vw_backward(vw, gw.astype(np.float32), alpha, pi, eps, thre)
    return g_a, g_t, g_c, gw


def _alpha_to_vw_backward_f64(alpha, gw64, pi, eps, thre):
    """kernel_packed_alpha_to_vw_forward / _backward (pack_ops_cuda.cu:1735-1848; oracle/pack_ops_oracle.c:321-360) with every VALUE
    in float64 -- weights, running sum, transmittance -- so the result carries no fp32 rounding of its own.  Every DECISION (the
    early stop T < eps, the threshold tests, forward `a <= thre` / backward `a < thre` as in the reference) is taken on the float32
    recurrences of the reference, which run alongside."""
    ga = np.zeros(alpha.shape[0], np.float64)
    one, eps32, thre32 = np.float32(1.0), np.float32(eps), np.float32(thre)
    for b, n in pi:
        a32 = alpha[b:b + n]
        a64 = a32.astype(np.float64)
        w64 = np.zeros(n, np.float64)
        T32, T64 = one, 1.0
        for j in range(n):                               # forward
            if T32 < eps32:
                break
            if a32[j] <= thre32:
                continue
            w64[j] = a64[j] * T64
            T32 = np.float32(T32 * np.float32(one - a32[j])); T64 *= 1.0 - a64[j]
        accum = float(np.dot(gw64[b:b + n], w64))
        T32, T64 = one, 1.0
        for j in range(n):                               # backward
            if T32 < eps32:
                break
            if a32[j] < thre32:
                continue
            ga[b + j] = (gw64[b + j] * T64 - accum) / max(1.0 - a64[j], 1e-10)
            accum -= gw64[b + j] * w64[j]
            T32 = np.float32(T32 * np.float32(one - a32[j])); T64 *= 1.0 - a64[j]
    return ga


@pytest.fixture(params=["scan", "serial"])
def composite_mode(request, hip_option):
    """the fused composite's two transmittance forms: wave prefix products (default up to 16384 packs; T within rounding
    of the serial recurrence, the early-stop cut identical) and the serial replay (option pack_scan = 0: vw bit-identical to
    packed_alpha_to_vw; wave-per-pack below 2048 packs, lane-per-pack from there)"""
    hip_option("pack_scan", 1 if request.param == "scan" else 0)
    return request.param


def _composite_case(oracle, dev, P, pi, S, seed, eps, thre, normalize, with_rgb, scatter, mode="serial"):
    import nr3d_lib_amd.graphics.pack_ops as po
    rng = np.random.default_rng(seed)
    n_packs = pi.shape[0]
    alpha = (0.99 * rng.uniform(0, 1, S) ** 2).astype(np.float32)        # alpha == 1 makes the reference's own backward
    alpha[rng.random(S) < 0.1] = 0.0                                     # ill-conditioned (division by max(1-a, 1e-10))
    t = np.sort(rng.uniform(0.5, 6.0, S)).astype(np.float32)
    rgb = rng.uniform(0, 1, (S, 3)).astype(np.float32) if with_rgb else None
    num_rays = n_packs + 37 if scatter else n_packs
    hit = np.sort(rng.choice(num_rays, n_packs, replace=False)).astype(np.int64) if scatter else None
    vw_r, mask_r, depth_r, col_r = _chain_forward(oracle, alpha, t, rgb, pi, eps, thre, normalize)
    a_t = T(alpha, dev).requires_grad_(True)
    t_t = T(t, dev).requires_grad_(True)
    c_t = T(rgb, dev).requires_grad_(True) if with_rgb else None
    vw, mask, depth, col = po.packed_composite(a_t, t_t, c_t, T(pi, dev), T(hit, dev) if scatter else None, num_rays,
                                               early_stop_eps=eps, alpha_thre=thre, normalize_depth=normalize)
    sel = (lambda a: a[torch.from_numpy(hit).to(dev)]) if scatter else (lambda a: a)
    if mode == "serial":
        assert_equal(vw, vw_r, "vw")                                      # same serial transmittance chain
    else:                                                                 # tree-ordered products: same cut, values to rounding
        assert np.array_equal(vw.detach().cpu().numpy() == 0, vw_r == 0), "vw: early-stop / threshold cut differs"
        assert_close(vw, vw_r, name="vw")
    assert_close(sel(mask), mask_r, name="mask")
    assert_close(sel(depth), depth_r, name="depth")
    if with_rgb:
        assert_close(sel(col), col_r, name="rgb")
    if scatter:                                                           # rays that are not hit stay zero
        miss = torch.ones(num_rays, dtype=torch.bool, device=dev)
        miss[torch.from_numpy(hit).to(dev)] = False
        assert not mask[miss].any() and not depth[miss].any()
    g_mask = rng.standard_normal(n_packs).astype(np.float32)
    g_depth = rng.standard_normal(n_packs).astype(np.float32)
    g_rgb = rng.standard_normal((n_packs, 3)).astype(np.float32) if with_rgb else None
    g_vw = (0.1 * rng.standard_normal(S)).astype(np.float32)
    full = lambda g: (torch.zeros((num_rays,) + g.shape[1:], device=dev).index_copy_(0, torch.from_numpy(hit).to(dev), T(g, dev))
                      if scatter else T(g, dev))
    loss = (vw * T(g_vw, dev)).sum() + (mask * full(g_mask)).sum() + (depth * full(g_depth)).sum()
    if with_rgb:
        loss = loss + (col * full(g_rgb)).sum()
    grads = torch.autograd.grad(loss, [a_t, t_t] + ([c_t] if with_rgb else []))
    ga_r, gt_r, gc_r, gw64 = _chain_backward(oracle, alpha, t, rgb, pi, eps, thre, normalize, vw_r, mask_r, g_mask, g_depth, g_rgb, g_vw)
    # grad_alpha_j = (gw_j T_j - sum_{k>=j} gw_k w_k) / max(1 - alpha_j, 1e-10): the division amplifies the fp32 rounding of
    # the numerator by up to 1 / (1 - alpha) = 100 here -- in the reference's fp32 chain just as much as in the fused kernel,
    # which forms the same gw_j in a different (fma) order.  Round 5: the reference for it is the same formula in FLOAT64
    # (_alpha_to_vw_backward_f64), which has no rounding of its own to add: the fused kernel is held to 2e-4 of max|grad_alpha|
    # against it (it was 2e-3 against the fp32 chain); the fp32 chain itself is checked
    # against the same reference, so a looser bound could not hide behind the chain's error.
    one_minus = np.maximum(1.0 - alpha.astype(np.float64), 1e-10)
    ga64 = _alpha_to_vw_backward_f64(alpha, gw64, pi, eps, thre)
    assert_close(ga_r.astype(np.float64), ga64, rel=2e-4, name="fp32 chain grad_alpha vs float64")
    # (numerator: a running fp32 sum over up to 512 samples of a pack -- 2e-5 of its scale, as it was against the fp32 chain)
    assert_close(grads[0].double().cpu().numpy() * one_minus, ga64 * one_minus, rel=2e-5, name="grad_alpha numerator")
    assert_close(grads[0].double(), ga64, rel=2e-4, name="grad_alpha")
    assert_close(grads[1], gt_r, name="grad_t")
    if with_rgb:
        assert_close(grads[2], gc_r, name="grad_rgb")


@pytest.mark.parametrize("n_packs,hi", [(257, 300), (2500, 90)])           # wave-per-pack / lane-per-pack kernels
@pytest.mark.parametrize("normalize,with_rgb,scatter", [(True, True, True), (False, True, False), (True, False, False)])
def test_fused_composite_against_chain(oracle, dev, P, n_packs, hi, normalize, with_rgb, scatter, composite_mode):
    rng = np.random.default_rng(41)
    pi, S = random_packs(rng, n_packs, 0, hi, 0.1)
    _composite_case(oracle, dev, P, pi, S, 5, 1e-4, 0.0, normalize, with_rgb, scatter, composite_mode)
    _composite_case(oracle, dev, P, pi, S, 6, 1e-2, 0.02, normalize, with_rgb, scatter, composite_mode)


def test_composite_scan_stop_decisions(oracle, dev, P, hip_option):
    """packs built so that the transmittance crosses early_stop_eps within rounding distance: constant alpha with
    (1 - alpha)^k == eps to a few ulp.  The prefix-product kernels must cut exactly the samples the serial recurrence cuts
    (they replay such a pack serially) -- then vw is not merely close, it is identical."""
    import nr3d_lib_amd.graphics.pack_ops as po
    hip_option("pack_scan", 1)
    rng = np.random.default_rng(3)
    n_packs, L = 600, 200
    pi = np.stack([np.arange(n_packs) * L, np.full(n_packs, L)], 1).astype(np.int64)
    eps = np.float32(1e-3)
    ks = rng.integers(20, 150, n_packs)
    alpha = np.empty((n_packs, L), np.float32)
    for p in range(n_packs):
        base = 1.0 - float(eps) ** (1.0 / ks[p])                        # (1 - base)^k == eps
        alpha[p] = np.float32(base) + np.float32(rng.integers(-2, 3)) * np.spacing(np.float32(base))
    alpha = alpha.reshape(-1)
    t = np.sort(rng.uniform(0.5, 6.0, alpha.size)).astype(np.float32)
    vw_r, mask_r, depth_r, _ = _chain_forward(oracle, alpha, t, None, pi, float(eps), 0.0, True)
    vw, mask, depth, _ = po.packed_composite(T(alpha, dev), T(t, dev), None, T(pi, dev), None, n_packs, early_stop_eps=float(eps),
                                             alpha_thre=0.0, normalize_depth=True)
    cut_r = (vw_r.reshape(n_packs, L) == 0).sum(1)
    assert cut_r.min() > 0 and len(set(cut_r.tolist())) > 20              # the stop really happens, at many different lengths
    assert np.array_equal(vw.cpu().numpy() == 0, vw_r == 0)
    assert_close(vw, vw_r, name="vw")
    assert_close(mask, mask_r, name="mask")


def test_c3_composite_shape(oracle, dev, P, composite_mode):
    """BASELINE configs[2]'s composite half at its stated shape: 4096 packs x <= 512 samples (~1.1 M samples), unfused ops
    bit-exact against the oracle and the fused composite against the chain"""
    rng = np.random.default_rng(7)
    pi, S = random_packs(rng, 4096, 0, 512, 0.02)
    alpha = (1 - np.exp(-10.0 * rng.random(S) * (2 * 3 ** 0.5 / 512))).astype(np.float32)      # sigma * delta of configs[2]
    alpha[rng.random(S) < 0.05] = np.float32(0.9)
    w, _, _ = P.packed_alpha_to_vw_forward(T(alpha, dev), T(pi, dev), 1e-4, 0.0, False)
    rw, _, _ = oracle.packed_alpha_to_vw_forward(alpha, pi, 1e-4, 0.0, False)
    assert_equal(w, rw, "weights")
    gw = rng.standard_normal(S).astype(np.float32)
    ga = P.packed_alpha_to_vw_backward(T(rw, dev), T(gw, dev), T(alpha, dev), T(pi, dev), 1e-4, 0.0)
    assert_equal(ga, oracle.packed_alpha_to_vw_backward(rw, gw, alpha, pi, 1e-4, 0.0), "grad_alphas")
    assert_close(P.packed_sum(T(rw, dev), T(pi, dev)), oracle.packed_sum(rw, pi), name="packed_sum")
    _composite_case(oracle, dev, P, pi, S, 8, 1e-4, 0.0, True, True, True, composite_mode)


# ------------------------------------------------------------------------------------------------
# ray-query glue (csrc/ray_glue.hip): fused pruning and sigma -> alpha against the op chains they replace
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_packs,hi,empty", [(1, 40, 0.0), (37, 150, 0.1), (5000, 70, 0.3), (40000, 12, 0.2), (9, 0, 0.0)])
def test_fused_compression_against_chain(oracle, dev, P, n_packs, hi, empty):
    """packed_volume_render_compression(+ gathers): one scan + one compaction pass inside the library against the
    reference's chain (compaction-mode alpha_to_vw -> nonzero x2 -> index gathers); integers bit-exact, and the oracle's
    selector as the third opinion.  40 000 packs take the three-launch scan, the others the single-workgroup one."""
    import nr3d_lib_amd.graphics.pack_ops.pack_ops as po
    rng = np.random.default_rng(n_packs + hi)
    pi, S = random_packs(rng, n_packs, 0, hi, empty)
    alpha = (rng.random(S) ** 2).astype(np.float32)
    alpha[rng.random(S) < 0.1] = 0.0
    depths, deltas = rng.random(S).astype(np.float32), rng.random(S).astype(np.float32)
    samples = rng.random((S, 3)).astype(np.float32)
    sidx = rng.integers(0, 1 << 40, S).astype(np.int64)
    tag = rng.integers(0, 1 << 40, n_packs).astype(np.int64)
    eps, thre = 1e-2, 0.05
    a_t, pi_t = T(alpha, dev), T(pi, dev)
    outs = {}
    for fused in (True, False):
        po.FUSED_COMPRESSION = fused
        try:
            outs[fused] = po.packed_volume_render_compression(a_t, pi_t, eps, thre)
        finally:
            po.FUSED_COMPRESSION = True
    for a, b in zip(outs[True], outs[False]):
        assert a.dtype == b.dtype == torch.int64 and torch.equal(a, b)
    nidx, cpi, pidx = outs[True]
    _, _, sel_ref = oracle.packed_alpha_to_vw_forward(alpha, pi, eps, thre, True)
    assert_equal(pidx, np.nonzero(sel_ref)[0], "kept samples")
    nidx2, cpi2, kept = po.packed_volume_render_compression_gather(a_t, pi_t, eps, thre, pack_tag=T(tag, dev), depths=T(depths, dev),
                                                                   deltas=T(deltas, dev), samples=T(samples, dev), sample_idx=T(sidx, dev))
    assert torch.equal(nidx2, nidx) and torch.equal(cpi2, cpi)
    k = pidx.cpu().numpy()
    assert_equal(kept['pack_tag'], tag[nidx.cpu().numpy()], "pack tags")
    assert_equal(kept['sample_idx'], sidx[k], "sample_idx")
    for name, src in (("depths", depths), ("deltas", deltas), ("samples", samples)):
        assert np.array_equal(kept[name].cpu().numpy(), src[k]), name
    if S:
        assert int(cpi[:, 1].sum()) == len(k) and (cpi[:, 1] > 0).all()


def test_sigma_delta_to_alpha(dev):
    from nr3d_lib_amd.graphics.nerf.nerf_utils import sigma_delta_to_alpha, tau_to_alpha
    g = torch.Generator().manual_seed(3)
    sigma = (torch.rand(10007, generator=g) * 40).to(dev).requires_grad_(True)
    delta = (torch.rand(10007, generator=g) * 0.1).to(dev)
    up = torch.randn(10007, generator=g).to(dev)
    a = sigma_delta_to_alpha(sigma, delta)
    a.backward(up)
    s2 = sigma.detach().clone().requires_grad_(True)
    a_ref = tau_to_alpha(s2 * delta)
    a_ref.backward(up)
    assert (a - a_ref).abs().max() <= 2e-7                       # one exp each, <= 1 ulp of values in [0, 1]
    assert (sigma.grad - s2.grad).abs().max() <= 1e-6 * s2.grad.abs().max()
    with torch.no_grad():
        assert torch.equal(sigma_delta_to_alpha(sigma, delta), a.detach())
    assert sigma_delta_to_alpha(sigma[:0], delta[:0]).numel() == 0
    # shapes that do not match fall back to the plain expression
    assert torch.allclose(sigma_delta_to_alpha(sigma.detach().view(-1, 1), delta.view(-1, 1)), a.detach().view(-1, 1), atol=2e-7)


def test_sigma_delta_to_alpha_double_backward(dev):
    """create_graph=True through the fused sigma -> alpha (round-3 advisor finding: the reference expression
    `tau_to_alpha(sigma * deltas)` supports double backward, the fused Function used to be once_differentiable): the
    second-order gradient of sum(alpha * up) wrt sigma, via a first gradient built with create_graph, against eager"""
    from nr3d_lib_amd.graphics.nerf.nerf_utils import sigma_delta_to_alpha, tau_to_alpha
    g = torch.Generator().manual_seed(5)
    delta = (torch.rand(4099, generator=g) * 0.1).to(dev)
    up = torch.randn(4099, generator=g).to(dev)
    v = torch.randn(4099, generator=g).to(dev)
    outs = []
    for f in (lambda s: sigma_delta_to_alpha(s, delta), lambda s: tau_to_alpha(s * delta)):
        sigma = (torch.rand(4099, generator=torch.Generator().manual_seed(6)) * 40).to(dev).requires_grad_(True)
        (g1,) = torch.autograd.grad((f(sigma) * up).sum(), sigma, create_graph=True)
        (g2,) = torch.autograd.grad((g1 * v).sum(), sigma)
        outs.append((g1.detach(), g2))
    assert (outs[0][0] - outs[1][0]).abs().max() <= 1e-6 * outs[1][0].abs().max()
    assert (outs[0][1] - outs[1][1]).abs().max() <= 1e-6 * outs[1][1].abs().max()
