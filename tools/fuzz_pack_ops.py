"""Randomised check of the packed (segmented) ops against the CPU oracle: random pack layouts (empty packs, 1 .. 70 000
packs, lane-per-pack and wave-per-pack regimes), feature widths and dtypes.  usage: fuzz_pack_ops.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("NR3D_POISON_EMPTY", "1")                   # every empty() output NaN-filled: a row a kernel misses shows up
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                   # noqa: E402  (test infrastructure)
from nr3d_lib_amd.bindings import _pack_ops as P                 # noqa: E402
from nr3d_lib_amd import _hip                                  # noqa: E402

dev = torch.device("cuda:0")


def packs(rng):
    n_packs = int(rng.choice([1, 2, 63, 64, 65, 500, 2047, 2048, 5000, 70000]))
    hi = int(rng.choice([1, 4, 33, 130, 700])) if n_packs < 5000 else int(rng.choice([1, 4, 20]))
    n = rng.integers(0 if rng.random() < 0.5 else 1, hi + 1, n_packs).astype(np.int64)
    if n.sum() == 0:
        n[0] = 1
    if rng.random() < 0.4:
        # ordered packs with rows in front of, between and behind them (the kernels zero those when the tensor is tagged)
        gap = rng.integers(0, 3, n_packs) * (rng.random(n_packs) < 0.3)
        begin = int(rng.integers(0, 5)) + np.cumsum(n + gap) - n
        return np.ascontiguousarray(np.stack([begin, n], 1)), int(begin[-1] + n[-1] + rng.integers(0, 70))
    cs = np.cumsum(n)
    return np.ascontiguousarray(np.stack([cs - n, n], 1)), int(cs[-1])


def close(got, want, name, tol=1e-5, exact=False, atol=0.0):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    if exact or want.dtype.kind in "iub":
        assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} of {want.size} differ"
    else:
        sc = max(float(np.abs(want).max()) if want.size else 0.0, 1e-30)
        e = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
        assert e <= tol * sc + atol, f"{name}: err {e:.2e} vs scale {sc:.2e}"


def one(rng):
    pi, S = packs(rng)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pit = t(pi)
    if rng.random() < 0.6:
        _hip.mark_ordered(pit)                                  # as a producer of this library would hand it over: no zero-fill launch
    fd = int(rng.choice([0, 1, 3, 4]))
    shape = (S,) if fd == 0 else (S, fd)
    f = rng.standard_normal(shape).astype(np.float32)
    tag = f"P={pi.shape[0]} S={S} fd={fd}"
    # atol: a lone short pack whose N(0, 1) values nearly cancel has no scale of its own (seed 4: one pack of 24 values summing to
    # 3e-3, 2e-7 apart: summation order) -- the yardstick is the magnitude of what is summed
    close(P.packed_sum(t(f), pit), oracle.packed_sum(f, pi), "sum " + tag, 2e-5, atol=1e-6 * float(np.abs(f).max() if f.size else 0.0))
    ex, rev = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    close(P.packed_cumsum(t(f), pit, ex, rev), oracle.packed_cumsum(f, pi, ex, rev), f"cumsum ex={ex} rev={rev} " + tag, 5e-5)
    small = (0.5 + rng.random(shape)).astype(np.float32)
    close(P.packed_cumprod(t(small), pit, ex, rev), oracle.packed_cumprod(small, pi, ex, rev), f"cumprod ex={ex} rev={rev} " + tag, 1e-4)
    close(P.packed_diff(t(f), pit), oracle.packed_diff(f, pi), "diff " + tag, exact=True)
    close(P.packed_backward_diff(t(f), pit), oracle.packed_backward_diff(f, pi), "bdiff " + tag, exact=True)
    other = rng.standard_normal((pi.shape[0],) + shape[1:]).astype(np.float32) + 2.0
    for op in ("add", "mul", "div", "gt"):
        close(getattr(P, "packed_" + op)(t(f), t(other), pit), oracle.packed_binary(op, f, other, pi), op + " " + tag, exact=True)
    # alpha -> weights, forward + backward + compaction
    a = rng.random(S).astype(np.float32) * float(rng.choice([0.05, 0.5, 1.0]))
    eps, thre = float(rng.choice([1e-4, 1e-2])), float(rng.choice([0.0, 0.01]))
    w_ref = oracle.packed_alpha_to_vw_forward(a, pi, eps, thre, False)[0]
    close(P.packed_alpha_to_vw_forward(t(a), pit, eps, thre, False)[0], w_ref, "alpha fwd " + tag, exact=True)
    _, cpi_ref, sel_ref = oracle.packed_alpha_to_vw_forward(a, pi, eps, thre, True)
    _, cpi, sel = P.packed_alpha_to_vw_forward(t(a), pit, eps, thre, True)
    close(cpi, cpi_ref, "alpha compact pack_infos " + tag, exact=True)
    close(sel, sel_ref, "alpha selector " + tag, exact=True)
    gw = rng.standard_normal(S).astype(np.float32)
    close(P.packed_alpha_to_vw_backward(t(w_ref), t(gw), t(a), pit, eps, thre),
          oracle.packed_alpha_to_vw_backward(w_ref, gw, a, pi, eps, thre), "alpha bwd " + tag, exact=True)
    # fused composite: prefix-product kernels (default) against the serial replay (option pack_scan = 0: vw bit-identical to
    # alpha_to_vw) -- same early-stop cut, values to rounding -- and against the oracle's weights
    if rng.random() < 0.5:                                  # opaque and near-opaque samples drive T through eps quickly
        a = a.copy(); a[rng.random(S) < 0.05] = np.float32(rng.choice([1.0, 0.999, 0.9]))
        w_ref = oracle.packed_alpha_to_vw_forward(a, pi, eps, thre, False)[0]
    tm = np.sort(rng.random(S)).astype(np.float32) + 0.5
    rgb = rng.random((S, 3)).astype(np.float32)
    outs = {}
    for mode in ("1", "0"):
        _hip.set_option("pack_scan", int(mode))
        vw, mask, depth, col = P.packed_composite_forward(t(a), t(tm), t(rgb), pit, None, pi.shape[0], eps, thre, True)
        gm, gd, gc = (rng.standard_normal(pi.shape[0]).astype(np.float32), rng.standard_normal(pi.shape[0]).astype(np.float32),
                      rng.standard_normal((pi.shape[0], 3)).astype(np.float32))
        if mode == "1":
            g_out = [t(g_) for g_ in (gm, gd, gc)]
        ga, gt_, gc_ = P.packed_composite_backward(t(a), vw, t(tm), t(rgb), pit, None, eps, thre, True, mask, depth, g_out[0], g_out[1],
                                                   g_out[2], None)
        outs[mode] = (vw, mask, depth, col, ga, gt_, gc_)
    _hip.set_option("pack_scan", -1)
    # backward: the division by max(1 - alpha, 1e-10) amplifies rounding, so the numerators are compared
    om = np.maximum(1.0 - a.astype(np.float64), 1e-10)
    close(outs["1"][4].double().cpu().numpy() * om, outs["0"][4].double().cpu().numpy() * om, "composite scan vs serial dalpha numerator " + tag, 1e-4,
          atol=5e-7)     # "gradient still to come" = total - prefix (scan) vs a running difference (serial): both carry ~1e-7 of
                         # the pack's TOTAL, which a short pack whose numerators are all small (late, faint samples) shows
    close(outs["1"][5], outs["0"][5].cpu().numpy(), "composite scan vs serial dt " + tag, 2e-5)
    close(outs["1"][6], outs["0"][6].cpu().numpy(), "composite scan vs serial drgb " + tag, 2e-5)
    close(outs["0"][0], w_ref, "composite serial vw " + tag, exact=True)
    assert np.array_equal(outs["1"][0].cpu().numpy() == 0, w_ref == 0), "composite scan: cut differs " + tag
    close(outs["1"][0], w_ref, "composite scan vw " + tag, 1e-5)
    for k, nm in ((1, "mask"), (2, "depth"), (3, "rgb")):
        close(outs["1"][k], outs["0"][k].cpu().numpy(), f"composite scan vs serial {nm} " + tag, 2e-5)
    # sorted bins: searchsorted + inverse CDF on the non-empty packs
    if pi.shape[0] <= 5000:
        nz = pi[pi[:, 1] > 1]
        if len(nz):
            bins = np.zeros(S, np.float32)                              # (packs may leave rows out: each pack's bins at its own rows)
            for j, (b0, k) in enumerate(pi):
                bins[b0:b0 + k] = np.sort(rng.random(int(k)).astype(np.float32)) + j
            vals = (rng.random((pi.shape[0], 5)).astype(np.float32) * 1.2 - 0.1) + np.arange(pi.shape[0], dtype=np.float32)[:, None]
            close(P.packed_searchsorted(t(bins), t(vals), pit), oracle.packed_searchsorted(bins, vals, pi), "searchsorted " + tag, exact=True)
    b = rng.integers(0, 3, S).cumsum().astype(np.int64)
    close(P.mark_pack_boundaries_cuda(t(b)), oracle.mark_pack_boundaries(b), "boundaries " + tag, exact=True)
    return S


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n, total, t0 = 0, 0, time.time()
    while time.time() - t0 < budget:
        total += one(rng)
        n += 1
    print({"pack layouts ok": n, "elements": total})


if __name__ == "__main__":
    main()
