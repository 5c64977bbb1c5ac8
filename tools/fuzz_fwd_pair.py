"""random 3-D Dense / Hash metas (2-feature pseudo levels) through the two-lane + LDS-staged forward against the oracle:
cuboid resolutions, F in {2, 4, 6, 8}, power-of-two and odd hash sizes, smoothstep, max_level, N across the LDS-staging
threshold (2^18), a level above 2^24 entries (the non-24-bit index path) now and then, half tables.
usage: python tools/fuzz_fwd_pair.py [iterations]"""
import sys, os, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle
from nr3d_lib_amd.bindings import _lotd
oracle.build(); oracle.set_num_threads(oracle.host_cores())
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(iters):
    L = int(rng.integers(1, 10))
    types, res, nf = [], [], []
    for l in range(L):
        tp = "Dense" if rng.random() < 0.5 else "Hash"
        if tp == "Dense":
            r = [int(v) for v in rng.integers(3, 40, 3)] if rng.random() < 0.5 else [int(rng.integers(3, 48))] * 3
        else:
            r = [int(v) for v in rng.integers(3, 3000, 3)] if rng.random() < 0.3 else [int(rng.integers(3, 3000))] * 3
        types.append(tp); res.append(r); nf.append(int(rng.choice([2, 2, 2, 4, 6, 8])))
    big = it % 13 == 5
    if big:                               # one Dense level above 2^24 entries: the plain 32-bit multiply path
        types[-1], res[-1], nf[-1] = "Dense", [260, 260, 260], 2
    T = int(rng.choice([2 ** int(rng.integers(4, 20)), int(rng.integers(17, 100000))]))
    smooth = bool(rng.random() < 0.3)
    n = int(rng.choice([1, 63, 1000, 4097, 100003, (1 << 18) + 5, 300001])) if not big else 4097
    m_ref = oracle.lotd_create_meta(3, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(3, res, nf, types, T, smooth)
    d = m_ref.as_dict()
    x = rng.random((n, 3), dtype=np.float32).clip(1e-6, 1 - 1e-6)
    # keep x * (R - 2) + 0.5 away from integers: a last-bit difference must not move a point into another cell
    for r in res:
        for k in range(3):
            vv = x[:, k] * np.float32(r[k] - 2) + np.float32(0.5)
            near = np.abs(vv - np.round(vv)) < 2e-3
            x[near, k] = np.clip(x[near, k] + np.float32(4e-3 / max(r[k] - 2, 1)), 1e-6, 1 - 1e-6)
    p = (rng.standard_normal(d["n_params"]) * 0.1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    ml = int(rng.integers(-1, m.n_levels + 1)) if it % 3 == 0 else None
    kw = {} if ml is None else dict(max_level=ml)
    half = it % 5 == 4 and bool(_lotd._native_half(m, t(p).half(), False))
    pt = t(p).half() if half else t(p)
    p_r = pt.float().cpu().numpy()
    y, j = _lotd.lod_fwd(m, t(x), pt, need_input_grad=True, **kw)
    y0 = _lotd.lod_fwd(m, t(x), pt, need_input_grad=False, **kw)[0]
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p_r, need_dydx=True, **kw)
    def err(a, b):        # per output column, as tests/util.py: y over the rows, the Jacobian over rows and dims
        a = a.float().cpu().numpy().reshape(b.shape)
        s = np.maximum(np.abs(b).max((0, 2) if b.ndim == 3 else 0, keepdims=True), 1e-2 if b.ndim == 2 else 1e-1)   # floor: tables ~ N(0, 0.1), a lone point may cancel to ~0
        return float((np.abs(a - b) / s).max())
    # points that still land on a cell face in one level may differ there: count rows instead of failing on one
    ey, ej = err(y, y_ref), err(j.reshape(n, -1, 3), j_ref.reshape(n, -1, 3))
    tol = 1e-3 if half else 1e-5
    ok = ey <= tol and ej <= 1e-5 and torch.equal(y, y0)
    if not ok:
        rows = (np.abs(y.float().cpu().numpy() - y_ref) > tol * np.abs(y_ref).max(0)).any(1).sum()
        bad += 1; print("MISMATCH", it, dict(L=L, types=types, res=res, nf=nf, T=T, smooth=smooth, n=n, ml=ml, half=half), ey, ej, "rows", int(rows))
print(f"fuzz_fwd_pair: {iters} configurations, mismatches: {bad}")
