import sys, os, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle
from util import LOTD_CASES, lotd_inputs
from nr3d_lib_amd.bindings import _lotd
from nr3d_lib_amd import _hip
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
bad = 0
cases = [c for c in LOTD_CASES if c not in ("cp_4d",)]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    case = cases[it % len(cases)]
    # half of the iterations: VM levels over sorted points whatever their size (lotd_sorted.hip), with / without k_vm_direct in front
    _hip.set_option("vm_sorted", 2 if (it // len(cases)) % 2 else -1)
    _hip.set_option("vm_direct", 0 if (it // len(cases)) % 4 >= 2 else -1)
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 511, 513, 1000, 4097, 9999]))
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    x, p, g, v = lotd_inputs(m_ref.as_dict(), n, it)
    t = lambda a: torch.from_numpy(a).to(dev)
    ml = int(rng.integers(-1, m.n_levels + 1)) if it % 3 == 0 else None
    kw = {} if ml is None else dict(max_level=ml)
    y, j = _lotd.lod_fwd(m, t(x), t(p), need_input_grad=True, **kw)
    y_ref, j_ref = oracle.lotd_fwd(m_ref, x, p, need_dydx=True, **kw)
    dx, dp = _lotd.lod_bwd(m, t(g), t(x), t(p), j, need_input_grad=True, need_param_grad=True, **kw)
    dp_ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True, **kw)
    if it % 2 == 1 and m.n_levels > 1:        # the same gradient in random level buckets (a random partition, random order)
        cuts = sorted(set(int(c) for c in rng.integers(1, m.n_levels, size=int(rng.integers(1, 4)))))
        edges = [0] + cuts + [m.n_levels]
        bk = [(a, b - 1) for a, b in zip(edges[:-1], edges[1:])]
        bk = [bk[i] for i in rng.permutation(len(bk))]
        seen = []
        _, dpb = _lotd.lod_bwd(m, t(g), t(x), t(p), None, need_input_grad=False, need_param_grad=True, level_buckets=bk,
                               on_bucket=lambda k, sl: seen.append(sl.numel()), **kw)
        eb = float((dpb - dp).abs().max()) / max(float(dp.abs().max()), 1e-30)
        if eb > 2e-6 or sum(seen) != m.n_params:
            bad += 1; print("BUCKET MISMATCH", case, n, ml, bk, eb, sum(seen), m.n_params)
    _, dp2, dx2 = _lotd.lod_bwd_bwd_input(m, t(v), t(g), t(x), t(p), j, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True, need_dLdinput_dinput=True, **kw)
    dp2_ref = oracle.lotd_bwd_bwd_dparam(m_ref, v, g, x, p, accum_double=True, **kw)
    def err(a, b):
        a = a.cpu().numpy().reshape(b.shape); s = max(np.abs(b).max(), 1e-30); return np.abs(a - b).max() / s
    if it % 4 == 2:           # half tables read by the kernels themselves against the run on the fp32 copy: the same bits
        ph, gh = t(p).half(), t(g).half()
        outs = []
        for nat in (True, False):
            _lotd.NATIVE_HALF = nat
            try:
                yh, jh = _lotd.lod_fwd(m, t(x), ph, need_input_grad=True, **kw)
                dxh, dph = _lotd.lod_bwd(m, gh, t(x), ph, jh, need_input_grad=True, need_param_grad=True, **kw)
                _, dp2h, dx2h = _lotd.lod_bwd_bwd_input(m, t(v), gh, t(x), ph, jh, need_dLdinput_ddLdoutput=False, need_dLdinput_dparams=True,
                                                        need_dLdinput_dinput=True, **kw)
            finally:
                _lotd.NATIVE_HALF = True
            outs.append((yh, jh, dxh, dp2h, dx2h) + (() if _lotd._native_half(m, ph, False) else (dph,)))
        for k, (a, b) in enumerate(zip(*outs)):
            if not torch.equal(a, b):
                bad += 1; print("HALF TABLE MISMATCH", case, n, ml, k, float((a.float() - b.float()).abs().max()))
    es = (err(y, y_ref), err(j, j_ref), err(dp, dp_ref), err(dp2, dp2_ref))
    ok = all(e <= 1e-5 for e in es)
    bad += not ok
    if not ok: print("MISMATCH", case, n, ml, es)
print("fuzz done, mismatches:", bad)
