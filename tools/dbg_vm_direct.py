import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from util import LOTD_CASES, lotd_inputs
import oracle
from nr3d_lib_amd import _hip
from nr3d_lib_amd.bindings import _lotd
oracle.build()
dev = torch.device("cuda:0")
for case in ("mixed", "mixed_cuboid"):
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    x, p, g, v = lotd_inputs(m_ref.as_dict(), 4099, 1)
    xt, pt, gt = (torch.from_numpy(a).to(dev) for a in (x, p, g))
    _hip.set_option("vm_direct", 0)
    ref = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)[1].cpu().numpy()
    _hip.set_option("vm_direct", 1)
    offs = list(m.level_offsets)
    for it in range(6):
        got = _lotd.lod_bwd(m, gt, xt, pt, None, need_input_grad=False, need_param_grad=True)[1].cpu().numpy()
        bad = np.nonzero(np.abs(got - ref) > 1e-4 * np.abs(ref).max())[0]
        if bad.size:
            lv = np.searchsorted(offs, bad, side="right") - 1
            for l in np.unique(lv):
                b = bad[lv == l] - offs[l]
                F = nf[l]; R = res[l] if isinstance(res[l], (list, tuple)) else [res[l]] * 3
                print(case, "iter", it, "level", l, "F", F, "res", R, "n_lines", sum(R), "bad entries", np.unique(b // F)[:12], "feats", np.unique(b % F), "count", b.size,
                      "got", got[bad[:3]], "ref", ref[bad[:3]])
        else:
            print(case, "iter", it, "ok")
