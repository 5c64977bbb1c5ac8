"""debug: sorted-points dL/dparam against the record path on a small mixed meta -- where do they differ?"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from nr3d_lib_amd import _hip
from nr3d_lib_amd.bindings import _lotd
dev = torch.device('cuda:0')
from util import LOTD_CASES
D, res, nf, types, T, smooth = LOTD_CASES[sys.argv[1] if len(sys.argv) > 1 else "mixed"]
m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
print(res, nf, types)
rng = np.random.default_rng(0)
n = 40013
x = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).to(dev)
p = torch.from_numpy(rng.standard_normal(m.n_params).astype(np.float32)).to(dev)
g = torch.from_numpy(rng.standard_normal((n, m.n_encoded_dims)).astype(np.float32)).to(dev)
_hip.set_option("vm_direct", 0)
out = {}
for mode in (0, 2):
    _hip.set_option("vm_sorted", mode)
    out[mode] = _lotd.lod_bwd(m, g, x, p, None, need_input_grad=False, need_param_grad=True)[1].cpu().numpy()
d = np.abs(out[2] - out[0])
print("n_params", m.n_params, "max diff", d.max(), "at", int(d.argmax()))
bad = np.nonzero(d > 1e-4 * np.abs(out[0]).max())[0]
print("bad count", bad.size, "first", bad[:40])
off = 0
for lv, (r, f, ty) in enumerate(zip(res, nf, types)):
    pass
import ctypes
cm = m._cmeta()
for lv in range(m.n_levels):
    L = cm.levels[lv]
    lo, hi = L.offset, L.offset + L.size * L.n_feats
    sel = bad[(bad >= lo) & (bad < hi)]
    print("level", lv, types[lv], "res", list(L.res)[:3], "F", L.n_feats, "elements", lo, hi, "bad", sel.size, sel[:12] - lo)
L = cm.levels[2]
for i in bad[:24]:
    e, f = divmod(int(i) - L.offset, L.n_feats)
    print(int(i), "entry", e, "feat", f, "records", out[0][i], "sorted", out[2][i], "diff", out[2][i] - out[0][i])
