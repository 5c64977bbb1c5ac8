"""forward-kernel experiments: one subprocess per setting, the kernels' HIP-event time over 20 launches at 2^20 (and 2^22)
points + a digest of y / dy_dx.
   python tools/exp_fwd_variants.py 0 20,22 NR3D_FWD_DBG=<bits>     timing decomposition of k_fwd_pairlane (bit 0 no stores,
                                                                    bit 1 no gathers, bit 2 no x loads; results wrong by design)
NR3D_FWD_VARIANT selected among the experimental kernels of commit a3cfea3 (profiles/r03a_fwd_experiments.txt); the
library now holds only the kernel that came out of them, so the first argument is kept for the log format only.

Needs the experiments build of the library (round 4: measurement knobs are compiled out of the production library):
    make -C nr3d_lib_amd/csrc clean && make -C nr3d_lib_amd/csrc -j8 EXTRA=-DNR3D_EXPERIMENTS"""
import os, sys, subprocess, json, hashlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child(log2n):
    import torch
    from nr3d_lib_amd import _hip as H
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    dev = torch.device("cuda", 0)
    cfg = gen_ngp_cfg()
    meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    N = (1 << log2n) if log2n < 64 else log2n
    gen = torch.Generator(device="cpu").manual_seed(42)
    params = torch.empty(meta.n_params).uniform_(-1e-2, 1e-2, generator=gen).to(dev)
    x = torch.rand(N, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
    for _ in range(3):
        y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
    torch.cuda.synchronize()
    names = ("lotd_fwd", "lotd_fwd_lds")
    for k in names: H.prof_read(k)
    H.prof_enable(*names)
    iters = 20
    for _ in range(iters):
        y, j = _lotd.lod_fwd(meta, x, params, need_input_grad=True)
    torch.cuda.synchronize()
    H.prof_enable()
    us = {k: H.prof_read(k)[0] / iters * 1e3 for k in names}
    dig = hashlib.sha256(y.contiguous().cpu().numpy().tobytes() + j.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
    v = os.environ.get("NR3D_FWD_VARIANT", "0")
    err = None
    if log2n == 20:
        ref = f"/tmp/exp_fwd_ref_{log2n}.pt"
        if v == "0":
            torch.save((y.cpu(), j.cpu()), ref)
        elif os.path.exists(ref):
            y0, j0 = torch.load(ref)
            err = (float(((y.cpu() - y0).abs().amax(0) / y0.abs().amax(0)).max()),
                   float(((j.cpu() - j0).abs().amax((0, 2)) / j0.abs().amax((0, 2))).max()))
    print(json.dumps(dict(us_pairlane=round(us["lotd_fwd"], 1), us_lds=round(us["lotd_fwd_lds"], 1), digest=dig,
                          max_rel_err_vs_v0=err)))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2])); sys.exit(0)
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "3", "4", "5", "6", "7"]
    sizes = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["20", "22"])]
    extra_env = dict(kv.split("=") for kv in sys.argv[3:])
    for n in sizes:
        for v in variants:
            env = dict(os.environ, NR3D_FWD_VARIANT=v, **extra_env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"2^{n} V{v} {extra_env}: {line[-1] if line else r.stderr[-300:]}", flush=True)
