"""per-level cost of the two-lane forward on RAY-COHERENT samples (the full loop's marched samples) and on uniformly random
points: one subprocess per pseudo level (NR3D_FWD_ONLY_LEVEL), HIP-event time of k_fwd_pairlane, y only (no Jacobian)

Needs the experiments build of the library (round 4: measurement knobs are compiled out of the production library):
    make -C nr3d_lib_amd/csrc clean && make -C nr3d_lib_amd/csrc -j8 EXTRA=-DNR3D_EXPERIMENTS"""
import os, sys, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

def child(kind):
    import torch
    from nr3d_lib_amd import _hip as H
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    dev = torch.device("cuda", 0)
    cfg = gen_ngp_cfg()
    meta = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    gen = torch.Generator(device="cpu").manual_seed(42)
    params = torch.empty(meta.n_params).uniform_(-1e-2, 1e-2, generator=gen).to(dev)
    if kind == "rays":
        import bench
        model, n, _ = bench._full_loop_setup(dev, 512)
        rays = bench.pinhole_rays if False else None
        from demo_field import pinhole_rays
        o, d, near, far = pinhole_rays(512, dev)
        m = model.accel.ray_march(o, d, near, far)
        x = ((m.samples + 1) * 0.5).clamp(1e-6, 1 - 1e-6).contiguous()
    else:
        x = torch.rand(6897732, 3, generator=gen).clamp_(1e-6, 1 - 1e-6).to(dev)
    for _ in range(3):
        _lotd.lod_fwd(meta, x, params, need_input_grad=False)
    torch.cuda.synchronize()
    H.prof_read("lotd_fwd"); H.prof_enable("lotd_fwd")
    for _ in range(10):
        _lotd.lod_fwd(meta, x, params, need_input_grad=False)
    torch.cuda.synchronize(); H.prof_enable()
    print(json.dumps(dict(n=x.shape[0], us=round(H.prof_read("lotd_fwd")[0] / 10 * 1e3, 1))))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    for kind in ("rays", "random"):
        for lv in [-1] + list(range(2, 16)):
            env = dict(os.environ, NR3D_FWD_ONLY_LEVEL=str(lv))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", kind], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"{kind} level {lv if lv >= 0 else 'all'}: {line[-1] if line else r.stderr[-300:]}", flush=True)
