"""GPU-box host probe for the CPU baseline: how many cores does this container really get, and how does the oracle scale?"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import oracle
from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
cfg = gen_ngp_cfg()
m = oracle.lotd_create_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
rng = np.random.default_rng(42); n = 1 << 17
x = rng.random((n, 3)).astype(np.float32).clip(1e-6, 1 - 1e-6)
p = rng.uniform(-1e-4, 1e-4, m.n_params).astype(np.float32)
g = (rng.standard_normal((n, m.n_encoded_dims)) / 1e4).astype(np.float32)
oracle.lotd_fwd(m, x, p, need_dydx=True); oracle.lotd_bwd_dparam(m, g, x, p, accum_double=2)
t = time.perf_counter(); y, j = oracle.lotd_fwd(m, x, p, need_dydx=True); t1 = time.perf_counter()
oracle.lotd_bwd_dx(m, g, j); t2 = time.perf_counter()
oracle.lotd_bwd_dparam(m, g, x, p, accum_double=2); t3 = time.perf_counter()
print(os.environ.get("OMP_NUM_THREADS"), "fwd %.4f dx %.4f dparam %.4f  -> %.3f Mpts/s" % (t1 - t, t2 - t1, t3 - t2, n / (t3 - t) / 1e6))
'''
for nt in (8, 16, 32, 64, 128, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PROC_BIND="false")
    subprocess.run([sys.executable, "-c", code], env=env)
