"""GPU: the one collective of the path -- the RCCL all-reduce of dL/dparam -- exercised with the REAL kernels on a
one-rank `nccl` process group (one GPU is what a gpurun box has; the N > 1 protocol is covered on CPU with gloo,
tests/test_dist_cpu.py).  What this pins:
  * `lotd_backward_allreduce` (level buckets, every bucket's all-reduce started with async_op=True on RCCL's stream while
    the next bucket is binned and accumulated on the compute stream) returns the same gradient as the plain call --
    i.e. a bucket's kernels never write outside their own slice of dL/dparam while an earlier slice is on the wire;
  * two- and three-bucket splits and the single post-backward all-reduce agree;
  * `allreduce_grads` (flat buckets over LoTD tables + small decoder tensors) leaves a one-rank gradient unchanged."""
import os

import numpy as np
import pytest
import torch

from util import LOTD_CASES, assert_close, lotd_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group(dev):
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=dev)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ngp_pair", "mixed"])
def test_bucketed_allreduce_with_the_real_kernels(oracle, dev, nccl_group, case):
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.distributed import allreduce_grads, lotd_backward_allreduce, lotd_level_buckets
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    n = 1 << 16
    x, p, g, _ = lotd_inputs(m_ref.as_dict(), n, 77)
    t = lambda a: torch.from_numpy(a).to(dev)
    xt, pt, gt = t(x), t(p), t(g)
    _, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    dx0, dp0 = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)
    assert_close(dp0, ref, name="plain dL_dparam", levels=m_ref)
    for frac in (0.8, (0.4, 0.8), (0.2, 0.5, 0.9)):
        assert len(lotd_level_buckets(m, frac)) >= 2
        for rep in range(3):                                   # repeated: a race between the streams would not be stable
            dx1, dp1 = lotd_backward_allreduce(_lotd.lod_bwd, m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True,
                                               first_fraction=frac)
            torch.cuda.synchronize()
            assert torch.equal(dx1, dx0)
            assert_close(dp1, ref, name=f"bucketed {frac} dL_dparam", levels=m_ref)
            assert_close(dp1, dp0.cpu().numpy(), rel=1e-6, name=f"bucketed {frac} vs plain", levels=m_ref)
    dp2 = dp0.clone()
    nccl_group.all_reduce(dp2)
    assert torch.equal(dp2, dp0)                               # world size 1: SUM over one rank
    small = [torch.randn(33, device=dev), None, torch.randn(5, 7, device=dev)]
    keep = [None if s is None else s.clone() for s in small]
    big = dp0.clone()
    allreduce_grads([big] + small, bucket_bytes=1 << 20)
    assert torch.equal(big, dp0) and all(a is None or torch.equal(a, b) for a, b in zip(small, keep))


@pytest.mark.parametrize("case", ["ngp_pair", "mixed"])
def test_two_ranks_on_one_gpu(case):
    """world size 2 with the real kernels: two processes share cuda:0 over a gloo group (tests/dist_worker_gpu.py)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    port = 29600 + (os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "dist_worker_gpu.py"), case], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and f"rank {r} OK" in out, f"rank {r} failed:\n{out[-3000:]}"
