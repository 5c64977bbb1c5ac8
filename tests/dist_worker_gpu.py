"""Worker of tests/test_dist_gpu.py::test_two_ranks_on_one_gpu -- one of WORLD_SIZE processes sharing cuda:0.
RCCL refuses two ranks on one device, so the group is `gloo` (which all-reduces device tensors through host staging): what
is exercised is the path's own protocol with the REAL kernels at world size 2 -- points sharded over ranks, per-rank
lod_bwd with level buckets, every bucket's all-reduce started async while the next bucket is accumulated, and the result
compared with the oracle's gradient of the FULL batch (sum over ranks == gradient of the union) and with one plain
all-reduce after a plain backward."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from util import LOTD_CASES, assert_close, lotd_inputs  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.distributed import lotd_backward_allreduce, lotd_level_buckets, shard_range
    oracle.build()
    oracle.set_num_threads(max(1, oracle.host_cores() // world))
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    m_ref = oracle.lotd_create_meta(D, res, nf, types, T, smooth)
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    n = 1 << 15
    x, p, g, _ = lotd_inputs(m_ref.as_dict(), n, 91)              # same seed on every rank: the global batch
    lo, hi = shard_range(n, rank, world)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xt, pt, gt = t(x[lo:hi]), t(p), t(g[lo:hi])
    _, j = _lotd.lod_fwd(m, xt, pt, need_input_grad=True)
    ref = oracle.lotd_bwd_dparam(m_ref, g, x, p, accum_double=True)          # gradient of the whole batch
    dx0, dp0 = _lotd.lod_bwd(m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True)
    plain = dp0.clone()
    dist.all_reduce(plain)
    assert_close(plain, ref, name="plain backward + one all-reduce", levels=m_ref)
    for frac in (0.8, (0.3, 0.7)):
        assert len(lotd_level_buckets(m, frac)) >= 2
        for rep in range(2):
            dx1, dp1 = lotd_backward_allreduce(_lotd.lod_bwd, m, gt, xt, pt, j, need_input_grad=True, need_param_grad=True,
                                               first_fraction=frac)
            torch.cuda.synchronize()
            assert torch.equal(dx1, dx0)
            assert_close(dp1, ref, name=f"bucketed {frac} (rank {rank})", levels=m_ref)
            assert_close(dp1, plain.cpu().numpy(), rel=1e-6, name=f"bucketed {frac} vs plain", levels=m_ref)
    # every rank holds the same reduced gradient
    chk = torch.stack([dp1.double().sum(), dp1.double().abs().sum()])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(b, both[0]) for b in both)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK", flush=True)


if __name__ == "__main__":
    main()
