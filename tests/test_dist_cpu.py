"""CPU, world_size = 2 over gloo: the data-parallel plumbing of nr3d_lib_amd.distributed (shards, bucketed
gradient all-reduce, global packed offsets).  On the GPU the same code runs over RCCL (backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nr3d_lib_amd import distributed as D
        assert D.is_dist() and D.rank_world() == (rank, world)
        # shards tile [0, n) exactly, sizes differ by <= 1
        for n in (0, 1, 7, 1024, 1025):
            a, b = D.shard_range(n)
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(s[1] == t[0] for s, t in zip(spans, spans[1:]))
            assert max(s[1] - s[0] for s in spans) - min(s[1] - s[0] for s in spans) <= 1 and (a, b) == spans[rank]
        x = torch.arange(10.0)
        assert torch.equal(D.shard(x), x[slice(*D.shard_range(10))])
        # gradient all-reduce: one big tensor (its own bucket) + several small ones packed together + a None
        torch.manual_seed(rank)
        big, s1, s2 = torch.randn(5000), torch.randn(3, 4), torch.randn(7)
        want = []
        for t in (big, s1, s2):
            parts = []
            for r in range(world):
                torch.manual_seed(r)
                b_, a1, a2 = torch.randn(5000), torch.randn(3, 4), torch.randn(7)
                parts.append({5000: b_, 12: a1, 7: a2}[t.numel()])
            want.append(sum(parts))
        D.allreduce_grads([big, None, s1, s2], bucket_bytes=4096)
        for got, w in zip((big, s1, s2), want):
            torch.testing.assert_close(got, w.view_as(got))
        avg = torch.full((4,), float(rank + 1))
        D.allreduce_grads([avg], average=True)
        torch.testing.assert_close(avg, torch.full((4,), sum(range(1, world + 1)) / world))
        # sharded occupancy-grid update: every rank scatters the maxima of ITS samples, all-reduce(MAX) of the per-voxel
        # maxima, then the decay is applied once -- must equal the unsharded update (oracle restatement on the CPU; on
        # the GPU nr3d_lib_amd.models.accelerations.occgrid.update_*(..., group=...) runs the same three steps)
        import numpy as np
        import oracle
        rng = np.random.default_rng(5)
        res, n = (6, 5, 7), 600
        grid0 = rng.uniform(0, 1, res).astype(np.float32)
        gidx = np.stack([rng.integers(0, r, n) for r in res], 1)
        val = rng.uniform(-0.5, 2, n).astype(np.float32)
        a, b = D.shard_range(n)
        vmax = torch.from_numpy(oracle.occ_scatter_max(res, gidx[a:b], val[a:b]))
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        got = oracle.occ_apply_max(grid0, vmax.numpy(), 0.9)
        assert np.array_equal(got, oracle.occ_update_grid(grid0, gidx, val, 0.9))
        # ... whereas all-reducing the locally decayed grids would be wrong wherever only the other rank has samples
        local = torch.from_numpy(oracle.occ_update_grid(grid0, gidx[a:b], val[a:b], 0.9))
        dist.all_reduce(local, op=dist.ReduceOp.MAX)
        assert not np.array_equal(local.numpy(), got)
        # bucketed backward + overlapped all-reduce: the wrapper drives any lod_bwd-shaped callable; here a host stand-in
        # that honours level_buckets / on_bucket the way bindings._lotd.lod_bwd does (the HIP one needs a GPU)
        class Meta: n_levels, n_params, level_n_params, level_offsets = 4, 100, [10, 20, 30, 40], [0, 10, 30, 60, 100]
        calls = []

        def fake_bwd(meta, scale, level_buckets=None, on_bucket=None):
            grad = torch.zeros(meta.n_params)
            for k, (lo, hi) in enumerate(level_buckets or [(0, meta.n_levels - 1)]):
                a, b = meta.level_offsets[lo], meta.level_offsets[hi + 1]
                grad[a:b] = scale * torch.arange(a, b, dtype=torch.float32)
                calls.append((lo, hi))
                if on_bucket is not None:
                    on_bucket(k, grad[a:b])
            return None, grad
        assert D.lotd_level_buckets(Meta) == [(1, 3), (0, 0)] and D.lotd_level_buckets(Meta, 0.3) == [(3, 3), (0, 2)]
        assert D.lotd_level_buckets(Meta, (0.3, 0.6)) == [(3, 3), (2, 2), (0, 1)] and D.lotd_level_buckets(Meta, 1.5) == [(0, 3)]
        _, gsum = D.lotd_backward_allreduce(fake_bwd, Meta, float(rank + 1))
        assert calls == [(1, 3), (0, 0)]
        torch.testing.assert_close(gsum, sum(range(1, world + 1)) * torch.arange(100, dtype=torch.float32))
        off, total = D.global_pack_offsets(10 * (rank + 1))
        assert total == sum(10 * (r + 1) for r in range(world)) and off == sum(10 * (r + 1) for r in range(rank))
        q.put((rank, "ok"))
    except Exception as e:   # surface the failure to the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(20)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_is_a_noop():
    from nr3d_lib_amd import distributed as D
    assert D.rank_world() == (0, 1) and D.shard_range(10) == (0, 10)
    g = torch.ones(3)
    D.allreduce_grads([g])
    assert torch.equal(g, torch.ones(3)) and D.global_pack_offsets(5) == (0, 5)
    # without a process group the bucketed backward is the plain call (no level_buckets passed)
    assert D.lotd_backward_allreduce(lambda meta, a, **kw: (a, kw), None, 7) == (7, {})


def test_level_buckets_of_the_ngp_config():
    """the headline config: ten Hash levels (40 of 46 MiB) first, then the six Dense ones"""
    from nr3d_lib_amd import distributed as D
    from nr3d_lib_amd.bindings import _lotd
    from nr3d_lib_amd.models.grid_encodings.lotd import gen_ngp_cfg
    cfg = gen_ngp_cfg()
    m = _lotd.LoDMeta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"])
    b = D.lotd_level_buckets(m)
    assert b == [(6, 15), (0, 5)], b
    assert D.lotd_level_buckets(m, (0.4, 0.8)) == [(11, 15), (6, 10), (0, 5)]
    assert sum(m.level_n_params[6:]) / m.n_params > 0.8
    one = _lotd.LoDMeta(3, [16], [2], ["Dense"], None)
    assert D.lotd_level_buckets(one) == [(0, 0)]
