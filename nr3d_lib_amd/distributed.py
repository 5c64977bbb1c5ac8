"""Multi-GPU data parallelism for the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference has no multi-GPU path for these ops (its only distributed code initialises a process group,
nr3d_lib/distributed.py:40-46,99, and the --ddp flag is deprecated, nr3d_lib/config.py:75-76).  The path shards
naturally:

  * points (LoTD) and rays (march + composite) are independent -> contiguous shards, NO data-path collective;
  * the LoTD parameters are replicated, so the one collective is the all-reduce(SUM) of dL/dparam per step -- after
    the backward, or in level buckets overlapped with it (`lotd_backward_allreduce`)
    (first- and second-order parameter gradients are summed into the same buffer by autograd before it);
  * packed offsets (packed_info / pack_infos) are shard-local; a global layout, when needed, is the local one
    shifted by an exclusive scan of the per-rank totals (one small all_gather).
"""
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank_world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_range(n: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    """[start, stop) of this rank's contiguous shard of ``n`` items; sizes differ by at most one and the
    first ``n % world`` ranks take the extra item."""
    r, w = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(t: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """this rank's contiguous slice of ``t`` along dim 0 (a view)"""
    a, b = shard_range(t.shape[0], rank, world)
    return t[a:b]


def allreduce_grads(grads: Iterable[Optional[torch.Tensor]], average: bool = False, bucket_bytes: int = 64 << 20,
                    single_rank_too: bool = False):
    """In-place all-reduce(SUM) of parameter gradients, coalesced into flat buckets of <= ``bucket_bytes``.

    xGMI is point-to-point (7 links per GPU), so a ring all-reduce is per-link bound and its latency term
    is paid per call: the LoTD gradient (46 MiB for the NGP config) goes out as ONE bucket, small tensors
    (MLP weights, ...) are packed together instead of being reduced one by one.
    ``single_rank_too``: issue the collectives on a one-rank group as well (a SUM over one rank: values unchanged) -- how the
    one-GPU bench figure and test exercise the flattening and RCCL's launch path at their real sizes; off by default, a
    one-rank job has nothing to reduce."""
    if not is_dist() or (dist.get_world_size() == 1 and not single_rank_too):
        return
    world = dist.get_world_size()
    pending, size = [], 0

    def flush():
        nonlocal pending, size
        if not pending:
            return
        if len(pending) == 1:
            flat = pending[0].view(-1)
            dist.all_reduce(flat)
            if average:
                flat.div_(world)
        else:
            flat = torch.cat([g.reshape(-1) for g in pending])
            dist.all_reduce(flat)
            if average:
                flat.div_(world)
            off = 0
            for g in pending:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        pending, size = [], 0

    for g in grads:
        if g is None:
            continue
        if not g.is_contiguous():
            raise RuntimeError("allreduce_grads: gradients must be contiguous")
        nbytes = g.numel() * g.element_size()
        if pending and (size + nbytes > bucket_bytes or pending[0].dtype != g.dtype):
            flush()
        pending.append(g)
        size += nbytes
    flush()


def lotd_level_buckets(meta, first_fraction=0.8):
    """Level buckets for ``lod_bwd(..., level_buckets=...)``, finest levels first.  ``first_fraction``: a number -> two
    buckets ``[(lo, L-1), (0, lo-1)]``, the first holding >= that fraction of the parameter bytes, so that the large part
    of the gradient is on the wire while the coarse levels (few bytes, a comparable share of the accumulation work --
    every level receives the same number of updates) are still being accumulated; a sequence of increasing fractions ->
    one more bucket per entry (cut whenever the cumulative share reaches the next fraction).
    NGP config: 0.8 -> levels 6..15 (40 of 46 MiB), then 0..5;  (0.4, 0.8) -> 11..15, 6..10, 0..5."""
    L, total = meta.n_levels, float(meta.n_params)
    fractions = [float(first_fraction)] if isinstance(first_fraction, (int, float)) else [float(f) for f in first_fraction]
    if L < 2:
        return [(0, L - 1)]
    buckets, hi, acc, k = [], L - 1, 0.0, 0
    for lvl in range(L - 1, 0, -1):              # level 0 always stays in the last bucket
        acc += meta.level_n_params[lvl]
        if k < len(fractions) and acc >= fractions[k] * total:
            buckets.append((lvl, hi))
            hi, k = lvl - 1, k + 1
            if k == len(fractions):
                break
    buckets.append((0, hi))
    return buckets


def lotd_backward_allreduce(lod_bwd, meta, *args, first_fraction=0.8, **kwargs):
    """``lod_bwd(meta, *args, **kwargs)`` with dL/dparam all-reduced (SUM) over the default group, the reduction of every
    level bucket but the last overlapped with the accumulation of the next one.  Returns (dL_dx, dL_dparam) like
    ``lod_bwd``; the gradient is complete when this returns (in stream order).  Without a process group: the plain call."""
    if not is_dist():
        return lod_bwd(meta, *args, **kwargs)
    works = []
    out = lod_bwd(meta, *args, level_buckets=lotd_level_buckets(meta, first_fraction),
                  on_bucket=lambda k, g: works.append(dist.all_reduce(g, async_op=True)), **kwargs)
    for w in works:
        w.wait()
    return out


def global_pack_offsets(local_total: int, device=None) -> Tuple[int, int]:
    """(offset of this rank's packed samples in the concatenation over ranks, global total)"""
    if not is_dist() or dist.get_world_size() == 1:
        return 0, int(local_total)
    t = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allt, t)
    totals = [int(v.item()) for v in allt]
    return sum(totals[:dist.get_rank()]), sum(totals)
