"""One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The descriptor path shards embarrassingly:
  * encode: videos/frames are split over ranks, no collective in the data path
    (the reference does the same with DistributedSampler, infer/extract_ref_feats.py:33-36);
  * search: the ONLY exchange step is assembling the reference bank -- an all_gather of each
    rank's [n_i, dim] descriptor shard over xGMI -- after which every rank sweeps its own query
    shard against the full bank (vsc_knn_ip_f32) and rank 0 concatenates the small [nq_i, k]
    results.  One big all_gather per bank (ring collectives are per-link bound on xGMI: few,
    large messages), never one per video.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of n items for `rank` (first n % world ranks get one more)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, always_collective: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Concatenate the ranks' [n_i, dim] row blocks in rank order.
    -> (bank [sum n_i, dim], offsets [world+1]).  Shards may differ in length: sizes are
    exchanged first, shards are padded to the longest for one fixed-size all_gather.
    always_collective: run the collectives even in a one-rank group (readiness tests on one GPU: the RCCL
    all_gather_into_tensor of a bank-sized buffer is then really issued)."""
    rank, ws = world()
    if ws == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return local, torch.tensor([0, local.shape[0]])
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    all_n = torch.zeros(ws, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(all_n, n)
    sizes = [int(v) for v in all_n.tolist()]
    longest = max(sizes)
    padded = local.contiguous()
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad])
    # one collective into one flat buffer: with equal shards (the usual case) that buffer IS the bank, no second copy
    flat = torch.empty((ws * longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, padded)
    if all(s == longest for s in sizes):
        bank = flat
    else:
        bank = torch.cat([flat[r * longest: r * longest + s] for r, s in enumerate(sizes)])
    offsets = torch.tensor([0] + sizes).cumsum(0)
    return bank, offsets


def sharded_knn(queries_local: torch.Tensor, refs_local: torch.Tensor, k: int,
                knn: Optional[Callable] = None, gather_to: Optional[int] = 0, always_collective: bool = False):
    """Exact top-k of every rank's queries against the union of every rank's references.

    refs_local shards are all_gathered into the full bank (ids = position in rank order);
    each rank searches its own queries; results are gathered on rank `gather_to` (None: stay
    local).  `knn(q, r, k) -> (scores, ids)` defaults to the HIP sweep; the CPU/gloo tests
    pass the oracle here -- a test hook, not a fallback: the default raises without a GPU."""
    if knn is None:
        from . import ops
        knn = ops.knn_ip
    bank, _ = all_gather_rows(refs_local, always_collective)
    scores, ids = knn(queries_local, bank, k)
    if gather_to is None or (world()[1] == 1 and not always_collective):
        return scores, ids
    all_scores, _ = all_gather_rows(scores, always_collective)
    all_ids, _ = all_gather_rows(ids, always_collective)
    if world()[0] == gather_to:
        return all_scores, all_ids
    return None, None
