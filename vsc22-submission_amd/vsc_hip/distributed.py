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


def gather_rows_pipelined(local: torch.Tensor, always_collective: bool = False):
    """The ranks' row blocks as SEPARATE buffers arriving one after the other: after one size exchange every source rank's shard
    is broadcast on its own (async), so the caller can work on shard s while shard s + 1 is still on the wire.
    -> (shards: per source rank a tensor [n_r, dim] -- this rank's own is `local` itself --, works: per source rank the pending
    broadcast (None for an empty shard / a group of one; the OWN shard's entry is this rank's send: nobody has to wait for it before
    READING `local`, only before the buffer is released or rewritten), offsets [world + 1])."""
    rank, ws = world()
    if ws == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return [local], [None], torch.tensor([0, local.shape[0]])
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    all_n = torch.zeros(ws, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(all_n, n)
    sizes = [int(v) for v in all_n.tolist()]
    shards, works = [], []
    # (collectives are issued in the same order -- source rank order -- on every rank; a rank starts sweeping at its own shard,
    # which needs no transfer, and then walks the sources cyclically)
    for r in range(ws):
        if r == rank:
            buf = local.contiguous()
        else:
            buf = torch.empty((sizes[r],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        shards.append(buf)
        works.append(dist.broadcast(buf, src=r, async_op=True) if sizes[r] > 0 else None)
    offsets = torch.tensor([0] + sizes).cumsum(0)
    return shards, works, offsets


def merge_parts(scores, ids, k: int, merge: Optional[Callable] = None):
    """k best of the union of per-shard top-k lists [parts][nq, k] in the search's order (score descending, ties by lower id)."""
    if len(scores) == 1:
        return scores[0], ids[0]
    if merge is None:
        from . import ops
        merge = ops.knn_merge_parts
    return merge(torch.stack(scores), torch.stack(ids))


def sweep_shards(queries: torch.Tensor, shards, k: int, knn: Callable, merge: Optional[Callable] = None, carry: bool = True):
    """The bank as a sequence of shards: `shards` yields (rows [n_s, dim], id offset) -- each when it has landed --, every shard is
    swept on its own and merged into the running top-k.  -> (scores, ids), or (None, None) for an empty sequence.
    carry: from the second shard on the sweep gets the running list's k-th score as a per-query FLOOR (vsc_knn_ip_floor_f32):
    nothing below it can enter the result, so a shard's candidate lists start with a threshold instead of paying their warm-up
    appends again (eight 125k-row shards otherwise sweep at 0.72 of the whole bank's rate).  Same bits either way: the floor only
    removes entries the merge would drop (ties at the floor are kept, the search's order decides them)."""
    scores = ids = None
    for rows, off in shards:
        if rows.shape[0] == 0:
            continue
        if carry and scores is not None:
            s, i = knn(queries, rows, k, floor=scores[:, k - 1].contiguous())     # (fewer than k so far: the slot holds -FLT_MAX, no floor)
        else:
            s, i = knn(queries, rows, k)      # (a shard with fewer than k rows: the search pads with (-FLT_MAX, -1))
        i = torch.where(i >= 0, i + off, i)
        if scores is None:
            scores, ids = s, i
        else:
            scores, ids = merge_parts([scores, s], [ids, i], k, merge)
    return scores, ids


def sharded_knn(queries_local: torch.Tensor, refs_local: torch.Tensor, k: int,
                knn: Optional[Callable] = None, gather_to: Optional[int] = 0, always_collective: bool = False,
                pipelined: bool = False, merge: Optional[Callable] = None, carry: bool = True):
    """Exact top-k of every rank's queries against the union of every rank's references.

    refs_local shards are all_gathered into the full bank (ids = position in rank order);
    each rank searches its own queries; results are gathered on rank `gather_to` (None: stay
    local).  `knn(q, r, k) -> (scores, ids)` defaults to the HIP sweep; the CPU/gloo tests
    pass the oracle here -- a test hook, not a fallback: the default raises without a GPU.

    pipelined: the bank arrives as one broadcast per source rank (gather_rows_pipelined) and every shard is swept as soon as it
    has landed -- the own shard first, while all the others are still on the wire -- with its id offset and, from the second one
    on, the running k-th score as a floor (sweep_shards; carry=False: plain per-shard sweeps); the per-shard lists are merged
    into the running list (vsc_knn_merge_parts_f32; `merge` is the tests' hook).  Same results as the one-gather form, bit for bit: a shard's
    top-k holds every member of the global top-k that lives in that shard, and the merge applies the search's own order."""
    if knn is None:
        from . import ops
        knn = ops.knn_ip
    if pipelined:
        rank, ws = world()
        shards, works, offsets = gather_rows_pipelined(refs_local, always_collective)
        def landed():
            for step in range(len(shards)):
                r = (rank + step) % len(shards)
                # the own shard is swept at once: its broadcast only READS `local` (the send side), and broadcasts complete in issue
                # order 0 .. ws - 1 on the communicator -- waiting for it here made rank r's first sweep wait for the transfers of
                # ranks 0 .. r - 1, and the time is the maximum over ranks
                if works[r] is not None and r != rank:
                    works[r].wait()          # orders the caller's stream behind that broadcast only
                yield shards[r], int(offsets[r])

        scores, ids = sweep_shards(queries_local, landed(), k, knn, merge, carry)
        if rank < len(works) and works[rank] is not None:
            works[rank].wait()               # the send is complete before `refs_local` may be released or rewritten by the caller
        if scores is None:
            scores, ids = knn(queries_local, refs_local, k)     # an empty bank everywhere: the search's own empty result
    else:
        bank, _ = all_gather_rows(refs_local, always_collective)
        scores, ids = knn(queries_local, bank, k)
    if gather_to is None or (world()[1] == 1 and not always_collective):
        return scores, ids
    all_scores, _ = all_gather_rows(scores, always_collective)
    all_ids, _ = all_gather_rows(ids, always_collective)
    if world()[0] == gather_to:
        return all_scores, all_ids
    return None, None


def score_norm_bias(queries: torch.Tensor, noise: torch.Tensor, beta: float = 1.0, nk: int = 1, knn: Optional[Callable] = None) -> torch.Tensor:
    """CSLS bias of every query row against the (replicated) noise bank: -beta * mean of its nk largest <q, noise>
    (infer/vsc/baseline/score_normalization.py:95-105, 141-150) -> [nq, 1]."""
    if knn is None:
        from . import ops
        knn = ops.knn_ip
    sims, _ = knn(queries, noise, nk)
    return -beta * sims[:, :nk].mean(dim=1, keepdim=True)


def sharded_knn_score_normalized(queries_local: torch.Tensor, refs_local: torch.Tensor, noise: torch.Tensor, k: int, beta: float = 1.0,
                                 nk: int = 1, knn: Optional[Callable] = None, **kw):
    """BASELINE.json configs[3]: "global top-k + score-norm" on the sharded path.  Score normalisation is one extra dimension
    (query' = [q, bias(q)], ref' = [r, 1]: score_normalization.py:34-105): every rank computes the bias of ITS query shard against
    the noise bank -- small, replicated on every rank, no collective -- appends the constant column to ITS reference shard, and the
    sharded search runs on the widened descriptors.  -> sharded_knn's result; scores are <q, r> + bias(q)."""
    bias = score_norm_bias(queries_local, noise, beta, nk, knn)
    q2 = torch.cat([queries_local, bias.to(queries_local.dtype)], dim=1).contiguous()
    r2 = torch.cat([refs_local, torch.ones_like(refs_local[:, :1])], dim=1).contiguous()
    return sharded_knn(q2, r2, k, knn=knn, **kw)
