"""Host-side mirror of the reference's vsc/index.py: same classes, same call shapes, but
the flat inner-product index is a float32 matrix in HBM searched by libvsc_hip.so
(vsc_knn_ip_f32) instead of faiss.  There is no CPU search path.

Reference: VSC22-Descriptor-Track-1st/infer/vsc/index.py (VideoMetadata :20, VideoFeature
:35, PairMatch :50, PairMatches :57, VideoIndex :76, search :100, _global_threshold_knn_search
:145, _knn_search :167).
"""
from __future__ import annotations

import collections
import logging
from dataclasses import dataclass
from typing import List, NamedTuple, Tuple

import numpy as np

METRIC_INNER_PRODUCT = 0   # faiss.METRIC_INNER_PRODUCT
METRIC_L2 = 1              # faiss.METRIC_L2
MAX_K = 1024               # vsc_knn_ip_f32 limit == the reference's GPU probe (exhaustive_search.py:66)


@dataclass
class VideoMetadata:
    video_id: str
    timestamps: np.ndarray  # [N] or [N,2] (start, end)

    def __len__(self):
        return self.timestamps.shape[0]

    def get_timestamps(self, idx: int) -> Tuple[float, float]:
        t = self.timestamps[idx]
        return (t, t) if self.timestamps.ndim == 1 else (t[0], t[1])


@dataclass
class VideoFeature(VideoMetadata):
    feature: np.ndarray  # [N, dim]

    def __post_init__(self):
        assert self.feature.shape[0] == len(self.timestamps), "Mismatched timestamps / feature size"

    def metadata(self) -> VideoMetadata:
        return VideoMetadata(video_id=self.video_id, timestamps=self.timestamps)

    def dimensions(self) -> int:
        return self.feature.shape[1]


class PairMatch(NamedTuple):
    query_timestamps: Tuple[float, float]
    ref_timestamps: Tuple[float, float]
    score: float


@dataclass
class PairMatches:
    query_id: str
    ref_id: str
    matches: List[PairMatch]

    def records(self):
        for m in self.matches:
            yield {"query_id": self.query_id, "ref_id": self.ref_id,
                   "query_start": m.query_timestamps[0], "query_end": m.query_timestamps[1],
                   "ref_start": m.ref_timestamps[0], "ref_end": m.ref_timestamps[1], "score": m.score}


class FlatIPBank:
    """What `faiss.index_factory(dim, "Flat", metric)` is to the reference: add() appends rows, search() is the exact
    top-k.  Rows live in HBM as one float32 matrix (bigger, fewer allocations: sized for 288 GB).

    METRIC_L2 (the reference's own unit test builds one, tests/test_index.py) runs on the same inner-product sweep:
    rows are stored as r' = [r, |r|^2, 1] and a query is presented as q' = [2q, -1, -|q|^2], so <q', r'> = -|q - r|^2
    and "largest inner product" is "smallest distance".  The sweep SELECTS; the squared distances handed back are
    recomputed on the host from the selected rows as sum((q - r)^2) in float32 (what faiss's flat L2 scan returns), so
    no cancellation error of the expanded form reaches the caller."""

    def __init__(self, dim: int, metric: int = METRIC_INNER_PRODUCT):
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise NotImplementedError(f"metric {metric}: the reference only builds inner-product and L2 flat indexes")
        self.d = dim
        self.metric_type = metric
        self._chunks: list = []
        self._bank = None
        self._bank_max_norm = None      # largest row norm of the bank (METRIC_L2 rounding bound), cached with the bank
        self.ntotal = 0

    @property
    def is_similarity(self) -> bool:
        return self.metric_type == METRIC_INNER_PRODUCT

    def add(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d, f"expected [n,{self.d}], got {x.shape}"
        self._chunks.append(x)
        self._bank = self._bank_max_norm = None
        self.ntotal += x.shape[0]

    def reset(self) -> None:
        self._chunks, self._bank, self._bank_max_norm, self.ntotal = [], None, None, 0

    def _host_rows(self) -> np.ndarray:
        if len(self._chunks) > 1:
            self._chunks = [np.concatenate(self._chunks)]
        return self._chunks[0] if self._chunks else np.zeros((0, self.d), np.float32)

    @staticmethod
    def _to_device(host: np.ndarray):
        """float32 rows -> a GPU tensor for the C ABI (there is no CPU search path: this raises without an MI355X)."""
        import torch
        from vsc_hip import _lib
        _lib.require_device()
        return torch.from_numpy(host).cuda()

    def device_bank(self):
        if self._bank is None:
            host = self._host_rows()
            if not self.is_similarity:
                sq = np.einsum("ij,ij->i", host, host, dtype=np.float32)[:, None]
                host = np.concatenate([host, sq, np.ones_like(sq)], axis=1)
            self._bank = self._to_device(host)
        return self._bank

    def _device_queries(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if not self.is_similarity:
            sq = np.einsum("ij,ij->i", x, x, dtype=np.float32)[:, None]
            x = np.concatenate([2.0 * x, -np.ones_like(sq), -sq], axis=1)
        return self._to_device(np.ascontiguousarray(x))

    def _exact_l2(self, x: np.ndarray, rows: np.ndarray, ids: np.ndarray) -> np.ndarray:
        """float32 sum((x[rows] - bank[ids])^2); ids < 0 (padding) -> FLT_MAX as faiss does."""
        bank = self._host_rows()
        x = np.asarray(x, np.float32)
        out = np.full(ids.shape, np.finfo(np.float32).max, np.float32)
        step = max(1, (64 << 20) // max(4 * self.d, 1))          # gathered rows of one piece stay within ~64 MiB per operand
        for lo in range(0, len(ids), step):
            sl = slice(lo, lo + step)
            ok = ids[sl] >= 0
            diff = x[rows[sl][ok]] - bank[ids[sl][ok]]
            out[sl][ok] = np.einsum("ij,ij->i", diff, diff, dtype=np.float32)
        return out

    def range_count(self, x: np.ndarray, radius: float, q_dev=None) -> int:
        """Number of pairs the sweep of range_search(x, radius) holds (one sweep, nothing materialised).  q_dev: the
        queries already on the device (_device_queries(x)) when several radii are counted for one query set.  For
        METRIC_L2 this is the count of the expanded-form sweep, which may differ from the exact-distance count by the
        pairs within its rounding of the radius (range_search re-filters them)."""
        from vsc_hip import ops
        radius = float(radius) if self.is_similarity else -float(radius)
        return ops.range_count_ip(self._device_queries(x) if q_dev is None else q_dev, self.device_bank(), radius)

    def _l2_sweep_slack(self, x: np.ndarray) -> float:
        """Bound on |expanded form - exact squared distance| in the fp32 sweep: the chain 2 q.r - |r|^2 - |q|^2 has d + 2
        terms, each fmaf rounds to 2^-24 of a partial sum that never exceeds (|q| + |r|)^2."""
        if self._bank_max_norm is None:                 # once per bank, not once per call (add() / reset() invalidate it)
            host = self._host_rows()
            self._bank_max_norm = float(np.sqrt(np.einsum("ij,ij->i", host, host, dtype=np.float64).max())) if len(host) else 0.0
        qn = float(np.sqrt(np.einsum("ij,ij->i", x, x, dtype=np.float64).max())) if len(x) else 0.0
        return (self.d + 3) * 2.0 ** -23 * (qn + self._bank_max_norm) ** 2

    def range_search(self, x: np.ndarray, radius: float):
        """All (query row, ref row, score) with score > radius (inner product) / distance < radius (L2) -> three flat
        arrays, query-major, ascending ref row inside a query (faiss.Index.range_search, flattened)."""
        from vsc_hip import ops
        bank = self.device_bank()
        q = self._device_queries(x)
        if self.is_similarity:
            sweep_radius = float(radius)
        else:
            # the sweep tests the expanded form, whose cancellation on un-normalised vectors can move a pair across the
            # radius: sweep a radius widened by the rounding bound, then keep dist < radius on the recomputed distances
            x = np.ascontiguousarray(x, dtype=np.float32)
            sweep_radius = -float(np.nextafter(np.float32(float(radius) + self._l2_sweep_slack(x)), np.float32(np.inf)))
        lims, D, I = ops.range_search_ip(q, bank, sweep_radius)
        lims = lims.cpu().numpy()
        rows = np.repeat(np.arange(len(lims) - 1), np.diff(lims))
        ids = I.cpu().numpy()
        if self.is_similarity:
            return rows, ids, D.cpu().numpy()
        dist = self._exact_l2(x, rows, ids)
        keep = dist < np.float32(radius)
        return rows[keep], ids[keep], dist[keep]

    def search(self, x: np.ndarray, k: int):
        """-> (D [nq,k] float32, I [nq,k] int64), faiss.Index.search semantics: inner products descending, or squared
        L2 distances ascending."""
        from vsc_hip import ops
        bank = self.device_bank()
        q = self._device_queries(x)
        if k > MAX_K:
            raise NotImplementedError(f"k={k} > {MAX_K} is not supported by vsc_knn_ip_f32")
        if self.is_similarity:
            D, I = ops.knn_ip(q, bank, k)
            return D.cpu().numpy(), I.cpu().numpy()
        # the sweep ranks by the expanded form, whose rounding can swap near-equal neighbours: probe a few ranks past k,
        # order by the exact distances handed back, then cut (lexsort: ties keep the lower id, as the IP path does)
        kk = int(min(max(k + 8, 2 * k), MAX_K, max(self.ntotal, k)))
        _, I = ops.knn_ip(q, bank, kk)
        I = I.cpu().numpy()
        rows = np.broadcast_to(np.arange(I.shape[0])[:, None], I.shape)
        dist = self._exact_l2(x, rows.reshape(-1), I.reshape(-1)).reshape(I.shape)
        ids_key = np.where(I >= 0, I, np.iinfo(np.int64).max)
        order = np.lexsort((ids_key, dist), axis=1)[:, :k]
        return np.take_along_axis(dist, order, 1), np.take_along_axis(I, order, 1)


def _best_first(scores: np.ndarray, want: int, descending: bool) -> np.ndarray:
    """Indices of the `want` best scores, best first, equal scores in their order of appearance: a stable sort -- of host bookkeeping,
    not of the search -- on the device where there is one (20M pairs: milliseconds), numpy's otherwise (the host-logic tests)."""
    import torch
    if len(scores) == 0:
        return np.zeros(0, np.int64)
    if not torch.cuda.is_available():
        return np.argsort(-scores if descending else scores, kind="stable")[:want]
    t = torch.from_numpy(np.ascontiguousarray(scores)).cuda()
    return torch.sort(t, descending=descending, stable=True).indices[:want].cpu().numpy()


class VideoIndex:
    def __init__(self, dim: int, codec_str: str = "Flat", metric: int = METRIC_INNER_PRODUCT):
        if codec_str != "Flat":
            raise NotImplementedError(f"codec {codec_str!r}: the descriptor track only uses 'Flat'")
        self.dim = dim
        self.index = FlatIPBank(dim, metric)
        self.video_clip_idx: list = []
        self.video_clip_to_video_ids: list = []
        self.video_metadata: dict = {}

    def add(self, db: List[VideoFeature]) -> None:
        for vf in db:
            n = vf.feature.shape[0]
            self.video_clip_idx.extend(range(n))
            self.video_clip_to_video_ids.extend([vf.video_id] * n)
            self.video_metadata[vf.video_id] = vf.metadata()
            self.index.add(vf.feature)

    def search(self, queries: List[VideoFeature], global_k: int) -> List[PairMatches]:
        query_ids, query_rows = [], []
        for q in queries:
            query_ids.extend([q.video_id] * len(q))
            query_rows.extend(range(len(q)))
        query_meta = {q.video_id: q.metadata() for q in queries}
        feats = np.concatenate([q.feature for q in queries])
        if global_k < 0:
            logging.warning("Using local k for KNN search: against the VSC rules, provided for comparison.")
            hits = self._knn_search(feats, -global_k)
        else:
            hits = self._global_threshold_knn_search(feats, global_k)
        return self.pair_matches(hits, query_ids, query_rows, query_meta)

    def pair_matches(self, hits, query_ids, query_rows, query_meta) -> List[PairMatches]:
        """(query row, ref row, score) triples -> per video pair frame matches."""
        grouped = collections.defaultdict(list)
        for i, j, score in hits:
            qid, rid = query_ids[i], self.video_clip_to_video_ids[j]
            grouped[qid, rid].append(PairMatch(
                query_timestamps=query_meta[qid].get_timestamps(query_rows[i]),
                ref_timestamps=self.video_metadata[rid].get_timestamps(self.video_clip_idx[j]),
                score=score))
        return [PairMatches(qid, rid, m) for (qid, rid), m in grouped.items()]

    def _knn_search(self, feats: np.ndarray, k: int):
        D, I = self.index.search(feats, k)
        return [(i, int(I[i, j]), float(D[i, j])) for i in range(I.shape[0]) for j in range(I.shape[1])
                if I[i, j] >= 0]

    def _global_threshold_knn_search(self, feats: np.ndarray, global_k: int):
        rows, refs, scores = self._global_threshold_hits(feats, global_k)
        return [(int(i), int(j), float(sc)) for i, j, sc in zip(rows.tolist(), refs.tolist(), scores.tolist())]

    def search_pair_maxima(self, queries: List[VideoFeature], global_k: int, limit: int = None):
        """The video pairs that own at least one of the global_k best frame pairs, each with its BEST frame score, best pair first --
        what `CandidateGeneration.query` makes of `search()` under `MaxScoreAggregation` (reference candidates.py:15-40), without a
        Python object per frame hit: the hits stay three flat arrays, and because they are sorted best first a video pair's first hit is
        its maximum and the order of first appearance is the order of the maxima (ties in the order of the hit list, as the stable
        sort over the grouped hits keeps them).  -> (query ids, ref ids, scores): lists of equal length (the first `limit` pairs)."""
        if global_k < 0:
            raise ValueError("search_pair_maxima is the global-threshold form (global_k >= 0)")
        if not queries:
            return [], [], []
        feats = np.concatenate([q.feature for q in queries])
        rows, refs, scores = self._global_threshold_hits(feats, global_k)
        if len(rows) == 0:
            return [], [], []
        q_of_row = np.repeat(np.arange(len(queries)), [len(q) for q in queries])
        r_names, r_of_row = np.unique(np.asarray(self.video_clip_to_video_ids), return_inverse=True)
        key = q_of_row[rows].astype(np.int64) * len(r_names) + r_of_row[refs]
        _, first = np.unique(key, return_index=True)         # first = index of every pair's first (= best) hit
        first.sort()
        first = first[:limit]
        qv, rv = q_of_row[rows[first]], r_of_row[refs[first]]
        return [queries[i].video_id for i in qv.tolist()], r_names[rv].tolist(), scores[first].tolist()

    def _global_threshold_hits(self, feats: np.ndarray, global_k: int):
        """-> (query rows, ref rows, scores): the min(global_k, nq * nr) best frame pairs, best first (ties: lower query row, then lower
        ref row).  The reference keeps every pair inside an adaptively tightened radius, sorts all of them by score and
        truncates to global_k (index.py:145-165): the result is the min(global_k, nq * nr) best (query row, ref row)
        pairs over ALL pairs.  Here: per-row exact top-k' (k' <= MAX_K), the same sort/truncate on the host, and
        whenever the probe cannot be shown to contain every winner -- a query row may own more than k' of them, or the
        probe holds fewer than global_k pairs in total -- an exact range sweep at a radius found by counting replaces
        the candidate set (since round 5 the probe is as small as the winners-per-row allow, not the fixed 1024).  Exact for inner-product indexes; for METRIC_L2 the distances handed back are recomputed exactly and the
        range sweep is widened by its rounding bound and re-filtered (FlatIPBank.range_search)."""
        sim = self.index.is_similarity
        nr, nq = self.index.ntotal, feats.shape[0]
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32))
        kk = int(min(global_k, nr, MAX_K))
        if kk <= 0 or nq == 0:
            return empty
        want = int(min(global_k, nq * nr))
        # The probe only has to hold `want` pairs and to show where the threshold lies; rows that own more winners than it holds are
        # caught below and answered by a range sweep.  A probe of twice the mean number of winners per row (a power of two, >= 16) keeps
        # the top-k on its fast path -- the reference's fixed 1024 (exhaustive_search.py:66) is the exact fp32 sweep here, 15 x slower,
        # and hands nq x 1024 pairs to the host where 2 want / nq suffice.
        mean = -(-want // nq)
        kk = int(min(kk, max(16, 1 << (2 * mean - 1).bit_length())))
        D, I = self.index.search(feats, kk)
        valid = I >= 0
        rows = np.broadcast_to(np.arange(I.shape[0])[:, None], I.shape)[valid]
        refs, scores = I[valid], D[valid]
        # best first, ties by (query row, ref row): both hit lists below arrive query-major with equal scores of a row in ascending
        # ref order (the search's own tie rule / the range sweep's CSR order), so a STABLE sort by score alone is that order -- on
        # the GPU: a three-key lexsort of 20M probe pairs on the host was 6 of the 8 s this function took at 160k x 1M frames
        order = _best_first(scores, want, sim)
        if kk < min(global_k, nr):   # the probe was capped: rows may hold winners beyond their k'-th hit
            radius = None
            if len(order) == want:
                threshold = scores[order[-1]]
                beyond = D[:, kk - 1] > threshold if sim else D[:, kk - 1] < threshold
                if beyond.any():
                    # some query row owns more than kk of the winners: every pair at least as good as the
                    # provisional threshold (one float32 step outwards: the sweep's comparison is strict)
                    radius = np.nextafter(np.float32(threshold), np.float32(-np.inf if sim else np.inf))
                    # (a small probe can put that threshold well outside the true one when a few rows own very many winners: if the
                    # sweep would return far more than asked for, find a tighter radius by counting first)
                    if self.index.range_count(feats, radius) > 2 * want + (1 << 20):
                        radius = self._radius_for(feats, want, D[:, kk - 1])
            else:
                radius = self._radius_for(feats, want, D[:, kk - 1])
            if radius is not None:
                rows, refs, scores = self.index.range_search(feats, radius)
                if not sim:
                    # _radius_for counted pairs of the expanded-form sweep; range_search re-filters on exact distances, so
                    # borderline pairs can drop out again: widen by the rounding bound until `want` pairs are inside
                    slack = max(self.index._l2_sweep_slack(np.asarray(feats, np.float32)), np.finfo(np.float32).tiny)
                    for _ in range(40):
                        if len(rows) >= want or not np.isfinite(np.float32(radius + slack)):
                            break
                        radius, slack = float(radius) + slack, 2.0 * slack
                        rows, refs, scores = self.index.range_search(feats, radius)
                order = _best_first(scores, want, sim)
        return rows[order], refs[order], scores[order]

    def _radius_for(self, feats: np.ndarray, want: int, kth_scores: np.ndarray) -> float:
        """A radius whose range sweep returns at least `want` pairs and, ties permitting, at most 2 * want (the
        reference's min_results / max_results window, index.py:150-156), found by count-only sweeps: step outwards
        from a radius known to hold fewer than `want` pairs until enough are inside, then bisect.  kth_scores: every
        query row's k'-th (last) probe score.  The start is the BEST of them over the rows: a pair beating it beats its
        own row's k'-th hit, so it is inside the probe, which holds fewer than `want` pairs -- starting from the worst
        probe score instead would admit unprobed pairs of rows whose k'-th hit beats it, and the window is lost."""
        sim = self.index.is_similarity
        out = -1.0 if sim else 1.0                      # direction of "looser"
        loose = np.float32(-3.0e38 if sim else 3.0e38)  # finite stand-in for the reference's -1e10 / 1e10 start
        if feats.shape[0] * self.index.ntotal <= 2 * want:
            return float(loose)
        tight = float(kth_scores.max() if sim else kth_scores.min())
        q_dev = self.index._device_queries(feats)       # uploaded once for every count below
        count = lambda r: self.index.range_count(feats, r, q_dev)
        step = max(abs(tight), 1.0) * 2.0 ** -6
        lo = tight                                      # known: count(lo) < want (the probe found every pair inside it)
        hi = None
        for _ in range(64):
            cand = lo + out * step
            if not np.isfinite(np.float32(cand)):
                return float(loose)
            if count(cand) >= want:
                hi = cand
                break
            lo, step = cand, step * 2.0
        if hi is None:
            return float(loose)
        for _ in range(48):                             # bisect [lo (too few), hi (enough)] until the count fits the window
            n_hi = count(hi)
            mid = 0.5 * (lo + hi)
            if n_hi <= 2 * want or np.float32(mid) in (np.float32(lo), np.float32(hi)):
                break
            if count(mid) >= want:
                hi = mid
            else:
                lo = mid
        return float(hi)
