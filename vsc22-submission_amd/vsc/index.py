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
    """What `faiss.index_factory(dim, "Flat", METRIC_INNER_PRODUCT)` is to the reference:
    add() appends rows, search() is the exact top-k.  Rows live in HBM as one float32
    matrix (bigger, fewer allocations: sized for 288 GB)."""

    def __init__(self, dim: int, metric: int = METRIC_INNER_PRODUCT):
        if metric != METRIC_INNER_PRODUCT:
            raise NotImplementedError(
                "only METRIC_INNER_PRODUCT is on the HIP path (the descriptor track never uses L2; "
                "for unit vectors L2 ranking == inner-product ranking)")
        self.d = dim
        self.metric_type = metric
        self._chunks: list = []
        self._bank = None
        self.ntotal = 0

    def add(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d, f"expected [n,{self.d}], got {x.shape}"
        self._chunks.append(x)
        self._bank = None
        self.ntotal += x.shape[0]

    def reset(self) -> None:
        self._chunks, self._bank, self.ntotal = [], None, 0

    def device_bank(self):
        import torch
        from vsc_hip import _lib
        _lib.require_device()
        if self._bank is None:
            host = np.concatenate(self._chunks) if self._chunks else np.zeros((0, self.d), np.float32)
            self._bank = torch.from_numpy(host).cuda()
        return self._bank

    def range_search(self, x: np.ndarray, radius: float):
        """All (query row, ref row, score) with score > radius -> three flat arrays, query-major,
        ascending ref row inside a query (faiss.Index.range_search, flattened)."""
        import torch
        from vsc_hip import ops
        bank = self.device_bank()
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
        lims, D, I = ops.range_search_ip(q, bank, float(radius))
        lims = lims.cpu().numpy()
        rows = np.repeat(np.arange(len(lims) - 1), np.diff(lims))
        return rows, I.cpu().numpy(), D.cpu().numpy()

    def search(self, x: np.ndarray, k: int):
        """-> (D [nq,k] float32 descending, I [nq,k] int64), faiss.Index.search semantics."""
        import torch
        from vsc_hip import ops
        bank = self.device_bank()
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
        if k > MAX_K:
            raise NotImplementedError(f"k={k} > {MAX_K} is not supported by vsc_knn_ip_f32")
        D, I = ops.knn_ip(q, bank, k)
        return D.cpu().numpy(), I.cpu().numpy()


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
        """The reference keeps every pair above an adaptively tightened radius, sorts all of
        them by score and truncates to global_k (index.py:145-165): the result is the
        global_k best (query row, ref row) pairs over ALL pairs.  Here: per-row exact
        top-k', then the same sort/truncate on the host; if a query row could own
        more than k' of the winners, the exact range sweep at the provisional threshold
        replaces the candidate set (always exact)."""
        nr = self.index.ntotal
        kk = int(min(global_k, nr, MAX_K))
        if kk <= 0 or feats.shape[0] == 0:
            return []
        D, I = self.index.search(feats, kk)
        valid = I >= 0
        rows = np.broadcast_to(np.arange(I.shape[0])[:, None], I.shape)[valid]
        refs, scores = I[valid], D[valid]
        order = np.lexsort((refs, rows, -scores.astype(np.float64)))[:global_k]
        if len(order) == global_k and kk < min(global_k, nr):
            threshold = scores[order[-1]]
            if (D[:, kk - 1] > threshold).any():
                # some query row owns more than kk of the winners: sweep every pair scoring
                # >= threshold (the reference's radius search) and redo the sort/truncate
                rows, refs, scores = self.index.range_search(feats, np.nextafter(threshold, -np.inf))
                order = np.lexsort((refs, rows, -scores.astype(np.float64)))[:global_k]
        return [(int(rows[o]), int(refs[o]), float(scores[o])) for o in order]
