"""Video-pair candidates from frame-level search hits (interface of the reference's infer/vsc/candidates.py:15-40).

`CandidateGeneration(refs, aggregation).query(queries, global_k)` is what `sscd_baseline.search` calls: the globally
best `global_k` frame pairs come from the HIP sweep behind `VideoIndex.search`, every (query video, reference video)
pair that owns at least one of them becomes a `CandidatePair` whose score is `aggregation.aggregate(hits)`, best first.
"""
from __future__ import annotations

from typing import List, Sequence

from vsc.index import PairMatches, VideoFeature, VideoIndex
from vsc.metrics import CandidatePair


class ScoreAggregation:
    """How the frame-pair hits of one video pair collapse into the pair's score."""

    def aggregate(self, match: PairMatches) -> float:
        raise NotImplementedError(f"{type(self).__name__} must implement aggregate()")

    def score(self, match: PairMatches) -> CandidatePair:
        return CandidatePair(match.query_id, match.ref_id, self.aggregate(match))


class MaxScoreAggregation(ScoreAggregation):
    """The pair is as good as its best frame pair."""

    def aggregate(self, match: PairMatches) -> float:
        best = match.matches[0].score
        for hit in match.matches[1:]:
            if hit.score > best:
                best = hit.score
        return float(best)


class CandidateGeneration:
    def __init__(self, references: Sequence[VideoFeature], aggregation: ScoreAggregation):
        if not references:
            raise ValueError("CandidateGeneration needs at least one reference video")
        self.aggregation = aggregation
        self.index = VideoIndex(references[0].dimensions())
        self.index.add(list(references))

    def query(self, queries: List[VideoFeature], global_k: int, limit: int = None) -> List[CandidatePair]:
        """limit: only the first `limit` candidates (== query(...)[:limit]; the caller that keeps 25 per query video of 1 200 does not pay
        for a CandidatePair object per dropped pair)"""
        # The fast path equals the object path only (i) on a similarity index -- with METRIC_L2 a pair's FIRST hit in best-first order is its
        # smallest distance, while MaxScoreAggregation takes the largest, and the final sort runs the other way -- and (ii) when no two
        # query VideoFeatures share a video_id: the object path groups hits by id, the flat path by position in `queries`.
        ids = [q.video_id for q in queries]
        if type(self.aggregation) is MaxScoreAggregation and global_k >= 0 and self.index.index.is_similarity and len(set(ids)) == len(ids):
            # the aggregation the descriptor track uses (sscd_baseline.py:100): the same list as below, built from flat arrays instead
            # of a PairMatch object per frame hit (2.4M hits of 2 000 query videos: 29 s of Python; tools/micro/candidates_bench.py)
            return [CandidatePair(q, r, s) for q, r, s in zip(*self.index.search_pair_maxima(queries, global_k, limit))]
        scored = map(self.aggregation.score, self.index.search(queries, global_k=global_k))
        return sorted(scored, key=lambda pair: -pair.score)[:limit]   # stable: ties keep the index's pair order
