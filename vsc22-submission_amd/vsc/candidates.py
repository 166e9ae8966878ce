"""Candidate generation: best frame-pair score per (query video, ref video)
(reference: infer/vsc/candidates.py:15-40)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List

import numpy as np

from vsc.index import PairMatches, VideoFeature, VideoIndex
from vsc.metrics import CandidatePair


class ScoreAggregation(ABC):
    @abstractmethod
    def aggregate(self, match: PairMatches) -> float:
        ...

    def score(self, match: PairMatches) -> CandidatePair:
        return CandidatePair(query_id=match.query_id, ref_id=match.ref_id, score=self.aggregate(match))


class MaxScoreAggregation(ScoreAggregation):
    def aggregate(self, match: PairMatches) -> float:
        return float(np.max([m.score for m in match.matches]))


class CandidateGeneration:
    def __init__(self, references: List[VideoFeature], aggregation: ScoreAggregation):
        self.aggregation = aggregation
        self.index = VideoIndex(references[0].dimensions())
        self.index.add(references)

    def query(self, queries: List[VideoFeature], global_k: int) -> List[CandidatePair]:
        pairs = [self.aggregation.score(m) for m in self.index.search(queries, global_k=global_k)]
        return sorted(pairs, key=lambda c: c.score, reverse=True)
