"""Candidate pair -> copied segments (reference: infer/vsc/baseline/localization.py).

Same classes and arguments as the reference.  What changes is where the per-candidate similarity matrices come
from: the reference multiplies `np.matmul(a, b.T)` per candidate on the host (localization.py:33-36), here all
candidates of a `localize_all` call go through ONE `vsc_pair_similarity_f32` launch.  The temporal alignment itself
is the reference's own VCSL code (`vcsl.vta.build_vta_model`, networkx / numba on the CPU): imported late exactly
as the reference does, or passed in as `model=` (anything with `forward_sim`); it is not rebuilt here.
"""
from __future__ import annotations

import abc
from typing import Callable, List, Optional

import numpy as np

from vsc.index import VideoFeature
from vsc.metrics import CandidatePair, Match


class Localization(abc.ABC):
    @abc.abstractmethod
    def localize(self, candidate: CandidatePair) -> List[Match]:
        pass

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        matches = []
        for candidate in candidates:
            matches.extend(self.localize(candidate))
        return matches


class LocalizationWithMetadata(Localization):
    def __init__(self, queries: List[VideoFeature], refs: List[VideoFeature], pair_similarity: Optional[Callable] = None):
        self.queries = {m.video_id: m for m in queries}
        self.refs = {m.video_id: m for m in refs}
        self._pair_similarity = pair_similarity   # test seam, see src.matching.pair_similarity_matrices

    def similarity_all(self, candidates: List[CandidatePair]) -> List[np.ndarray]:
        """[query frames, ref frames] matrix of every candidate, one launch."""
        from src.matching import pair_similarity_matrices
        query = {c.query_id: self.queries[c.query_id].feature for c in candidates}
        ref = {c.ref_id: self.refs[c.ref_id].feature for c in candidates}
        return pair_similarity_matrices(query, ref, [(c.query_id, c.ref_id, c.score) for c in candidates],
                                        self._pair_similarity)

    def similarity(self, candidate: CandidatePair) -> np.ndarray:
        return self.similarity_all([candidate])[0]


class VCSLLocalization(LocalizationWithMetadata):
    def __init__(self, queries, refs, model_type, similarity_bias=0.0, model=None, pair_similarity=None, **kwargs):
        super().__init__(queries, refs, pair_similarity)
        if model is None:
            try:
                from vcsl.vta import build_vta_model   # late import, as in the reference (:44)
            except ImportError as exc:
                raise ImportError("VCSL (the reference's infer/vcsl package) is needed for temporal alignment; put it on "
                                  "PYTHONPATH or pass model= with a forward_sim method") from exc
            model = build_vta_model(model_type, **kwargs)
        self.model = model
        self.similarity_bias = similarity_bias

    def similarity_all(self, candidates):
        """With the optional bias: some alignment methods do not tolerate negative values well."""
        return [s + self.similarity_bias for s in super().similarity_all(candidates)]

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        if not candidates:
            return []
        sims = [(f"{c.query_id}-{c.ref_id}", s) for c, s in zip(candidates, self.similarity_all(candidates))]
        results = self.model.forward_sim(sims)
        assert len(results) == len(candidates)
        matches = []
        for candidate, (key, sim), result in zip(candidates, sims, results):
            query, ref = self.queries[candidate.query_id], self.refs[candidate.ref_id]
            assert key == result[0]
            for box in result[1]:
                x1, y1, x2, y2 = box
                match = Match(query_id=candidate.query_id, ref_id=candidate.ref_id, score=0.0,
                              query_start=query.get_timestamps(x1)[0], query_end=query.get_timestamps(x2)[1],
                              ref_start=ref.get_timestamps(y1)[0], ref_end=ref.get_timestamps(y2)[1])
                matches.append(match._replace(score=self.score(candidate, match, box, sim)))
        return matches

    def localize(self, candidate: CandidatePair) -> List[Match]:
        return self.localize_all([candidate])

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        return 1.0


class VCSLLocalizationMaxSim(VCSLLocalization):
    def score(self, candidate, match, box, similarity) -> float:
        x1, y1, x2, y2 = box
        return similarity[x1:x2, y1:y2].max() - self.similarity_bias


class VCSLLocalizationCandidateScore(VCSLLocalization):
    def score(self, candidate, match, box, similarity) -> float:
        return candidate.score
