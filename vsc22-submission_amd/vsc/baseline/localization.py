"""Candidate pair -> copied segments (interface of the reference's infer/vsc/baseline/localization.py).

Class names, constructor arguments and the `localize` / `localize_all` / `score` hooks are the reference's, so
`sscd_baseline.localize_and_verify` reads the same.  Two things differ underneath:

  * similarity matrices: the reference multiplies `np.matmul(a, b.T)` per candidate on the host (:33-36); here the
    candidates of a `localize_all` call are laid out as one pair table over concatenated frame banks and computed by ONE
    `vsc_pair_similarity_f32` launch (`src.matching.pair_similarity_matrices`);
  * the temporal alignment model is not rebuilt: it is the reference's own VCSL code (`vcsl.vta.build_vta_model`, CPU
    graph search), imported late exactly as the reference does (:44) -- or any object with `forward_sim` passed as
    `model=`.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from vsc.index import VideoFeature
from vsc.metrics import CandidatePair, Match

Box = Tuple[int, int, int, int]   # (query frame from, ref frame from, query frame to, ref frame to)


def _load_vcsl_model(model_type: str, options: Dict):
    try:
        from vcsl.vta import build_vta_model
    except ImportError as exc:
        raise ImportError("VCSL (the reference's infer/vcsl package) is needed for temporal alignment; put it on "
                          "PYTHONPATH or pass model= with a forward_sim method") from exc
    return build_vta_model(model_type, **options)


class Localization:
    """Turns candidates into matches; subclasses provide `localize` (one candidate) or override `localize_all`."""

    def localize(self, candidate: CandidatePair) -> List[Match]:
        raise NotImplementedError

    def localize_all(self, candidates: Iterable[CandidatePair]) -> List[Match]:
        found: List[Match] = []
        for candidate in candidates:
            found += self.localize(candidate)
        return found


class LocalizationWithMetadata(Localization):
    """Keeps the videos' features and timestamps by id and serves similarity matrices from the HIP path."""

    def __init__(self, queries: Sequence[VideoFeature], refs: Sequence[VideoFeature],
                 pair_similarity: Optional[Callable] = None):
        self.queries = {video.video_id: video for video in queries}
        self.refs = {video.video_id: video for video in refs}
        self._pair_similarity = pair_similarity   # test seam of src.matching.pair_similarity_matrices

    def similarity_all(self, candidates: Sequence[CandidatePair]) -> List[np.ndarray]:
        """One [query frames, ref frames] matrix per candidate, all from a single launch."""
        from src.matching import pair_similarity_matrices
        q_feats = {c.query_id: self.queries[c.query_id].feature for c in candidates}
        r_feats = {c.ref_id: self.refs[c.ref_id].feature for c in candidates}
        table = [(c.query_id, c.ref_id, c.score) for c in candidates]
        return pair_similarity_matrices(q_feats, r_feats, table, self._pair_similarity)

    def similarity(self, candidate: CandidatePair) -> np.ndarray:
        return self.similarity_all([candidate])[0]


class VCSLLocalization(LocalizationWithMetadata):
    """Alignment by a VCSL model over (biased) similarity matrices; every box it returns becomes a `Match` whose
    times are the timestamps of the box's corner frames and whose score comes from the `score` hook."""

    def __init__(self, queries, refs, model_type, similarity_bias=0.0, model=None, pair_similarity=None, **kwargs):
        super().__init__(queries, refs, pair_similarity)
        self.model = model if model is not None else _load_vcsl_model(model_type, kwargs)
        self.similarity_bias = similarity_bias   # some alignment methods do not take negative similarities well

    def similarity_all(self, candidates):
        return [matrix + self.similarity_bias for matrix in super().similarity_all(candidates)]

    def _match(self, candidate: CandidatePair, box: Box, matrix: np.ndarray) -> Match:
        q_from, r_from, q_to, r_to = box
        query, ref = self.queries[candidate.query_id], self.refs[candidate.ref_id]
        segment = Match(candidate.query_id, candidate.ref_id, 0.0,
                        query_start=query.get_timestamps(q_from)[0], query_end=query.get_timestamps(q_to)[1],
                        ref_start=ref.get_timestamps(r_from)[0], ref_end=ref.get_timestamps(r_to)[1])
        return segment._replace(score=self.score(candidate, segment, box, matrix))

    def localize_all(self, candidates: Sequence[CandidatePair]) -> List[Match]:
        candidates = list(candidates)
        if not candidates:
            return []
        keyed = [(f"{c.query_id}-{c.ref_id}", m) for c, m in zip(candidates, self.similarity_all(candidates))]
        aligned = self.model.forward_sim(keyed)
        if len(aligned) != len(candidates):
            raise AssertionError(f"alignment model returned {len(aligned)} results for {len(candidates)} candidates")
        found: List[Match] = []
        for candidate, (key, matrix), (result_key, boxes) in zip(candidates, keyed, aligned):
            if result_key != key:
                raise AssertionError(f"alignment result {result_key!r} does not belong to candidate {key!r}")
            found += [self._match(candidate, tuple(box), matrix) for box in boxes]
        return found

    def localize(self, candidate: CandidatePair) -> List[Match]:
        return self.localize_all([candidate])

    def score(self, candidate: CandidatePair, match: Match, box: Box, similarity: np.ndarray) -> float:
        return 1.0


class VCSLLocalizationMaxSim(VCSLLocalization):
    """Score = best (unbiased) similarity inside the box, with the reference's half-open slicing."""

    def score(self, candidate, match, box, similarity) -> float:
        q_from, r_from, q_to, r_to = box
        return similarity[q_from:q_to, r_from:r_to].max() - self.similarity_bias


class VCSLLocalizationCandidateScore(VCSLLocalization):
    """Score = the candidate's own (descriptor-track) score."""

    def score(self, candidate, match, box, similarity) -> float:
        return candidate.score
