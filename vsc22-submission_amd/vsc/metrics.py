"""The slice of the reference's vsc/metrics.py that the descriptor path touches:
video-id formatting, candidate pairs, predicted matches (the rows of matches.csv) and the
descriptor-track micro-AP (infer/vsc/metrics.py:21-95, 183-243, 423-455).  The matching-track
segment metric is out of scope of this path."""
from __future__ import annotations

import dataclasses
import enum
from typing import Collection, List, NamedTuple, Optional, Union

import numpy as np


class Dataset(enum.Enum):
    QUERIES = "Q"
    REFS = "R"


def format_video_id(video_id: Union[str, int], dataset: Optional[Dataset]) -> str:
    """int ids become 'Q000123' / 'R000123'; str ids are checked against `dataset`."""
    if isinstance(video_id, (int, np.integer)):
        if dataset is None:
            raise ValueError("Unable to convert integer video_id without a Dataset enum")
        return f"{dataset.value}{int(video_id):06d}"
    if not isinstance(video_id, str):
        raise AssertionError(f"unexpected video_id: {video_id} of type {type(video_id)}")
    if dataset is not None and video_id[0] != dataset.value:
        raise AssertionError(f"dataset mismatch? got {video_id} for dataset {dataset}")
    return video_id


@dataclasses.dataclass
class CandidatePair:
    query_id: str
    ref_id: str
    score: float

    @classmethod
    def to_dataframe(cls, candidates: Collection["CandidatePair"]):
        import pandas as pd
        rows = [{"query_id": format_video_id(c.query_id, Dataset.QUERIES),
                 "ref_id": format_video_id(c.ref_id, Dataset.REFS), "score": c.score}
                for c in candidates]
        return pd.DataFrame(rows, columns=["query_id", "ref_id", "score"])

    @classmethod
    def write_csv(cls, candidates, file):
        cls.to_dataframe(candidates).to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file) -> List["CandidatePair"]:
        import pandas as pd
        df = pd.read_csv(file)
        return [CandidatePair(format_video_id(q, Dataset.QUERIES), format_video_id(r, Dataset.REFS), s)
                for q, r, s in zip(df.query_id, df.ref_id, df.score)]


class Match(NamedTuple):
    """A predicted (or ground-truth) copied segment: one row of matches.csv (metrics.py:183-243)."""
    query_id: str
    ref_id: str
    score: float
    query_start: float
    query_end: float
    ref_start: float
    ref_end: float

    def pair_id(self):
        return (self.query_id, self.ref_id)

    @classmethod
    def write_csv(cls, matches: Collection["Match"], file) -> None:
        import pandas as pd
        df = pd.DataFrame([m._asdict() for m in matches], columns=cls._fields)
        df = df.loc[:, ["query_id", "ref_id", "query_start", "query_end", "ref_start", "ref_end", "score"]]
        for col in ("query_start", "query_end", "ref_start", "ref_end"):
            df[col] = df[col].astype(np.float64)
        df.to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file, is_gt: bool = False, check: bool = True) -> List["Match"]:
        import pandas as pd
        df = pd.read_csv(file)
        df["query_id"] = df.query_id.map(lambda x: format_video_id(x, Dataset.QUERIES))
        df["ref_id"] = df.ref_id.map(lambda x: format_video_id(x, Dataset.REFS))
        if is_gt:
            df["score"] = 1.0
        if check:
            for field in cls._fields:
                assert not df[field].isna().any()
        return [Match(**{f: rec[f] for f in cls._fields}) for rec in df.to_dict("records")]


def candidate_pairs_from_matches(matches: Collection[Match]) -> List[CandidatePair]:
    """Best score per (query, ref) over a list of matches (CandidatePair.from_matches, metrics.py:84-93)."""
    scores: dict = {}
    for m in matches:
        key = (m.query_id, m.ref_id)
        scores[key] = max(m.score, scores.get(key, 0.0))
    return [CandidatePair(q, r, s) for (q, r), s in scores.items()]


def micro_average_precision(ground_truth: Collection[CandidatePair],
                            predictions: Collection[CandidatePair]) -> float:
    """uAP over (query, ref) pairs: sum_i P(i) * correct(i) / |gt| with predictions in
    descending score order (metrics.py:439-451, the `simple_ap`)."""
    gt = {(p.query_id, p.ref_id) for p in ground_truth}
    if len(gt) != len(ground_truth):
        raise AssertionError("Duplicates detected in ground truth")
    seen = {(p.query_id, p.ref_id) for p in predictions}
    if len(seen) != len(predictions):
        raise AssertionError("Duplicates detected in predictions")
    ranked = sorted(predictions, key=lambda p: p.score, reverse=True)
    hit = np.array([(p.query_id, p.ref_id) in gt for p in ranked], dtype=np.float64)
    if not len(hit) or not gt:
        return 0.0
    precision = np.cumsum(hit) / (np.arange(len(hit)) + 1)
    return float(np.sum(precision * hit) / len(gt))
