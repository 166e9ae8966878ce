"""The slice of the reference's vsc/metrics.py that the descriptor path touches:
video-id formatting, candidate pairs, predicted matches (the rows of matches.csv) and the
descriptor-track micro-AP (infer/vsc/metrics.py:21-119, 183-243, 423-494).  The matching-track
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


@dataclasses.dataclass
class PrecisionRecallCurve:
    """Precision / recall / score at every correct prediction (metrics.py:96-112; plotting is out of scope)."""
    precisions: np.ndarray
    recalls: np.ndarray
    scores: np.ndarray


@dataclasses.dataclass
class AveragePrecision:
    """metrics.py:115-119: `ap` is the challenge's canonical number, `simple_ap` the tie-blind sum."""
    ap: float
    pr_curve: PrecisionRecallCurve
    simple_ap: Optional[float] = None


def _tie_grouped_ap(actual: np.ndarray, scores: np.ndarray) -> float:
    """sklearn.metrics.average_precision_score(actual, scores) for binary labels: the precision / recall
    curve has ONE point per distinct score (all predictions sharing a score enter together), and
    AP = sum_n (R_n - R_{n-1}) P_n.  What drivendata_average_precision calls (metrics.py:481-485)."""
    order = np.argsort(-scores, kind="mergesort")
    s = scores[order]
    y = actual[order].astype(np.float64)
    last = np.r_[np.nonzero(np.diff(s))[0], len(s) - 1]      # last index of every tie group
    tps = np.cumsum(y)[last]
    fps = (1 + last) - tps
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def drivendata_average_precision(predicted, ground_truth) -> float:
    """Canonical AP of the challenge backend (metrics.py:458-494): tie-grouped AP over the predicted
    pairs, rescaled by predicted positives / ground-truth positives.  Takes the reference's data frames
    (columns query_id, ref_id, score / query_id, ref_id)."""
    scores = np.asarray(predicted["score"], dtype=np.float64)
    if not np.isfinite(scores).all():
        raise ValueError("Scores must be finite.")
    gt_q, gt_r = list(ground_truth["query_id"]), list(ground_truth["ref_id"])
    # a pair listed twice in the ground truth duplicates its prediction row in the reference's left merge
    weight: dict = {}
    for q, r in zip(gt_q, gt_r):
        weight[(q, r)] = weight.get((q, r), 0) + 1
    order = np.argsort(-scores, kind="mergesort")              # DataFrame.sort_values is not stable; ties are
    keys = list(zip(predicted["query_id"], predicted["ref_id"]))  # grouped below, so their order is immaterial
    rows_actual, rows_score = [], []
    for i in order:
        w = weight.get(keys[i], 0)
        for _ in range(max(w, 1)):
            rows_actual.append(1.0 if w else 0.0)
            rows_score.append(scores[i])
    actual = np.asarray(rows_actual, dtype=np.float64)
    predicted_n_pos = int(actual.sum())
    unadjusted = _tie_grouped_ap(actual, np.asarray(rows_score, dtype=np.float64)) if predicted_n_pos else 0.0
    # rows with a blank ref_id only validate query ids (metrics.py:490-491)
    actual_n_pos = int(sum(1 for r in gt_r if r is not None and r == r))
    return unadjusted * (predicted_n_pos / actual_n_pos)


def average_precision(ground_truth: Collection[CandidatePair],
                      predictions: Collection[CandidatePair]) -> AveragePrecision:
    """The descriptor-track uAP exactly as the reference reports it (metrics.py:423-455):
    `.ap` = drivendata_average_precision (tied scores grouped, rescaled), `.simple_ap` =
    sum_i P(i) correct(i) / |gt| over the descending-score list, `.pr_curve` at the correct predictions."""
    gt = {(p.query_id, p.ref_id) for p in ground_truth}
    if len(gt) != len(ground_truth):
        raise AssertionError("Duplicates detected in ground truth")
    seen = {(p.query_id, p.ref_id) for p in predictions}
    if len(seen) != len(predictions):
        raise AssertionError("Duplicates detected in predictions")
    canonical = drivendata_average_precision(
        predicted=CandidatePair.to_dataframe(predictions),
        ground_truth=CandidatePair.to_dataframe(ground_truth))
    ranked = sorted(predictions, key=lambda p: p.score, reverse=True)
    scores = np.array([p.score for p in ranked], dtype=np.float64)
    correct = np.array([(p.query_id, p.ref_id) in gt for p in ranked], dtype=bool)
    cum = np.cumsum(correct)
    precision = cum / (np.arange(len(correct)) + 1)
    recall = cum / len(gt)
    simple = float(np.sum(precision * correct) / len(gt))
    idx = np.nonzero(correct)[0]
    return AveragePrecision(ap=canonical, simple_ap=simple,
                            pr_curve=PrecisionRecallCurve(precision[idx], recall[idx], scores[idx]))


def micro_average_precision(ground_truth: Collection[CandidatePair],
                            predictions: Collection[CandidatePair]) -> float:
    """`average_precision(...).simple_ap` (kept for callers of earlier rounds); 0.0 for empty inputs."""
    if not len(predictions) or not len(ground_truth):
        return 0.0
    return float(average_precision(ground_truth, predictions).simple_ap)
