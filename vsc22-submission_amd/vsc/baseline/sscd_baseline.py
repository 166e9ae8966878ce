#!/usr/bin/env python3
"""Descriptor-track evaluation entry point (reference: infer/vsc/baseline/sscd_baseline.py,
run by infer/eval.sh as `python3 -m vsc.baseline.sscd_baseline --query_features ...`).

On the HIP path: optional score normalisation, the exhaustive candidate search
(`search`, sscd_baseline.py:89-103 -> CandidateGeneration.query) and, with ground truth, the
descriptor-track micro-AP (:219-224).  `localize_and_verify` (:107-152) follows as in the reference when its
VCSL package is importable: the per-candidate similarity matrices come from one HIP launch per batch
(vsc.baseline.localization), the temporal alignment is the reference's own CPU code; without VCSL only
candidates.csv is written (in the reference's format) and matches.csv is skipped with a log line.
"""
from __future__ import annotations

import argparse
import logging
import os
from typing import List

from vsc.baseline.score_normalization import score_normalize
from vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc.index import VideoFeature
from vsc.metrics import CandidatePair, Dataset, Match, average_precision, micro_average_precision  # noqa: F401
from vsc.storage import load_features, store_features

logger = logging.getLogger("sscd_baseline.py")


def search(queries: List[VideoFeature], refs: List[VideoFeature], retrieve_per_query: float = 1200.0,
           candidates_per_query: float = 25.0) -> List[CandidatePair]:
    cg = CandidateGeneration(refs, MaxScoreAggregation())
    candidates = cg.query(queries, global_k=int(retrieve_per_query * len(queries)), limit=int(candidates_per_query * len(queries)))
    logger.info("Got %d candidates", len(candidates))
    return candidates


def localize_and_verify(queries: List[VideoFeature], refs: List[VideoFeature], candidates: List[CandidatePair],
                        localize_per_query: float = 5.0, score_normalization: bool = False, model=None) -> List[Match]:
    """sscd_baseline.py:107-152: the best `localize_per_query * len(queries)` candidates, aligned in batches of 512."""
    from vsc.baseline.localization import VCSLLocalizationCandidateScore, VCSLLocalizationMaxSim
    candidates = candidates[: int(len(queries) * localize_per_query)]
    if score_normalization:
        alignment = VCSLLocalizationMaxSim(queries, refs, model_type="TN", tn_max_step=5, min_length=4, concurrency=16,
                                           similarity_bias=0.5, model=model)
    else:
        from vsc.baseline.score_normalization import normalize_videos      # row-wise, a block of videos per device round trip
        alignment = VCSLLocalizationCandidateScore(normalize_videos(queries), normalize_videos(refs), model_type="TN",
                                                   tn_max_step=5, min_length=4, concurrency=16, model=model)
    matches: List[Match] = []
    logger.info("Aligning %s candidate pairs", len(candidates))
    for i in range(0, len(candidates), 512):
        matches.extend(alignment.localize_all(candidates[i:i + 512]))
        logger.info("Aligned %d pairs of %d; %d predictions so far", min(i + 512, len(candidates)), len(candidates),
                    len(matches))
    return matches


def read_ground_truth_pairs(path: str) -> List[CandidatePair]:
    """(query_id, ref_id) pairs of a matching ground-truth csv (Match.read_csv + from_matches)."""
    import pandas as pd
    from vsc.metrics import format_video_id
    df = pd.read_csv(path)
    pairs = {(format_video_id(q, Dataset.QUERIES), format_video_id(r, Dataset.REFS))
             for q, r in zip(df.query_id, df.ref_id)}
    return [CandidatePair(q, r, 1.0) for q, r in sorted(pairs)]


def main(args) -> None:
    if os.path.exists(args.output_path) and not args.overwrite:
        raise Exception(f"Output path already exists: {args.output_path}. Do you want to --overwrite?")
    queries = load_features(args.query_features, Dataset.QUERIES)
    refs = load_features(args.ref_features, Dataset.REFS)
    os.makedirs(args.output_path, exist_ok=True)
    if args.score_norm_features:
        queries, refs = score_normalize(queries, refs, load_features(args.score_norm_features, Dataset.REFS),
                                        beta=1.2)
        store_features(os.path.join(args.output_path, "sn_queries.npz"), queries)
        store_features(os.path.join(args.output_path, "sn_refs.npz"), refs)
    candidates = search(queries, refs)
    candidate_file = os.path.join(args.output_path, "candidates.csv")
    CandidatePair.write_csv(candidates, candidate_file)
    logger.info("Candidates: %s", candidate_file)
    try:
        matches = localize_and_verify(queries, refs, candidates, score_normalization=bool(args.score_norm_features))
    except ImportError as exc:
        logger.warning("matches.csv not written: %s", exc)
    else:
        matches_file = os.path.join(args.output_path, "matches.csv")
        Match.write_csv(matches, matches_file)
        logger.info("Matches: %s", matches_file)
    if args.ground_truth:
        # the reference logs the canonical `.ap` (tied scores grouped), sscd_baseline.py:213-219
        uap = average_precision(read_ground_truth_pairs(args.ground_truth), candidates)
        logger.info("Candidate uAP: %.4f", uap.ap)
        print(f"Candidate uAP: {uap.ap:.4f}")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("--query_features", required=True)
    ap.add_argument("--ref_features", required=True)
    ap.add_argument("--score_norm_features")
    ap.add_argument("--output_path", required=True)
    ap.add_argument("--ground_truth")
    ap.add_argument("--overwrite", action="store_true")
    return ap


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO)
    main(build_parser().parse_args())
