#!/usr/bin/env python
"""Matching-track entry point: VSC22-Matching-Track-1st/infer/infer_matching.py `Main.run()` from the point where the
query descriptors exist, on the HIP path end to end.

The reference embeds the query videos in the same process (four SSCD backbones + PCA, :118-156) -- that part IS the
descriptor track's query pipeline (extract_query_feats.py here) -- and then, per :211-310:

  1. query score normalisation against the normalisation set                (vsc.baseline.score_normalization)
  2. candidate (query video, ref video) retrieval above SEARCH_THRESHOLD    (src.matching.search_candidate_pairs:
                                                                             one vsc_video_pair_max_f32 sweep)
  3. per-candidate similarity maps -> pair classifier (2 x MobileNetV3)     (generate_candidates_classfiy_feature,
                                                                             match_classify: vsc_pair_similarity_f32, vsc_conv2d_f32 ..)
  4. surviving candidates -> refinement nets (2 x HRNet, map + transpose)   (generate_matching_feature, match_refine)
  5. connected components + RANSAC at three thresholds -> matches csv       (generate_matching_result; host code, as in the reference)

This script runs 1-5 from descriptor files:

  python infer_matching.py --query_features outputs/test_queries.npz --norm_refs outputs/train_refs.npz \\
      --refs outputs/test_refs.npz --sn_refs outputs/test_refs_sn.npz \\
      --cls_models ckpt/submit_cls_model1.pt ckpt/submit_cls_model2.pt \\
      --refine_models ckpt/submit_match_model1.pt ckpt/submit_match_model2.pt --output outputs/matching/test_matching.csv

Model files are the reference's TorchScript checkpoints (their state dicts carry timm's parameter names) or plain
`torch.save`d state dicts.  `--query_frames video_id,frames` gives the number of frames of each query video when the
descriptor file holds several views per video (the reference's vid_feature_len_map, :155); videos it does not list get
their frame count from the timestamps (rows / views, `frames_per_video`), which must then form whole views.
"""
import argparse
import collections
import csv
import os

import numpy as np

MATCH_CLS_THRESHOLD = 0.0005          # infer_matching.py:63
MATCH_REFINE_THRESHOLD_LOW = 0.001    # :64
MATCH_REFINE_THRESHOLD_MID = 0.1      # :65
MATCH_REFINE_THRESHOLD_HIGH = 0.35    # :66


def load_state_dict(path):
    """State dict of a TorchScript module (what the reference ships) or of a plain checkpoint (tensors only)."""
    import torch
    try:
        return {k: v for k, v in torch.jit.load(path, map_location="cpu").state_dict().items()}
    except (RuntimeError, ValueError, OSError):   # not a TorchScript archive: torch.jit.load raises any of these
        obj = torch.load(path, map_location="cpu", weights_only=True)
        return obj.state_dict() if hasattr(obj, "state_dict") else obj


def frames_per_video(vf) -> int:
    """Frames of one query video in a (possibly multi-view) descriptor file: the query pipeline emits `views` copies of the
    frame sequence, each with the same timestamps (extract_query_feats.py:197-210), so the frame count is the number of
    distinct timestamp rows.  The reference carries it from extraction (infer_matching.py:155, vid_feature_len_map)."""
    ts = np.asarray(vf.timestamps)
    n = len(vf.feature)
    distinct = len(np.unique(ts.reshape(n, -1), axis=0)) if n else 0
    if distinct == 0 or n % distinct:
        raise ValueError(f"{vf.video_id}: {n} descriptors over {distinct} distinct timestamps -- pass --query_frames")
    return distinct


def query_len_map(query_list, query_frames=None) -> dict:
    """video_id -> frames per view.  --query_frames wins; only videos it does not cover are counted from their timestamps
    (which raises for files whose rows do not form whole views)."""
    query_frames = query_frames or {}
    return {vf.video_id: int(query_frames[vf.video_id]) if vf.video_id in query_frames else frames_per_video(vf)
            for vf in query_list}


def run(query_list, score_norm_refs, refs, sn_refs, cls_models, refine_models, query_frames=None, candidates_csv=None,
        device="cuda"):
    """Steps 1-5 on loaded VideoFeature lists and HIP models -> rows [query_id, ref_id, query_start, query_end, ref_start,
    ref_end, score] (the reference's output columns, :306-310)."""
    from src import matching
    from vsc.baseline.score_normalization import normalize, query_score_normalize, transform_features

    low_var_dim = matching.calclualte_low_var_dim(score_norm_refs)                               # :212
    keep_all = collections.defaultdict(lambda: 1.0)   # the matching track's normalisation has no video-score gate (:107-115 of its score_normalization.py)
    sn_query_list = query_score_normalize(query_list, score_norm_refs, keep_all, low_var_dim=low_var_dim, beta=1.5, nk=10)   # :213
    search_res_list = matching.search_candidate_pairs(sn_query_list, sn_refs, matching.SEARCH_THRESHOLD)              # :216-262
    if candidates_csv:
        with open(candidates_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["query_id", "ref_id", "score"])
            w.writerows(search_res_list)
    query_list, refs = [transform_features(x, normalize) for x in (query_list, refs)]             # :272-274
    query_map = {vf.video_id: vf.feature for vf in query_list}
    ref_map = {vf.video_id: vf.feature for vf in refs}
    len_map = query_len_map(query_list, query_frames)   # views share timestamps: best-view selection needs the frame count
    cls_feature, cls_info = matching.generate_candidates_classfiy_feature(query_map, ref_map, search_res_list, len_map)   # :277-279
    cls_rows = matching.match_classify(cls_models, cls_feature, [(q, r) for q, r, _ in cls_info], device=device)       # :280
    best = {}
    for q, r, p in cls_rows:                                                                      # groupby(query_id, ref_id).prob.max() (:281)
        best[q, r] = max(best.get((q, r), -1.0), p)
    candidate_score_list = [(q, r, p) for (q, r), p in sorted(best.items()) if p > MATCH_CLS_THRESHOLD]               # :283-284
    match_meta = matching.generate_matching_feature(query_map, ref_map, len_map, candidate_score_list)                # :285-286
    refine_res = matching.match_refine(refine_models, match_meta, device=device)                                      # :288
    found = []
    for thr, std_ratio in ((MATCH_REFINE_THRESHOLD_HIGH, 0.5), (MATCH_REFINE_THRESHOLD_MID, 1.25), (MATCH_REFINE_THRESHOLD_LOW, 2)):
        found += matching.generate_matching_result(refine_res, threshold=thr, std_ratio=std_ratio)                  # :289-291
    top = {}
    for qid, rid, qs, rs, qe, re_, score in found:                                                # groupby(...).score.max() (:294)
        key = (qid, rid, float(qs), float(rs), float(qe), float(re_))
        top[key] = max(top.get(key, -np.inf), float(score))
    return [[q, r, qs, qe, rs, re_, s] for (q, r, qs, rs, qe, re_), s in sorted(top.items())]


def main(args):
    from src import matching
    from vsc.metrics import Dataset
    from vsc.storage import load_features
    query_frames = None
    if args.query_frames:
        with open(args.query_frames, newline="") as f:
            query_frames = {r[0]: int(r[1]) for r in csv.reader(f) if r and r[0] != "video_id"}
    cls_models, refine_models = matching.load_match_models([load_state_dict(p) for p in args.cls_models],
                                                           [load_state_dict(p) for p in args.refine_models])
    rows = run(load_features(args.query_features, Dataset.QUERIES), load_features(args.norm_refs, Dataset.REFS),
               load_features(args.refs, Dataset.REFS), load_features(args.sn_refs, Dataset.REFS), cls_models, refine_models,
               query_frames, args.candidates_csv)
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    with open(args.output, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["query_id", "ref_id", "query_start", "query_end", "ref_start", "ref_end", "score"])
        w.writerows(rows)
    print(f"{len(rows)} matches -> {args.output}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--query_features", required=True)
    ap.add_argument("--norm_refs", required=True, help="NORM_DATA_FILE: descriptors of the normalisation set")
    ap.add_argument("--refs", required=True, help="REF_DATA_FILE: reference descriptors")
    ap.add_argument("--sn_refs", required=True, help="REF_NORM_DATA_FILE: score-normalised reference descriptors")
    ap.add_argument("--cls_models", nargs="+", required=True)
    ap.add_argument("--refine_models", nargs="+", required=True)
    ap.add_argument("--query_frames", default=None)
    ap.add_argument("--candidates_csv", default="match_candidates_score.csv")
    ap.add_argument("--output", required=True)
    main(ap.parse_args())
