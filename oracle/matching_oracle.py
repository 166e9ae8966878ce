"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the matching track's candidate retrieval and of its
segment extraction (generate_matching_result).

Follows VSC22-Matching-Track-1st/infer/infer_matching.py:229-262 statement by statement, with the
faiss flat inner-product index replaced by oracle/knn_oracle (the same stand-in the descriptor-track
oracle uses: exact fp32 ascending-k chains, faiss result layout).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this module; nothing under vsc22-submission_amd/ does.

Pinned by: knn_oracle is pinned on the reference's unit-test vectors and float64 (tests/test_oracle_knn.py);
this file adds only the reference's Python bookkeeping around it, and tests/test_matching.py checks it
against a brute-force float64 statement of "best frame score above the threshold per video pair".
"""
import numpy as np

from . import knn_oracle

SEARCH_THRESHOLD = -0.1  # infer_matching.py:62


def candidate_pairs(sn_query_list, sn_refs, threshold=SEARCH_THRESHOLD, top=1024):
    """-> [(query_id, ref_id, score)], best first (stable over first encounter, as the reference's dict + sort).

    Control flow of infer_matching.py:229-271 with knn_oracle in faiss's place: per query video a top-`top` search
    over all reference frames (:246); frames whose last hit still clears the threshold are re-searched with
    range_search (:247-249), the others keep their hits above the threshold (:253-256); every hit votes for its
    reference video and a (query video, reference video) pair keeps its best vote (:263-269)."""
    owner = []                                   # reference video id of every bank row (:232-236)
    for video in sn_refs:
        owner += [video.video_id] * video.feature.shape[0]
    bank = np.concatenate([np.asarray(video.feature, dtype=np.float32) for video in sn_refs], axis=0)
    k = min(len(bank), top)
    best = {}                                    # insertion-ordered, like the reference's search_res_map
    for video in sn_query_list:
        frames = np.asarray(video.feature, dtype=np.float32)
        top_scores, top_rows = knn_oracle.knn_ip(frames, bank, k)
        crowded = top_scores[:, k - 1] > threshold
        if crowded.any():
            lims, range_scores, range_rows = knn_oracle.range_search_ip(frames[crowded], bank, threshold)
        nth_crowded = 0
        for f in range(len(frames)):
            if crowded[f]:
                lo, hi = lims[nth_crowded], lims[nth_crowded + 1]
                hits = zip(range_scores[lo:hi], range_rows[lo:hi])
                nth_crowded += 1
            else:
                n_above = int((top_scores[f] > threshold).sum())
                hits = zip(top_scores[f, :n_above], top_rows[f, :n_above])
            for score, row in hits:
                pair = (video.video_id, owner[row])
                if pair not in best or score > best[pair]:
                    best[pair] = score
    ranked = [(qid, rid, score) for (qid, rid), score in best.items()]
    ranked.sort(key=lambda triple: -triple[2])
    return ranked


def video_pair_max(q_bank, q_video, n_q_videos, r_bank, r_video, n_r_videos, threshold):
    """Same contract as vsc_video_pair_max_f32 (CSR over query videos, ascending reference video)."""
    sim = knn_oracle.ip_matrix(q_bank, r_bank)
    table = np.full((n_q_videos, n_r_videos), -np.inf, dtype=np.float32)
    qi, ri = np.nonzero(sim > np.float32(threshold))
    np.maximum.at(table, (q_video[qi], r_video[ri]), sim[qi, ri])
    hit = np.isfinite(table)
    lims = np.concatenate([[0], np.cumsum(hit.sum(1))]).astype(np.int64)
    rows, cols = np.nonzero(hit)
    return lims, cols.astype(np.int32), table[rows, cols]


# ---- segment extraction (VSC22-Matching-Track-1st/infer/src/utils.py:80-116) -----------------------------------
def components8(mask):
    """8-connected labelling by flood fill (stands in for cv2.connectedComponentsWithStats, utils.py:88):
    -> (number of labels including background 0, label map), components numbered in raster order."""
    lab = np.zeros(mask.shape, np.int32)
    count = 0
    for r0, c0 in zip(*np.nonzero(mask)):
        if lab[r0, c0]:
            continue
        count += 1
        lab[r0, c0] = count
        todo = [(r0, c0)]
        while todo:
            r, c = todo.pop()
            for rr in range(max(r - 1, 0), min(r + 2, mask.shape[0])):
                for cc in range(max(c - 1, 0), min(c + 2, mask.shape[1])):
                    if mask[rr, cc] and not lab[rr, cc]:
                        lab[rr, cc] = count
                        todo.append((rr, cc))
    return count + 1, lab


def matching_result(res_list, threshold=0.05, std_ratio=2):
    """Follows utils.py:80-116 in its own order of operations: threshold (:86-87), components (:88), components of
    more than ten pixels are fitted one by one and leave the loose-pixel mask (:91-96), all loose pixels otherwise
    form label 1 (:97-99); per kept label a weighted RANSAC line over its pixels plus the loose ones (:101-106),
    accepted when the slope is positive and enough distinct frames lie within one frame of it (:107-113)."""
    from sklearn.linear_model import RANSACRegressor
    rows = []
    for qid, rid, prob, _unused in res_list:
        loose = prob > threshold
        n_labels, lab = components8(prob > threshold)
        kept = []
        for k in range(1, n_labels):
            pixels = lab == k
            if pixels.sum() > 10:
                kept.append(k)
                loose[pixels] = False
        if not kept:
            lab = loose.astype(np.int32)
            kept = [1]
        for k in kept:
            qs, rs = np.where((lab == k) | loose)
            if len(set(qs)) <= 3:
                continue
            fit = RANSACRegressor(max_trials=200, random_state=2023, residual_threshold=2)
            fit.fit(qs[:, None], rs[:, None], sample_weight=np.square(prob[qs, rs]))
            near = abs(rs - fit.predict(qs[:, None]).flatten()) < 1
            slope = fit.estimator_.coef_[0][0]
            if slope <= 0:
                continue
            slope = max(1 / slope, slope)
            if near.sum() > 5 and len(set(qs[near])) > 3 and len(set(rs[near])) > 3:
                top = prob[qs[near], rs[near]]
                rows.append([qid, rid, qs[near][0], rs[near][0], qs[near][-1], rs[near][-1],
                             top.max() - top.std() * std_ratio - abs(slope - 1) / 10])
    return rows
