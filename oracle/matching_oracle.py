"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the matching track's candidate retrieval.

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
    """-> [(query_id, ref_id, score)], sorted by descending score (stable over first encounter)."""
    ref_id_list = []
    for ref_vf in sn_refs:                                   # :232-236
        ref_id_list.extend([ref_vf.video_id for _ in range(ref_vf.feature.shape[0])])
    bank = np.concatenate([np.asarray(r.feature, dtype=np.float32) for r in sn_refs], axis=0)
    search_res_map = {}
    k = min(len(bank), top)                                  # :242
    for vf in sn_query_list:                                 # :243
        vf_id, vf_feature = vf.video_id, np.asarray(vf.feature, dtype=np.float32)
        D, I = knn_oracle.knn_ip(vf_feature, bank, k)        # :246
        mask = D[:, k - 1] > threshold                       # :247
        if mask.sum() > 0:
            lim_remain, D_remain, I_remain = knn_oracle.range_search_ip(vf_feature[mask], bank, threshold)
        D_res, I_res = [], []
        nr = 0
        for i in range(len(vf_feature)):                     # :252-262
            if not mask[i]:
                nv = (D[i, :] > threshold).sum()
                D_res.extend(list(D[i, :nv]))
                I_res.extend(list(I[i, :nv]))
            else:
                l0, l1 = lim_remain[nr], lim_remain[nr + 1]
                D_res.extend(list(D_remain[l0:l1]))
                I_res.extend(list(I_remain[l0:l1]))
                nr += 1
        for dis, idx in zip(D_res, I_res):                   # :263-269
            recall_pair = (vf_id, ref_id_list[idx])
            if recall_pair in search_res_map:
                search_res_map[recall_pair] = max(search_res_map[recall_pair], dis)
            else:
                search_res_map[recall_pair] = dis
    search_res_list = [(qid, rid, dis) for (qid, rid), dis in search_res_map.items()]
    search_res_list.sort(key=lambda x: -x[2])                # :271
    return search_res_list


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
