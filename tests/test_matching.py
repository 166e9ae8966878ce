"""Matching-track candidate features (row (f)): host logic on CPU with the oracle behind the test seam, and the
HIP kernel bit-exact against the oracle on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn_oracle  # noqa: E402

from src import matching, synth  # noqa: E402


def _oracle_pairs(q_bank, r_bank, pairs):
    out, off = [], [0]
    for q0, qn, r0, rn in pairs:
        m = knn_oracle.ip_matrix(q_bank[q0:q0 + qn], r_bank[r0:r0 + rn]) if qn and rn else np.zeros((qn, rn), np.float32)
        out.append(m.reshape(-1))
        off.append(off[-1] + qn * rn)
    return (np.concatenate(out) if out else np.zeros(0, np.float32)), np.array(off, dtype=np.int64)


def _videos(seed, lens, d=64):
    return {f"V{seed}{i:03d}": synth.descriptor_bank(seed * 100 + i, n, d) for i, n in enumerate(lens)}


def _reference_semantics(query, ref, candidate_list, len_map):
    """utils.py:21-51 restated with float64 products (the selection must agree with the fp32 chain here)."""
    feats, infos = [], []
    for qid, rid, score in candidate_list:
        q, r = query[qid], ref[rid]
        sim = knn_oracle.ip_matrix(q, r)
        n = len_map[qid]
        if n != len(q):
            views = [np.sort(sim[s:s + n].max(1))[-10:].mean() for s in range(0, len(q), n)]
            s = int(np.argmax(views)) * n
            q = q[s:s + n]
        feats += [knn_oracle.ip_matrix(q, r), [knn_oracle.ip_matrix(r, q)]]
        infos += [[qid, rid, score]] * 2
    return feats, infos


def test_classify_features_match_reference_semantics():
    query, ref = _videos(1, [12, 36, 5]), _videos(2, [20, 7, 150])
    qids, rids = list(query), list(ref)
    len_map = {qids[0]: 12, qids[1]: 12, qids[2]: 5}  # the second query holds three 12-frame views
    cands = [(qids[1], rids[0], 0.9), (qids[0], rids[2], 0.5), (qids[1], rids[2], 0.4), (qids[2], rids[1], 0.1)]
    feats, infos = matching.generate_candidates_classfiy_feature(query, ref, cands, len_map, pair_similarity=_oracle_pairs)
    want_f, want_i = _reference_semantics(query, ref, cands, len_map)
    assert infos == want_i and len(feats) == len(want_f) == 8
    for got, want in zip(feats, want_f):
        if isinstance(want, list):
            assert isinstance(got, list) and np.array_equal(got[0], want[0])
        else:
            assert np.array_equal(got, want)
    assert feats[0].shape == (12, 20) and feats[1][0].shape == (20, 12)


def test_matching_feature_selects_view_rows():
    query, ref = _videos(3, [30]), _videos(4, [9])
    qid, rid = next(iter(query)), next(iter(ref))
    query[qid][10:20] = ref[rid][:1]  # the middle view copies a reference frame: its row maxima win
    res = matching.generate_matching_feature(query, ref, {qid: 10}, [(qid, rid, 1.0)], pair_similarity=_oracle_pairs)
    assert len(res) == 1 and res[0][0] == qid and res[0][1] == rid
    assert np.array_equal(res[0][2], query[qid][10:20]) and res[0][3] is ref[rid]


def test_empty_candidates_and_dataset_padding():
    assert matching.generate_candidates_classfiy_feature({}, {}, [], {}, pair_similarity=_oracle_pairs) == ([], [])
    big = np.arange(200 * 170, dtype=np.float32).reshape(200, 170)
    ds = matching.MatchClassifyDataset([big, [big[:3, :4]]], [["q", "r", 1.0]] * 2)
    x, qid, rid = ds[0]
    assert x.shape == (3, 160, 160) and np.array_equal(x[1], big[:160, :160]) and (qid, rid) == ("q", "r")
    y = ds[1][0]
    assert np.array_equal(y[0, :3, :4], big[:3, :4]) and y[0, 3:].sum() == 0 and y[0, :, 4:].sum() == 0


def test_low_var_dim_and_transform_features():
    from vsc.index import VideoFeature
    feats = [VideoFeature(video_id="R1", timestamps=np.arange(4.0), feature=synth.descriptor_bank(9, 4, 8))]
    feats[0].feature[:, 5] = 0.25
    assert matching.calclualte_low_var_dim(feats) == 5
    doubled = matching.transform_features(feats, lambda f: f * 2)
    assert np.array_equal(doubled[0].feature, feats[0].feature * 2) and doubled[0].video_id == "R1"


@pytest.mark.gpu
@pytest.mark.parametrize("d", [512, 100])
def test_pair_similarity_bit_exact(d):
    import torch
    from vsc_hip import ops
    q, r = synth.descriptor_bank(11, 700, d), synth.descriptor_bank(12, 900, d)
    pairs = np.array([[0, 1, 0, 1], [1, 129, 5, 300], [130, 64, 305, 128], [194, 300, 433, 17], [494, 0, 0, 10],
                      [494, 206, 450, 450], [0, 700, 0, 900]], dtype=np.int64)
    flat, off = ops.pair_similarity(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), pairs)
    want, want_off = _oracle_pairs(q, r, pairs)
    assert np.array_equal(off, want_off)
    got = flat.cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_pair_similarity_edges():
    import torch
    from vsc_hip import ops
    from vsc_hip._lib import VscHipError
    q, r = torch.from_numpy(synth.descriptor_bank(1, 8, 32)).cuda(), torch.from_numpy(synth.descriptor_bank(2, 8, 32)).cuda()
    flat, off = ops.pair_similarity(q, r, np.zeros((0, 4), np.int64))
    assert flat.numel() == 0 and off.tolist() == [0]
    with pytest.raises(VscHipError, match="outside the banks"):
        ops.pair_similarity(q, r, np.array([[4, 5, 0, 8]], np.int64))


@pytest.mark.gpu
def test_generate_features_on_hip_path():
    query, ref = _videos(5, [40, 16]), _videos(6, [33, 140], d=64)
    qids, rids = list(query), list(ref)
    len_map = {qids[0]: 20, qids[1]: 16}
    cands = [(qids[0], rids[1], 0.7), (qids[1], rids[0], 0.6)]
    feats, infos = matching.generate_candidates_classfiy_feature(query, ref, cands, len_map)
    want_f, want_i = _reference_semantics(query, ref, cands, len_map)
    assert infos == want_i
    for got, want in zip(feats, want_f):
        g, w = (got[0], want[0]) if isinstance(want, list) else (got, want)
        assert np.array_equal(g, w)
