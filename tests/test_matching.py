"""Matching-track candidate features (row (f)): host logic on CPU with the oracle behind the test seam, and the
HIP kernel bit-exact against the oracle on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn_oracle  # noqa: E402

from src import matching  # noqa: E402
from tools import synth  # noqa: E402
from vsc_hip import _lib as _vsc_lib


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


# ---- candidate retrieval (infer_matching.py:229-262) ----------------------------------------------------------
def _sn_videos(seed, lens, prefix, d=64, copies=()):
    """Score-normalised-looking video features: unit rows scaled so random pairs sit below the -0.1 threshold
    of the reference only by a shift the test applies; `copies` = (video, frame, source rows) planted matches."""
    from vsc.index import VideoFeature
    vids = []
    for i, n in enumerate(lens):
        f = synth.descriptor_bank(seed * 1000 + i, n, d)
        vids.append(VideoFeature(video_id=f"{prefix}{i:04d}", timestamps=np.arange(float(n)), feature=f))
    for n, (vid, frame, rows) in enumerate(copies):   # scaled so that no two planted pairs tie bit for bit
        vids[vid].feature[frame:frame + len(rows)] = rows * np.float32(0.97 - 0.06 * n)
    return vids


def _planted(seed=7, d=64):
    refs = _sn_videos(seed, [20, 7, 150, 1, 64, 33], "R", d)
    queries = _sn_videos(seed + 1, [12, 36, 5, 130], "Q", d, copies=[
        (0, 3, refs[2].feature[40:46]), (1, 30, refs[0].feature[5:9]), (3, 100, refs[4].feature[:25]),
        (3, 0, refs[2].feature[100:103])])
    return queries, refs


def _threshold_for(queries, refs, frac):
    """A radius that a fraction `frac` of all frame pairs exceed (so both faiss branches are exercised)."""
    q = np.concatenate([v.feature for v in queries]).astype(np.float64)
    r = np.concatenate([v.feature for v in refs]).astype(np.float64)
    return float(np.quantile(q @ r.T, 1.0 - frac))


def test_candidate_oracle_agrees_with_float64_statement():
    from oracle import matching_oracle
    queries, refs = _planted()
    thr = 0.35
    got = matching_oracle.candidate_pairs(queries, refs, thr, top=16)
    want = {}
    for qv in queries:
        for rv in refs:
            s = (qv.feature.astype(np.float64) @ rv.feature.astype(np.float64).T).max()
            if s > thr:
                want[(qv.video_id, rv.video_id)] = s
    assert len(got) >= 4 and {(q, r) for q, r, _ in got} == set(want)   # the planted copies score 0.8-0.97
    for q, r, s in got:
        assert abs(float(s) - want[(q, r)]) < 1e-5
    assert all(got[i][2] >= got[i + 1][2] for i in range(len(got) - 1))


@pytest.mark.parametrize("frac,top", [(0.002, 16), (0.05, 8), (0.5, 4)])
def test_search_candidate_pairs_equals_reference_loop(frac, top):
    """Host logic with the oracle behind the seam == the reference loop (top-k + range fallback + dict + sort)."""
    from oracle import matching_oracle
    queries, refs = _planted()
    thr = _threshold_for(queries, refs, frac)
    want = matching_oracle.candidate_pairs(queries, refs, thr, top=top)
    got = matching.search_candidate_pairs(queries, refs, thr, video_pair_max=matching_oracle.video_pair_max)
    scores = [s for _, _, s in want]
    assert len(set(np.array(scores).view(np.uint32).tolist())) == len(scores), "test data must not tie"
    assert [(q, r) for q, r, _ in got] == [(q, r) for q, r, _ in want]
    assert np.array_equal(np.array([s for _, _, s in got]).view(np.uint32), np.array(scores).view(np.uint32))


def test_search_candidate_pairs_empty_and_no_hits():
    from oracle import matching_oracle
    queries, refs = _planted()
    assert matching.search_candidate_pairs([], refs, video_pair_max=matching_oracle.video_pair_max) == []
    assert matching.search_candidate_pairs(queries, [], video_pair_max=matching_oracle.video_pair_max) == []
    assert matching.search_candidate_pairs(queries, refs, 2.0, video_pair_max=matching_oracle.video_pair_max) == []


def test_match_refine_dataset_matches_reference_items():
    query, ref = _videos(7, [200, 30]), _videos(8, [170, 12])
    meta = [[qid, rid, query[qid], ref[rid]] for qid, rid in zip(query, ref)]
    ds = matching.MatchRefineDataset(meta, resolution=(160, 160), pair_similarity=_oracle_pairs)
    assert len(ds) == 2
    for item, (qid, rid, qf, rf) in enumerate(meta):
        x, q, r, h, w = ds[item]
        sim = knn_oracle.ip_matrix(qf, rf)
        assert (q, r, h, w) == (qid, rid, min(len(qf), 160), min(len(rf), 160)) and x.shape == (3, 160, 160)
        assert np.array_equal(x[0, :h, :w], sim[:h, :w]) and np.array_equal(x[0], x[2])
        assert x[0, h:].sum() == 0 and x[0, :, w:].sum() == 0


def _bank(videos):
    return (np.concatenate([v.feature for v in videos]).astype(np.float32),
            np.repeat(np.arange(len(videos), dtype=np.int32), [len(v.feature) for v in videos]))


@pytest.mark.gpu
@pytest.mark.parametrize("d,frac", [(64, 0.01), (512, 0.002), (100, 0.3)])
def test_video_pair_max_bit_exact(d, frac):
    import torch
    from oracle import matching_oracle
    from vsc_hip import ops
    refs = _sn_videos(21, [37] * 60 + [1, 2, 300, 129], "R", d)
    queries = _sn_videos(22, [50, 3, 128, 1, 260], "Q", d, copies=[(0, 10, refs[5].feature[:20]), (4, 200, refs[62].feature[100:140])])
    thr = _threshold_for(queries, refs, frac)
    (qb, qv), (rb, rv) = _bank(queries), _bank(refs)
    want = matching_oracle.video_pair_max(qb, qv, len(queries), rb, rv, len(refs), thr)
    for capacity in (1 << 20, 1):      # 1: the count-then-recall protocol
        lims, rvid, score = ops.video_pair_max(torch.from_numpy(qb).cuda(), torch.from_numpy(qv).cuda(), len(queries),
                                               torch.from_numpy(rb).cuda(), torch.from_numpy(rv).cuda(), len(refs), thr,
                                               capacity=capacity)
        assert np.array_equal(lims.cpu().numpy(), want[0])
        assert np.array_equal(rvid.cpu().numpy(), want[1])
        assert np.array_equal(score.cpu().numpy().view(np.uint32), want[2].view(np.uint32))
    assert want[0][-1] > 2


@pytest.mark.gpu
@pytest.mark.parametrize("nr,frac,expect_path", [(40000, 0.0005, 2), (300000, 0.4, 3)])
def test_video_pair_max_prefilter_path_equals_exact(nr, frac, expect_path):
    """The bf16 pre-filter path (fixed-threshold sweep + exact re-scoring) must produce the table of the exact fp32 sweep bit
    for bit -- including scores a hair above / below the threshold -- and fall back per query block when lists overflow
    (300 k references, 40 % of the pairs above the threshold: ~1.4 k survivors per (query, 3.6 k-reference split) list > 1024)."""
    import os
    import torch
    from vsc_hip import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(5)
    nq, d, nqv, nrv = 700, 512, 9, 1300
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1)
    r = torch.nn.functional.normalize(torch.randn(nr, d, generator=g), dim=1)
    r[1234] = q[17]                                    # a planted copy (score 1) and a near copy
    r[30000] = torch.nn.functional.normalize(q[650] + 0.05 * torch.randn(d, generator=g), dim=0)
    qv = torch.sort(torch.randint(0, nqv, (nq,), generator=g)).values.to(torch.int32)
    rv = torch.sort(torch.randint(0, nrv, (nr,), generator=g)).values.to(torch.int32)
    s = (q[:64] @ r.t()).flatten()
    thr = float(torch.quantile(s[torch.randperm(s.numel(), generator=g)[:200000]], 1.0 - frac))
    thr = float((q[3] * r[77]).sum()) if frac < 0.01 else thr    # sparse case: sit the threshold exactly on one pair's score
    outs = {}
    for path in ("exact", "bf16"):
        _vsc_lib.set_option("VSC_PAIRMAX_PATH", path)
        try:
            outs[path] = [t.cpu() for t in ops.video_pair_max(q.cuda(), qv.cuda(), nqv, r.cuda(), rv.cuda(), nrv, thr)]
            ran = _lib.require_device().vsc_video_pair_max_last_path()
        finally:
            _vsc_lib.set_option("VSC_PAIRMAX_PATH", None)
        assert ran == (1 if path == "exact" else expect_path)
    for a, b in zip(outs["exact"], outs["bf16"]):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
    assert outs["exact"][0][-1] > 0


@pytest.mark.gpu
def test_video_pair_max_every_pair_and_none():
    """threshold below every score: the dense table is full (negative maxima included); above: empty."""
    import torch
    from oracle import matching_oracle
    from vsc_hip import ops
    queries, refs = _planted()
    (qb, qv), (rb, rv) = _bank(queries), _bank(refs)
    args = (torch.from_numpy(qb).cuda(), torch.from_numpy(qv).cuda(), len(queries), torch.from_numpy(rb).cuda(),
            torch.from_numpy(rv).cuda(), len(refs))
    lims, rvid, score = ops.video_pair_max(*args, -3.0)
    want = matching_oracle.video_pair_max(qb, qv, len(queries), rb, rv, len(refs), -3.0)
    assert lims[-1].item() == len(queries) * len(refs) and np.array_equal(rvid.cpu().numpy(), want[1])
    assert np.array_equal(score.cpu().numpy().view(np.uint32), want[2].view(np.uint32))
    lims, rvid, score = ops.video_pair_max(*args, 5.0)
    assert lims.tolist() == [0] * (len(queries) + 1) and rvid.numel() == 0 and score.numel() == 0


@pytest.mark.gpu
def test_search_candidate_pairs_on_hip_path():
    from oracle import matching_oracle
    queries, refs = _planted(seed=9, d=512)
    thr = _threshold_for(queries, refs, 0.01)
    want = matching_oracle.candidate_pairs(queries, refs, thr, top=8)
    got = matching.search_candidate_pairs(queries, refs, thr)
    assert [(q, r) for q, r, _ in got] == [(q, r) for q, r, _ in want]
    assert np.array_equal(np.array([s for _, _, s in got]).view(np.uint32),
                          np.array([s for _, _, s in want]).view(np.uint32))


# ---- generate_matching_result (utils.py:80-116): host code, scipy in cv2's place ------------------------------
def _probability_map(seed, h, w, bands, noise=0.03):
    rng = np.random.default_rng(seed)
    m = (rng.random((h, w)) * noise).astype(np.float32)
    for q0, r0, n, slope, p in bands:                     # a copied segment: a (thick) line of high probability
        for t in range(n):
            i, j = q0 + t, int(round(r0 + slope * t))
            if 0 <= i < h and 0 <= j < w:
                m[i, j] = p - 0.002 * t
                if j + 1 < w and t % 3 == 0:
                    m[i, j + 1] = p - 0.1
    specks = rng.integers(0, [h, w], size=(12, 2))          # isolated hits: the "small components"
    m[specks[:, 0], specks[:, 1]] = 0.5
    return m


def test_matching_result_single_diagonal():
    m = np.zeros((40, 50), np.float32)
    for t in range(20):
        m[5 + t, 8 + t] = 0.9
    res = matching.generate_matching_result([["Q1", "R1", m, None]], threshold=0.35, std_ratio=0.5)
    assert len(res) == 1
    qid, rid, qs, rs, qe, re, score = res[0]
    assert (qid, rid, qs, rs, qe, re) == ("Q1", "R1", 5, 8, 24, 27) and abs(score - 0.9) < 1e-6
    assert matching.generate_matching_result([["Q1", "R1", np.zeros((8, 8), np.float32), None]], 0.35, 0.5) == []
    assert matching.generate_matching_result([], 0.35, 0.5) == []


@pytest.mark.parametrize("threshold,std_ratio", [(0.35, 0.5), (0.1, 1.25), (0.001, 2)])   # infer_matching.py:291-293
def test_matching_result_equals_restatement(threshold, std_ratio):
    maps = [
        ["Q1", "R1", _probability_map(1, 60, 80, [(3, 10, 30, 1.0, 0.95), (35, 50, 20, 1.0, 0.8)]), None],
        ["Q1", "R2", _probability_map(2, 48, 48, [(0, 0, 40, 1.0, 0.7)]), None],
        ["Q2", "R1", _probability_map(3, 70, 40, [(10, 5, 24, 0.5, 0.9)]), None],          # slope 1/2
        ["Q2", "R3", _probability_map(4, 30, 30, [(20, 5, 9, -1.0, 0.9)]), None],           # anti-diagonal: rejected
        ["Q3", "R3", _probability_map(5, 25, 25, []), None],                                # specks only
    ]
    got = matching.generate_matching_result(maps, threshold=threshold, std_ratio=std_ratio)
    from oracle import matching_oracle
    want = matching_oracle.matching_result(maps, threshold, std_ratio)
    key = lambda r: (r[0], r[1], int(r[2]), int(r[3]), int(r[4]), int(r[5]))
    got, want = sorted(got, key=key), sorted(want, key=key)
    assert [key(r) for r in got] == [key(r) for r in want]
    assert np.allclose([r[6] for r in got], [r[6] for r in want], rtol=0, atol=1e-12)
    if threshold >= 0.1:
        assert len(got) >= 3


# ---- vsc.baseline.localization (reference tests/test_localization.py, with a stand-in alignment model) --------
class _BoxModel:
    """Alignment stand-in with the VCSL model interface (forward_sim): one box around the entries above a threshold."""

    def __init__(self, threshold):
        self.threshold = threshold

    def forward_sim(self, sims):
        out = []
        for key, sim in sims:
            x, y = np.nonzero(sim > self.threshold)
            out.append((key, [[int(x.min()), int(y.min()), int(x.max()), int(y.max())]] if len(x) else []))
        return out


def _localization_case():
    """make_test_case_1 of the reference's test: query frames 20..29 copy frames 30..39 of the second reference."""
    from vsc.index import VideoFeature
    a, b, c = (synth.descriptor_bank(s, n, 64) for s, n in ((31, 45), (32, 30), (33, 60)))
    a[20:30] = c[30:40]
    mk = lambda i, f: VideoFeature(video_id=i, feature=f, timestamps=np.arange(len(f)) * 1.0)
    return [mk("Q1", a)], [mk("R2", b), mk("R3", c)]


def _check_localization(pair_similarity):
    from vsc.baseline.localization import VCSLLocalizationCandidateScore, VCSLLocalizationMaxSim
    from vsc.metrics import CandidatePair, Match
    queries, refs = _localization_case()
    loc = VCSLLocalizationMaxSim(queries, refs, "TN", similarity_bias=0.5, model=_BoxModel(1.4), pair_similarity=pair_similarity)
    assert loc.localize(CandidatePair("Q1", "R2", 1.0)) == []                 # no copy in this pair
    matches = loc.localize_all([CandidatePair("Q1", "R2", 1.0), CandidatePair("Q1", "R3", 2.0)])
    assert len(matches) == 1 and isinstance(matches[0], Match)
    m = matches[0]
    assert (m.query_id, m.ref_id, m.query_start, m.query_end, m.ref_start, m.ref_end) == ("Q1", "R3", 20.0, 29.0, 30.0, 39.0)
    sim = knn_oracle.ip_matrix(queries[0].feature, refs[1].feature)
    assert m.score == pytest.approx((sim + 0.5)[20:29, 30:39].max() - 0.5, abs=1e-6)   # the reference's half-open box
    assert np.array_equal(loc.similarity(CandidatePair("Q1", "R3", 2.0)), sim + np.float32(0.5))
    cs = VCSLLocalizationCandidateScore(queries, refs, "TN", model=_BoxModel(0.9), pair_similarity=pair_similarity)
    assert cs.localize(CandidatePair("Q1", "R3", 2.0))[0].score == 2.0
    assert loc.localize_all([]) == []


def test_localization_with_stand_in_model():
    _check_localization(_oracle_pairs)


def test_localization_needs_vcsl_or_a_model():
    from vsc.baseline.localization import VCSLLocalization
    queries, refs = _localization_case()
    with pytest.raises(ImportError, match="VCSL"):
        VCSLLocalization(queries, refs, "TN")


def test_match_csv_round_trip(tmp_path):
    from vsc.metrics import Match, candidate_pairs_from_matches
    ms = [Match("Q000001", "R000003", 0.75, 20, 29, 30, 39), Match("Q000001", "R000003", 0.5, 1, 2, 3, 4),
          Match("Q000002", "R000001", 0.25, 0, 5.5, 2, 7)]
    f = tmp_path / "matches.csv"
    Match.write_csv(ms, f)
    assert f.read_text().splitlines()[0] == "query_id,ref_id,query_start,query_end,ref_start,ref_end,score"
    assert Match.read_csv(f) == ms
    pairs = candidate_pairs_from_matches(ms)
    assert [(p.query_id, p.ref_id, p.score) for p in pairs] == [("Q000001", "R000003", 0.75), ("Q000002", "R000001", 0.25)]


@pytest.mark.gpu
def test_localization_on_hip_path():
    _check_localization(None)


@pytest.mark.gpu
def test_localize_and_verify_on_hip_path():
    """sscd_baseline.localize_and_verify (:107-152) with the stand-in model: both scoring branches, batches, csv."""
    import vsc.baseline.sscd_baseline as entry
    from vsc.metrics import CandidatePair
    queries, refs = _localization_case()
    cands = [CandidatePair("Q1", "R3", 0.8), CandidatePair("Q1", "R2", 0.1)]
    got = entry.localize_and_verify(queries, refs, cands, score_normalization=True, model=_BoxModel(1.4))
    assert len(got) == 1 and got[0].ref_id == "R3" and (got[0].query_start, got[0].ref_end) == (20.0, 39.0)
    got = entry.localize_and_verify(queries, refs, cands, score_normalization=False, model=_BoxModel(0.9))
    assert len(got) == 1 and got[0].score == 0.8
    assert entry.localize_and_verify(queries, refs, cands, localize_per_query=0.0, model=_BoxModel(0.9)) == []
