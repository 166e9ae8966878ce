"""Host-side logic that needs no GPU: descriptor file format, video-id formatting, pair
bookkeeping, weight-name translation -- and that the product path refuses to run without
the HIP device instead of falling back to anything."""
import io
import os

import numpy as np
import pytest
import torch

from vsc.index import VideoFeature, VideoIndex
from vsc.metrics import CandidatePair, Dataset, average_precision, format_video_id, micro_average_precision
from vsc.storage import load_features, same_value_ranges, store_features

no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")


def _vf(video_id, n, dims=32, interval=False, fps=1.0, seed=0):
    rs = np.random.RandomState(seed + n)
    ts = np.arange(n) / fps
    if interval:
        ts = np.stack([ts, ts + fps], axis=1)
    return VideoFeature(video_id=video_id, timestamps=ts, feature=rs.randn(n, dims))


@pytest.mark.parametrize("interval", [False, True])
def test_storage_round_trip(interval):
    """Mirrors the reference's tests/test_storage.py (merged storage, int -> 'Q%06d' ids)."""
    feats = [_vf(2, 10, interval=interval), _vf(3, 20, interval=interval, fps=3.0),
             _vf(1, 30, interval=interval, fps=0.5)]
    buf = io.BytesIO()
    store_features(buf, feats, Dataset.QUERIES)
    buf.seek(0)
    back = load_features(buf)
    assert [b.video_id for b in back] == ["Q000002", "Q000003", "Q000001"]
    for a, b in zip(feats, back):
        np.testing.assert_allclose(b.timestamps, a.timestamps)
        np.testing.assert_allclose(b.feature, a.feature.astype(np.float32))
        assert b.feature.dtype == np.float32
    buf2 = io.BytesIO()
    store_features(buf2, back)
    buf2.seek(0)
    for a, b in zip(back, load_features(buf2)):
        assert a.video_id == b.video_id
        np.testing.assert_array_equal(a.feature, b.feature)


def test_npz_layout_is_the_reference_layout():
    buf = io.BytesIO()
    store_features(buf, [_vf("R000007", 3), _vf("R000009", 2)])
    buf.seek(0)
    data = np.load(buf)
    assert sorted(data.files) == ["features", "timestamps", "video_ids"]
    assert data["video_ids"].tolist() == ["R000007"] * 3 + ["R000009"] * 2
    assert data["features"].shape == (5, 32) and data["features"].dtype == np.float32


def test_same_value_ranges_and_id_format():
    assert list(same_value_ranges(["a", "a", "b", "c", "c", "c"])) == [("a", 0, 2), ("b", 2, 3), ("c", 3, 6)]
    assert format_video_id(12, Dataset.REFS) == "R000012"
    assert format_video_id("Q000001", Dataset.QUERIES) == "Q000001"
    with pytest.raises(AssertionError):
        format_video_id("R000001", Dataset.QUERIES)
    with pytest.raises(ValueError):
        format_video_id(3, None)


def test_load_rejects_bad_timestamps():
    buf = io.BytesIO()
    np.savez(buf, video_ids=np.array(["Q1"] * 3), features=np.zeros((3, 4), np.float32), timestamps=np.zeros(2))
    buf.seek(0)
    with pytest.raises(ValueError):
        load_features(buf)


def test_pair_matches_bookkeeping():
    """The grouping the reference does after the search (index.py:129-143), fed with hits."""
    refs = [VideoFeature("R1", np.array([2.0, 4.0, 6.0]), np.zeros((3, 4))),
            VideoFeature("R2", np.array([[0.0, 5.0], [5.0, 10.0]]), np.zeros((2, 4)))]
    idx = VideoIndex(4)
    idx.add(refs)
    assert idx.index.ntotal == 5 and idx.video_clip_idx == [0, 1, 2, 0, 1]
    q = VideoFeature("Q1", np.array([0.0, 1.0]), np.zeros((2, 4)))
    hits = [(0, 1, 0.9), (1, 4, 0.8), (1, 2, 0.7)]
    pm = idx.pair_matches(hits, ["Q1", "Q1"], [0, 1], {"Q1": q.metadata()})
    got = {(p.query_id, p.ref_id): p.matches for p in pm}
    assert got[("Q1", "R1")][0].ref_timestamps == (4.0, 4.0) and got[("Q1", "R1")][1].score == 0.7
    assert got[("Q1", "R2")][0].ref_timestamps == (5.0, 10.0) and got[("Q1", "R2")][0].query_timestamps == (1.0, 1.0)


def test_micro_ap():
    gt = [CandidatePair("Q1", "R2", 1.0)]
    preds = [CandidatePair("Q2", "R2", 3.0), CandidatePair("Q1", "R1", 2.0), CandidatePair("Q1", "R2", 1.0)]
    assert micro_average_precision(gt, preds) == pytest.approx(1 / 3)   # reference tests/test_metrics.py:124-135
    assert micro_average_precision(gt, preds[2:]) == 1.0
    with pytest.raises(AssertionError):
        micro_average_precision(gt + gt, preds)


def _C(q, r, s):
    return CandidatePair(format_video_id(q, Dataset.QUERIES), format_video_id(r, Dataset.REFS), s)


def test_average_precision_reference_unit_vectors():
    """DescriptorTrackTest.test_uap of the reference (train/train_v115/tests/test_metrics.py:215-240), as data."""
    gt = [_C(1, 10, 1.0), _C(2, 11, 1.0)]
    for want, preds in [(1.0, [_C(1, 10, 8.0), _C(2, 11, 4.0), _C(99, 99, 2.0)]),
                        (np.mean([1, 2 / 3]), [_C(1, 10, 8.0), _C(2, 11, 4.0), _C(99, 99, 5.0)]),
                        (np.mean([1, 0]), [_C(1, 10, 3.0), _C(2, 10, 2.0), _C(99, 99, 1.0)]),
                        (np.mean([1 / 2, 0]), [_C(1, 10, 2.0), _C(2, 10, 3.0), _C(99, 99, 1.0)])]:
        m = average_precision(gt, preds)
        assert m.ap == pytest.approx(want, abs=1e-12) and m.simple_ap == pytest.approx(m.ap, abs=1e-12)
        assert len(m.pr_curve.scores) == len(m.pr_curve.recalls) == len(m.pr_curve.precisions)
    with pytest.raises(AssertionError):
        average_precision(gt, [_C(1, 10, 1.0), _C(1, 10, 2.0)])


def test_average_precision_matches_reference_outputs_under_ties():
    """tests/golden/uap_reference.json holds outputs of the reference's own average_precision (gen_uap_golden.py):
    `.ap` groups tied scores (sklearn) and is rescaled by predicted / actual positives; `.simple_ap` is tie-blind.
    They differ by up to 1.5e-2 on these vectors — the reported number must be `.ap`."""
    import json
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "uap_reference.json")))
    assert any(abs(c["ap"] - c["simple_ap"]) > 1e-3 for c in cases)
    for c in cases:
        preds = [_C(a, b, s) for a, b, s in c["pred"]]
        gt = [_C(a, b, 1.0) for a, b in c["gt"]]
        m = average_precision(gt, preds)
        assert m.ap == pytest.approx(c["ap"], abs=1e-12)
        assert m.simple_ap == pytest.approx(c["simple_ap"], abs=1e-12)
        assert float(np.sum(m.pr_curve.recalls)) == pytest.approx(c["curve_recalls_sum"], abs=1e-9)
        assert float(np.sum(m.pr_curve.precisions)) == pytest.approx(c["curve_precisions_sum"], abs=1e-9)
        assert micro_average_precision(gt, preds) == m.simple_ap
        # order independence of the canonical number (ties are grouped, so shuffling cannot move it)
        rng = np.random.default_rng(1)
        perm = rng.permutation(len(preds))
        assert average_precision(gt, [preds[i] for i in perm]).ap == pytest.approx(c["ap"], abs=1e-12)


def test_tie_grouped_ap_equals_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    from vsc.metrics import _tie_grouped_ap
    rng = np.random.default_rng(5)
    for dec in (None, 2, 1, 0):
        s = rng.random(500)
        s = s if dec is None else np.round(s, dec)
        y = (rng.random(500) < 0.2).astype(np.float64)
        assert _tie_grouped_ap(y, s) == pytest.approx(sk.average_precision_score(y, s), abs=1e-12)


def test_weight_name_translation_round_trip():
    from tools import synth
    from vsc_hip import weights as W
    from vsc_hip.config import get_config
    cfg = get_config("tiny")
    w = synth.encoder_weights(3, cfg)
    d = cfg.width
    hf_old = {"vit.embeddings.cls_token": w["cls"].reshape(1, 1, d),
              "vit.embeddings.position_embeddings": w["pos"][None],
              "vit.embeddings.patch_embeddings.projection.weight": w["patch.weight"],
              "vit.embeddings.patch_embeddings.projection.bias": w["patch.bias"],
              "vit.layernorm.weight": w["ln_post.weight"], "vit.layernorm.bias": w["ln_post.bias"],
              "vit.pooler.dense.weight": np.zeros((d, d), np.float32),
              "output_proj.weight": w["head.weight"], "output_proj.bias": w["head.bias"]}
    for i in range(cfg.layers):
        b, t = f"blocks.{i}.", f"vit.encoder.layer.{i}."
        for j, nm in enumerate(("query", "key", "value")):
            hf_old[t + f"attention.attention.{nm}.weight"] = w[b + "qkv.weight"][j * d:(j + 1) * d]
            hf_old[t + f"attention.attention.{nm}.bias"] = w[b + "qkv.bias"][j * d:(j + 1) * d]
        for src, dst in (("proj", "attention.output.dense"), ("ln1", "layernorm_before"), ("ln2", "layernorm_after"),
                         ("fc1", "intermediate.dense"), ("fc2", "output.dense")):
            hf_old[t + dst + ".weight"], hf_old[t + dst + ".bias"] = w[b + src + ".weight"], w[b + src + ".bias"]
    got = W.from_hf_vit(hf_old, cfg)
    W.check_complete(got, cfg)
    for k in W.canonical_names(cfg):
        np.testing.assert_array_equal(got[k].reshape(w[k].shape), w[k])

    timm = {"cls_token": w["cls"].reshape(1, 1, d), "pos_embed": w["pos"][None],
            "patch_embed.proj.weight": w["patch.weight"], "patch_embed.proj.bias": w["patch.bias"],
            "norm.weight": w["ln_post.weight"], "norm.bias": w["ln_post.bias"]}
    for i in range(cfg.layers):
        for src, dst in (("ln1", "norm1"), ("qkv", "attn.qkv"), ("proj", "attn.proj"), ("ln2", "norm2"),
                         ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            for kind in ("weight", "bias"):
                timm[f"blocks.{i}.{dst}.{kind}"] = w[f"blocks.{i}.{src}.{kind}"]
    got = W.from_timm_vit(timm, get_config("tiny", out_dim=0))
    W.check_complete(got, get_config("tiny", out_dim=0))
    # the vit_v68 checkpoint layout (train/train_v68/torch2scripts.py:17-23): model.backbone.* + model.embeddings.*
    scfg = get_config("tiny_sscd")
    sw = synth.encoder_weights(3, scfg)
    v68 = {"model.backbone." + k: v for k, v in timm.items()}
    v68.update({"model.embeddings.0.conv.weight": sw["head_conv.weight"][:, :, None],
                "model.embeddings.0.conv.bias": sw["head_conv.bias"],
                "model.embeddings.1.weight": sw["head.weight"], "model.embeddings.1.bias": sw["head.bias"]})
    got = W.from_timm_vit(v68, scfg)
    W.check_complete(got, scfg)
    np.testing.assert_array_equal(got["head_conv.weight"], sw["head_conv.weight"])
    np.testing.assert_array_equal(got["head.weight"], sw["head.weight"])

    ccfg = get_config("tiny_clip")
    cw = synth.encoder_weights(3, ccfg)
    clip = {"visual.conv1.weight": cw["patch.weight"], "visual.class_embedding": cw["cls"],
            "visual.positional_embedding": cw["pos"]}
    for nm in ("ln_pre", "ln_post"):
        clip[f"visual.{nm}.weight"], clip[f"visual.{nm}.bias"] = cw[nm + ".weight"], cw[nm + ".bias"]
    for i in range(ccfg.layers):
        t, b = f"visual.transformer.resblocks.{i}.", f"blocks.{i}."
        clip[t + "attn.in_proj_weight"], clip[t + "attn.in_proj_bias"] = cw[b + "qkv.weight"], cw[b + "qkv.bias"]
        for src, dst in (("proj", "attn.out_proj"), ("ln1", "ln_1"), ("ln2", "ln_2"), ("fc1", "mlp.c_fc"), ("fc2", "mlp.c_proj")):
            clip[t + dst + ".weight"], clip[t + dst + ".bias"] = cw[b + src + ".weight"], cw[b + src + ".bias"]
    got = W.from_clip_visual(clip, ccfg)
    W.check_complete(got, ccfg)


@no_gpu
def test_product_path_fails_loudly_without_gpu():
    from tools import synth
    from vsc_hip import _lib, ops
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    cfg = get_config("tiny")
    with pytest.raises(_lib.HipPathUnavailable, match="no CPU fallback"):
        HipEncoder(cfg, synth.encoder_weights(1, cfg))
    with pytest.raises(_lib.HipPathUnavailable):
        ops.l2_normalize_(torch.zeros(2, 4))
    idx = VideoIndex(4)
    idx.add([VideoFeature("R1", np.arange(3.0), np.ones((3, 4), np.float32))])
    with pytest.raises(_lib.HipPathUnavailable):
        idx.search([VideoFeature("Q1", np.arange(2.0), np.ones((2, 4), np.float32))], global_k=2)


def test_config_flops():
    from vsc_hip.config import get_config
    cfg = get_config("vit_b16_224")
    assert cfg.tokens == 197 and cfg.head_dim == 64 and cfg.patch_dim == 768
    assert 34.5e9 < cfg.flops_per_frame() < 36e9   # ViT-B/16: ~17.5 GMAC


class _NumpyOps:
    """CPU stand-in for the two device ops, only to exercise the host bookkeeping here."""

    @staticmethod
    def normalize(x):
        from sklearn.preprocessing import normalize
        return normalize(np.asarray(x, np.float32))

    @staticmethod
    def self_similarity(x):
        return np.matmul(x, x.T)

    @staticmethod
    def similarity(a, b):
        return np.matmul(a, b.T)


def test_hip_pca_mirrors_sklearn_transform():
    from tools import synth
    """HipPCA.transform == sklearn PCA.transform (plain and whitened) on the fitted attributes the reference pickles."""
    from sklearn.decomposition import PCA
    from src.query_postprocess import HipPCA
    x = synth.normalish(21, (200, 48)) * np.linspace(0.2, 3.0, 48, dtype=np.float32)
    for whiten in (False, True):
        fitted = PCA(n_components=16, whiten=whiten, random_state=0).fit(x)
        got = HipPCA(fitted, ops=_NumpyOps).transform(x[:50])
        np.testing.assert_allclose(got, fitted.transform(x[:50]), rtol=1e-4, atol=1e-5)


def test_query_postprocess_matches_reference_statement():
    """src/query_postprocess.py against a literal numpy restatement of extract_query_feats.py:176-228."""
    from sklearn.preprocessing import normalize
    from src.query_postprocess import process_query_video
    rs = np.random.RandomState(0)
    base = rs.randn(6, 16).astype(np.float32)
    frames = np.concatenate([base, base[:3] + 1e-3 * rs.randn(3, 16).astype(np.float32)])   # 3 near-duplicates
    subs = [frames[:, :8].copy(), frames[:, 8:].copy()]
    stamps = np.stack([np.arange(9.0), np.arange(9.0) + 1], axis=1)
    proj = rs.randn(16, 5).astype(np.float32)
    pca = lambda x: x @ proj

    feat, per_model, idx = process_query_video("Q000001", subs, stamps, 0.9, pca, rnd_idx=0, ops=_NumpyOps)
    # reference statement
    sub_n = [normalize(s) for s in subs]
    f = np.concatenate(sub_n, axis=1)
    fn = f / np.linalg.norm(f, axis=1, keepdims=True)
    sim = np.matmul(fn, fn.T) - np.eye(len(fn))
    to_remove = []
    for i in sim.mean(0).argsort()[::-1]:
        if i in to_remove:
            continue
        for j in np.where(sim[i] > 0.975)[0]:
            to_remove.append(j)
    keep = [i for i in range(len(sim)) if i not in to_remove]
    assert len(keep) == 6 and idx == 0
    np.testing.assert_allclose(feat.feature, pca(f[keep]), rtol=1e-6)
    np.testing.assert_array_equal(feat.timestamps, stamps[keep])
    assert len(per_model) == 2 and per_model[1].feature.shape == (9, 8)

    low, _, idx = process_query_video("Q000002", subs, stamps, 0.0001, pca, rnd_idx=4, ops=_NumpyOps)
    np.random.seed(5)
    assert idx == 5 and low.feature.shape == (1, 512)
    np.testing.assert_array_equal(low.feature[0], np.random.uniform(-1e-5, 1e-5, size=512).astype(np.float32))


class _NumpySweep:
    """Stand-in for vsc_hip.ops on CPU tensors: the same three calls on the oracle's chains.  Exercises the HOST logic of
    vsc.index (metric augmentation, probe / radius search, sorting); the HIP sweeps themselves are tested with -m gpu."""

    @staticmethod
    def knn_ip(q, r, k, ref_id_offset=0):
        import torch
        from oracle import knn_oracle
        D, I = knn_oracle.knn_ip(q.numpy(), r.numpy(), k)
        return torch.from_numpy(D), torch.from_numpy(I + ref_id_offset * (I >= 0))

    @staticmethod
    def range_search_ip(q, r, radius, ref_id_offset=0, capacity=0):
        import torch
        from oracle import knn_oracle
        lims, D, I = knn_oracle.range_search_ip(q.numpy(), r.numpy(), float(radius))
        return torch.from_numpy(lims), torch.from_numpy(D), torch.from_numpy(I)

    @staticmethod
    def range_count_ip(q, r, radius):
        from oracle import knn_oracle
        return int(knn_oracle.range_search_ip(q.numpy(), r.numpy(), float(radius))[0][-1])


@pytest.fixture
def cpu_sweep(monkeypatch):
    import torch
    import vsc.index as vi
    from vsc_hip import ops
    monkeypatch.setattr(vi.FlatIPBank, "_to_device", staticmethod(lambda host: torch.from_numpy(np.ascontiguousarray(host))))
    for name in ("knn_ip", "range_search_ip", "range_count_ip"):
        monkeypatch.setattr(ops, name, getattr(_NumpySweep, name))
    return vi


def test_global_threshold_search_host_logic(cpu_sweep):
    """_global_threshold_knn_search == the min(global_k, nq * nr) best pairs of the full score matrix in all three regimes:
    probe sufficient; one row owns more winners than the probe (threshold range sweep); probe smaller than global_k
    (radius found by counting) -- index.py:145-165 of the reference returns exactly that set."""
    vi = cpu_sweep
    from oracle import knn_oracle
    rng = np.random.RandomState(0)
    r = rng.randn(400, 16).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    for nq, gk, probe in ((6, 30, 1024), (6, 300, 16), (1, 250, 64), (2, 799, 32), (2, 5000, 32)):
        q = rng.randn(nq, 16).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        if probe == 16:
            r[50:200] = q[2] + 0.01 * rng.randn(150, 16).astype(np.float32)
        idx = vi.VideoIndex(16)
        idx.add([vi.VideoFeature("R1", np.arange(400.0), r)])
        S = knn_oracle.ip_matrix(q, r)
        old, vi.MAX_K = vi.MAX_K, probe
        try:
            hits = idx._global_threshold_knn_search(q, gk)
        finally:
            vi.MAX_K = old
        n = min(gk, nq * 400)
        flat = np.argsort(-S.ravel().astype(np.float64), kind="stable")[:n]
        assert len(hits) == n
        assert sorted((i, j) for i, j, _ in hits) == sorted((int(f // 400), int(f % 400)) for f in flat)
        assert all(s == S[i, j] for i, j, s in hits) and all(a[2] >= b[2] for a, b in zip(hits, hits[1:]))


def test_flat_l2_index_host_logic(cpu_sweep):
    """METRIC_L2 through the inner-product sweep (rows [r, |r|^2, 1] against [2q, -1, -|q|^2]): exact squared distances,
    brute-force ids, and the reference's own tests/test_index.py vectors through both VideoIndex search modes."""
    vi = cpu_sweep
    rng = np.random.RandomState(3)
    r = (rng.randn(300, 12) * 3).astype(np.float32)
    q = (rng.randn(9, 12) * 3).astype(np.float32)
    q[5] = r[77]
    bank = vi.FlatIPBank(12, vi.METRIC_L2)
    bank.add(r[:100])
    bank.add(r[100:])
    D, I = bank.search(q, 7)
    d64 = ((q[:, None, :].astype(np.float64) - r[None].astype(np.float64)) ** 2).sum(-1)
    want = np.argsort(d64, axis=1, kind="stable")[:, :7]
    assert np.array_equal(I, want) and D[5, 0] == 0.0 and (np.diff(D, axis=1) >= 0).all()
    assert np.allclose(D, np.take_along_axis(d64, want, 1), rtol=1e-5, atol=1e-5)
    rows, ids, dist = bank.range_search(q, 60.0)
    assert sorted(zip(rows.tolist(), ids.tolist())) == sorted(map(tuple, np.argwhere(d64 < 60.0).tolist()))
    assert bank.range_count(q, 60.0) == len(rows)
    feats = np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[11, 12, 13], [14, 15, 16], [17, 18, 19]],
                      [[111, 112, 113], [114, 115, 116], [117, 118, 119]]], np.float32)
    mk = lambda pre: [vi.VideoFeature(video_id=f"{pre}{i:06d}", feature=f, timestamps=np.arange(3, dtype=np.float32))
                      for i, f in enumerate(feats)]
    for gk in (1, -1, 4):
        idx = vi.VideoIndex(3, "Flat", vi.METRIC_L2)
        idx.add(mk("R"))
        res = idx.search(mk("Q"), gk)
        assert res and all(x.query_id[1:] == x.ref_id[1:] for x in res)


def test_matching_entry_derives_frames_per_video_from_timestamps():
    """infer_matching.py without --query_frames: a multi-view query file (views repeat the timestamps) must give the frame
    count, not the descriptor count (reference: vid_feature_len_map from extraction, infer_matching.py:155)."""
    import infer_matching
    from vsc.index import VideoFeature
    ts = np.arange(7, dtype=np.float32)
    one = VideoFeature(video_id="Q1", feature=np.zeros((7, 4), np.float32), timestamps=ts)
    three = VideoFeature(video_id="Q2", feature=np.zeros((21, 4), np.float32), timestamps=np.tile(ts, 3))
    spans = VideoFeature(video_id="Q3", feature=np.zeros((10, 4), np.float32),
                         timestamps=np.tile(np.stack([np.arange(5.0), np.arange(5.0) + 1], 1), (2, 1)))
    assert [infer_matching.frames_per_video(v) for v in (one, three, spans)] == [7, 7, 5]
    bad = VideoFeature(video_id="Q4", feature=np.zeros((5, 4), np.float32), timestamps=np.array([0, 1, 0, 1, 2], np.float32))
    with pytest.raises(ValueError, match="query_frames"):
        infer_matching.frames_per_video(bad)
    # --query_frames covers the irregular video: it must be used, not re-derived; uncovered ones are counted
    assert infer_matching.query_len_map([one, three, bad], {"Q4": 5}) == {"Q1": 7, "Q2": 7, "Q4": 5}
    with pytest.raises(ValueError, match="query_frames"):
        infer_matching.query_len_map([one, bad], {"Q1": 7})
