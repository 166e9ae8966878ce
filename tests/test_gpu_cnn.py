"""Matching-track networks on the HIP path (vsc_hip/cnn.py over vsc_conv2d_f32 & co) against the torch fp32 oracle
(oracle/cnn_oracle.py) on the same timm-named state dicts and similarity maps.  Tolerances: the convolutions run on
exact-fp32 MFMA chains, so layers agree with F.conv2d to summation-order rounding; 1e-3 on probabilities is the tier's
bound, the observed error is ~1e-5."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cnn_synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,act,res", [
    (2, 9, 7, 3, 16, 3, 2, "hard_swish", False),     # the classifier's stem: K = 27 (padded to 32), ragged everything
    (1, 16, 16, 18, 18, 3, 1, "relu", True),         # an HRNet BasicBlock convolution with its residual
    (3, 5, 6, 64, 256, 1, 1, "relu", True),          # Bottleneck expansion, 2 channel tiles
    (2, 12, 10, 36, 72, 3, 2, None, False),          # a fuse-layer downsampling step
    (4, 1, 1, 576, 1024, 1, 1, "hard_swish", False),  # conv_head on pooled features (4 rows)
    (1, 20, 20, 334, 64, 1, 1, "relu", False),       # fuse.0
    (2, 7, 7, 24, 8, 5, 1, "hard_sigmoid", False),   # 5x5, K = 600
])
def test_conv2d_matches_torch(dev, n, h, w, cin, cout, k, stride, act, res):
    from vsc_hip import cnn
    rng = np.random.RandomState(n * 100 + cin)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, cin, h, w).astype(np.float32))
    want = F.conv2d(x, sd["c.weight"], sd["c.bias"], stride=stride, padding=k // 2)
    r = torch.from_numpy(rng.randn(*want.shape).astype(np.float32)) if res else None
    if res:
        want = want + r
    want = {"relu": F.relu, "hard_swish": F.hardswish, "hard_sigmoid": F.hardsigmoid, None: lambda v: v}[act](want)
    conv = cnn.Conv(sd, "c", None, stride, dev)
    got = conv(x.permute(0, 2, 3, 1).contiguous().to(dev), act=act,
               residual=None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev))
    assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    # channel window of a wider buffer (the concat path)
    wide = torch.full((n, want.shape[2], want.shape[3], cout + 5), 7.0, device=dev)
    conv(x.permute(0, 2, 3, 1).contiguous().to(dev), act=act, residual=None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
         out=wide, coff=3)
    assert torch.equal(wide[..., 3:3 + cout], got) and (wide[..., :3] == 7).all() and (wide[..., 3 + cout:] == 7).all()


def test_depthwise_pool_scale_upsample(dev):
    from vsc_hip import cnn
    rng = np.random.RandomState(0)
    for c, k, stride in ((16, 3, 2), (96, 5, 1), (240, 5, 2)):
        sd = {"d.weight": torch.from_numpy(rng.randn(c, 1, k, k).astype(np.float32) * 0.2)}
        x = torch.from_numpy(rng.randn(2, c, 11, 9).astype(np.float32))
        want = F.hardswish(F.conv2d(x, sd["d.weight"], None, stride=stride, padding=k // 2, groups=c))
        got = cnn.DwConv(sd, "d", None, stride, dev)(x.permute(0, 2, 3, 1).contiguous().to(dev), act="hard_swish")
        assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=1e-5)
    x = torch.from_numpy(rng.randn(3, 6, 5, 70).astype(np.float32)).to(dev)
    assert torch.allclose(cnn.avgpool(x).reshape(3, 70), x.mean((1, 2)), atol=1e-6)
    src = torch.from_numpy(rng.randn(2, 3, 4, 18).astype(np.float32)).to(dev)
    out = torch.from_numpy(rng.randn(2, 12, 16, 30).astype(np.float32)).to(dev)
    want = out.clone()
    want[..., 5:23] = F.relu(want[..., 5:23] + F.interpolate(src.permute(0, 3, 1, 2), scale_factor=4, mode="nearest").permute(0, 2, 3, 1))
    cnn.upsample_into(src, out, 4, 5, True, "relu")
    assert torch.equal(out, want)


def test_mobilenetv3_classifier_matches_oracle(dev):
    from oracle import cnn_oracle
    from vsc_hip import cnn
    sds = [cnn_synth.mobilenetv3_small_state(11), cnn_synth.mobilenetv3_small_state(12)]
    x = cnn_synth.similarity_maps(13, 6, 160, 160)          # MatchClassifyDataset resolution (infer_matching.py:159)
    with torch.no_grad():
        strip = [{k[len("model."):]: v for k, v in sd.items()} for sd in sds]
        want_logits = cnn_oracle.mobilenetv3_small(strip[0], x)
        want = cnn_oracle.match_classify_probability(strip, x)
    models = [cnn.MobileNetV3SmallHip(sd, dev) for sd in sds]
    got_logits = models[0](x.to(dev)).cpu()
    got = cnn.match_classify_probability(models, x.to(dev)).cpu()
    assert got_logits.shape == (6, 2)
    assert float((got_logits - want_logits).abs().max()) < 1e-3 * max(1.0, float(want_logits.abs().max()))
    assert float((got - want).abs().max()) < 1e-4            # probabilities (tier bound 1e-3)


@pytest.mark.parametrize("n,h,w", [(2, 32, 48), (1, 224, 224)])   # 224 x 224 = MatchRefineDataset resolution (infer_matching.py:178)
def test_hrnet_refine_matches_oracle(dev, n, h, w):
    from oracle import cnn_oracle
    from vsc_hip import cnn
    sd = cnn_synth.hrnet_refine_state(21)
    x = cnn_synth.similarity_maps(22, n, h, w)
    with torch.no_grad():
        want_logits = cnn_oracle.hrnet_refine(sd, x)
        want = cnn_oracle.match_refine_probability([sd], x) if h <= 64 else want_logits.softmax(dim=1)
    model = cnn.HRNetRefineHip(sd, dev)
    got_logits = model(x.to(dev)).cpu()
    assert got_logits.shape == (n, 2, h, w)
    assert float((got_logits - want_logits).abs().max()) < 1e-3 * max(1.0, float(want_logits.abs().max()))
    got = cnn.match_refine_probability([model], x.to(dev)).cpu() if h <= 64 else got_logits.softmax(dim=1)
    assert float((got - want).abs().max()) < 1e-4            # probability map (tier bound 1e-3)


def test_match_classify_and_refine_entry_points(dev):
    """src.matching.match_classify / match_refine (infer_matching.py:158-204) end to end on synthetic candidates: datasets,
    batching, model averaging, transposed pass, crop to the valid h x w -- against the oracle's restatement of the same steps,
    and into generate_matching_result (a planted diagonal segment must come out as a match)."""
    from oracle import cnn_oracle
    from src import matching
    rng = np.random.RandomState(3)
    cls_sds = [cnn_synth.mobilenetv3_small_state(31), cnn_synth.mobilenetv3_small_state(32)]
    ref_sds = [cnn_synth.hrnet_refine_state(33)]
    cls_models, refine_models = matching.load_match_models(cls_sds, ref_sds, dev)
    feats = [(rng.randn(h, w) * 0.1).astype(np.float32) for h, w in ((30, 50), (200, 170), (12, 9))]   # one larger than 160: cropped
    infos = [("Q1", "R1"), ("Q2", "R7"), ("Q3", "R2")]
    got = matching.match_classify(cls_models, feats, infos, batch_size=2, device=dev)
    assert [(q, r) for q, r, _ in got] == infos
    data = matching.MatchClassifyDataset(feats, infos, matching.MATCH_CLS_RESOLUTION)
    x = torch.from_numpy(np.stack([data[i][0] for i in range(3)]))
    with torch.no_grad():
        want = cnn_oracle.match_classify_probability([{k[len("model."):]: v for k, v in sd.items()} for sd in cls_sds], x)
    assert np.allclose([p for _, _, p in got], want.numpy(), atol=1e-4)

    d = 64
    meta = []
    for qn, rn in ((20, 36), (7, 11)):
        q = rng.randn(qn, d).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        r = rng.randn(rn, d).astype(np.float32)
        r /= np.linalg.norm(r, axis=1, keepdims=True)
        meta.append((f"Q{qn}", f"R{rn}", q, r))
    res = matching.match_refine(refine_models, meta, batch_size=1, device=dev)
    assert [(a, b) for a, b, _, _ in res] == [("Q20", "R36"), ("Q7", "R11")]
    for (qid, rid, prob, sim), (_, _, q, r) in zip(res, meta):
        assert prob.shape == sim.shape == (len(q), len(r))
        canvas = np.zeros((224, 224), np.float32)
        canvas[:len(q), :len(r)] = sim
        with torch.no_grad():
            want = cnn_oracle.match_refine_probability(ref_sds, torch.from_numpy(np.stack([canvas] * 3)[None]))[0, 1, :len(q), :len(r)]
        assert np.abs(prob - want.numpy()).max() < 1e-4
        assert np.abs(sim - q @ r.T).max() < 1e-5
    out = matching.generate_matching_result([["Qx", "Ry", np.eye(40, dtype=np.float32) * 0.9, None]])
    assert len(out) == 1 and out[0][:2] == ["Qx", "Ry"]
