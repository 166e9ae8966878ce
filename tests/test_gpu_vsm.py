"""Video-score head on the HIP path vs the oracle (tiny config from the transformers-pinned golden, and the
full BERT-base shape on fresh inputs)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 2e-3   # the head runs in fp32 end to end (observed ~1e-5); the gate compares sigmoid(logit) with 1e-3


def _head(preset, seed):
    from tools import synth
    from vsc_hip.video_score import VideoScoreHead
    from vsc_hip.vsm_config import get_vsm_config
    cfg = get_vsm_config(preset)
    w = synth.vsm_weights(seed, cfg)
    return cfg, w, VideoScoreHead(cfg, w)


def test_tiny_head_matches_golden_and_oracle():
    from oracle import vsm_oracle
    from tools import synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "vsm_tiny_vsm.npz"))
    cfg, w, head = _head("tiny_vsm", int(g["weights_seed"]))
    for i, n in enumerate(g["n_valid"].tolist()):
        f = torch.from_numpy(synth.normalish(int(g["feats_seed"]) + i, (n, cfg.feat_dim)))
        got = float(head.logit(f.cuda()))
        assert abs(got - float(g["logits"][i])) < LOGIT_ATOL, (n, got, float(g["logits"][i]))
        assert abs(head.score(f.cuda()) - vsm_oracle.video_score(w, cfg, f)) < 1e-3


@pytest.mark.parametrize("n", [3, 200, 256, 300])
def test_full_size_head_vs_oracle(n):
    from oracle import vsm_oracle
    from tools import synth
    cfg, w, head = _head("vsm_roberta_base", 41)
    f = torch.from_numpy(synth.normalish(100 + n, (n, cfg.feat_dim)))
    fo = torch.zeros(cfg.max_frames, cfg.feat_dim)
    fo[: min(n, cfg.max_frames)] = f[: cfg.max_frames]
    want = float(vsm_oracle.ms_forward(w, cfg, fo[None])[0])
    got = float(head.logit(f.cuda()))
    assert abs(got - want) < LOGIT_ATOL, (n, got, want)


def test_no_video_near_the_gate_flips():
    """extract_query_feats.py:53,172-174 keeps a video when sigmoid(logit) >= SCORE_THRESHOLD = 0.001.  Heads whose output
    bias is shifted so that a set of synthetic videos lands within 1e-2 (in logit) of the gate on either side must take
    the oracle's decision for every one of them."""
    from oracle import vsm_oracle
    from tools import synth
    from vsc_hip.video_score import VideoScoreHead
    from src.query_postprocess import SCORE_THRESHOLD
    gate = float(np.log(SCORE_THRESHOLD / (1.0 - SCORE_THRESHOLD)))      # logit of the threshold
    cfg, w, _ = _head("vsm_roberta_base", 43)
    feats = [torch.from_numpy(synth.normalish(500 + i, (n, cfg.feat_dim))) for i, n in enumerate((5, 40, 120, 256, 31, 77))]
    pad = lambda f: torch.cat([f[: cfg.max_frames], torch.zeros(max(cfg.max_frames - f.shape[0], 0), cfg.feat_dim)])
    base = [float(vsm_oracle.ms_forward(w, cfg, pad(f)[None])[0]) for f in feats]
    checked = 0
    for f, b in zip(feats, base):
        for delta in (-1e-2, -3e-3, 3e-3, 1e-2):                       # move this video to gate + delta
            w2 = dict(w)
            w2["output_proj.bias"] = w["output_proj.bias"] + np.float32(gate + delta - b)
            want = float(vsm_oracle.ms_forward(w2, cfg, pad(f)[None])[0])
            got = float(VideoScoreHead(cfg, w2).logit(f.cuda()))
            assert abs(want - gate) < 1.5e-2 and abs(got - want) < LOGIT_ATOL
            assert (got >= gate) == (want >= gate), (delta, got, want, gate)
            checked += 1
    assert checked == 24


def test_head_rejects_bad_input():
    from vsc_hip._lib import HipPathUnavailable
    cfg, w, head = _head("tiny_vsm", 1)
    with pytest.raises(ValueError):
        head.logit(torch.zeros(0, cfg.feat_dim, device="cuda"))
    with pytest.raises(ValueError):
        head.logit(torch.zeros(4, cfg.feat_dim + 1, device="cuda"))
    with pytest.raises(HipPathUnavailable):
        head.logit(torch.zeros(4, cfg.feat_dim))


def test_batched_head_equals_one_video_at_a_time():
    """VideoScoreHead.logits: the videos of a group share every launch of the head whatever their lengths (row-wise Linears /
    LayerNorms, back-to-back sequences of their own lengths in vsc_attention_f32_varlen) -- bit for bit the logits of the
    one-at-a-time path, for mixed lengths, a video that fills all max_frames slots, and one longer than that."""
    from tools import synth
    from vsc_hip.video_score import VideoScoreHead
    from vsc_hip.vsm_config import get_vsm_config
    cfg = get_vsm_config("tiny_vsm")
    w = synth.vsm_weights(3, cfg)
    head = VideoScoreHead(cfg, w)
    lens = [5, 9, 5, cfg.max_frames, 1, 9, cfg.max_frames + 3, 5]
    vids = [torch.from_numpy(synth.normalish(40 + i, (n, cfg.feat_dim))).cuda() for i, n in enumerate(lens)]
    together = head.logits(vids)
    alone = torch.stack([head.logits([v])[0] for v in vids])
    assert together.shape == (len(lens),) and torch.equal(together, alone)
    assert torch.equal(head.logit(vids[1]), together[1])
