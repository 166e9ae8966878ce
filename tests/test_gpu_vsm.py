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

LOGIT_ATOL = 3e-2   # bf16 operands through 2-12 post-LN layers; the gate compares sigmoid(logit) with 1e-3


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
        assert abs(head.score(f.cuda()) - vsm_oracle.video_score(w, cfg, f)) < 1e-2


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


def test_head_rejects_bad_input():
    from vsc_hip._lib import HipPathUnavailable
    cfg, w, head = _head("tiny_vsm", 1)
    with pytest.raises(ValueError):
        head.logit(torch.zeros(0, cfg.feat_dim, device="cuda"))
    with pytest.raises(ValueError):
        head.logit(torch.zeros(4, cfg.feat_dim + 1, device="cuda"))
    with pytest.raises(HipPathUnavailable):
        head.logit(torch.zeros(4, cfg.feat_dim))
