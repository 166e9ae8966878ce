"""oracle/vsm_oracle.py pinned on transformers.BertModel outputs (tests/golden/gen_vsm_golden.py), and the
compaction identity the HIP path relies on."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vsm_oracle  # noqa: E402

from tools import synth  # noqa: E402
from vsc_hip.vsm_config import get_vsm_config  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "vsm_tiny_vsm.npz")


def _case(cfg, g, i):
    n = int(g["n_valid"][i])
    f = torch.zeros(cfg.max_frames, cfg.feat_dim)
    f[:n] = torch.from_numpy(synth.normalish(int(g["feats_seed"]) + i, (n, cfg.feat_dim)))
    return n, f


def test_oracle_matches_transformers_golden():
    cfg = get_vsm_config("tiny_vsm")
    g = np.load(GOLD)
    w = synth.vsm_weights(int(g["weights_seed"]), cfg)
    feats = torch.stack([_case(cfg, g, i)[1] for i in range(len(g["n_valid"]))])
    logits = vsm_oracle.ms_forward(w, cfg, feats).numpy()
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-4, atol=2e-5)


def test_compact_sequence_is_equivalent():
    """Tokens the mask hides neither act as keys nor enter the pooling, so [CLS, valid..., first padded frame] with
    positions 0..n+1 (or the whole 258-token sequence when the video fills all frames) gives the same logit."""
    from vsc_hip.video_score import compact_tokens
    cfg = get_vsm_config("tiny_vsm")
    g = np.load(GOLD)
    w = {k: torch.from_numpy(v) for k, v in synth.vsm_weights(int(g["weights_seed"]), cfg).items()}
    for i in range(len(g["n_valid"])):
        n, f = _case(cfg, g, i)
        rows, with_sep = compact_tokens(n, cfg.max_frames)
        assert rows == min(n + 1, cfg.max_frames) and with_sep == (n == cfg.max_frames)
        vision = torch.nn.functional.layer_norm(torch.nn.functional.linear(f[:rows], w["frame_proj.0.weight"], w["frame_proj.0.bias"]),
                                                (cfg.hidden,), w["frame_proj.1.weight"], w["frame_proj.1.bias"], cfg.proj_ln_eps)
        emb = w["bert.embeddings.word_embeddings.weight"]
        toks = [emb[cfg.cls_id][None], vision] + ([emb[cfg.sep_id][None]] if with_sep else [])
        x = torch.cat(toks)[None]
        states = vsm_oracle.bert_encoder(w, cfg, x, torch.ones(1, x.shape[1]))
        cat = torch.cat([states[:, 0], states.sum(1) / (x.shape[1] + 1e-5)], dim=1)
        logit = torch.nn.functional.linear(cat, w["output_proj.weight"], w["output_proj.bias"])[0, 0]
        assert abs(float(logit) - float(g["logits"][i])) < 5e-5


def test_video_score_pads_and_truncates():
    cfg = get_vsm_config("tiny_vsm")
    w = synth.vsm_weights(3, cfg)
    f = torch.from_numpy(synth.normalish(5, (cfg.max_frames + 4, cfg.feat_dim)))
    full = vsm_oracle.video_score(w, cfg, f)
    assert full == vsm_oracle.video_score(w, cfg, f[: cfg.max_frames]) and 0.0 < full < 1.0
