"""The encoder oracle against the golden vectors produced by transformers'
ViTModel / CLIPVisionModel (tests/golden/gen_vit_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle
from tools import synth
from vsc_hip.config import get_config


def _run(preset, golden_dir):
    g = np.load(os.path.join(golden_dir, f"vit_{preset}.npz"))
    cfg = get_config(preset)
    w = {k: torch.from_numpy(v) for k, v in synth.encoder_weights(int(g["weights_seed"]), cfg).items()}
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg))
    with torch.no_grad():
        tok = vit_oracle.encode_tokens(w, cfg, x)
        desc = vit_oracle.descriptors(w, cfg, x, l2=False)
        desc_l2 = vit_oracle.descriptors(w, cfg, x, l2=True)
    return g, tok.numpy(), desc.numpy(), desc_l2.numpy()


@pytest.mark.parametrize("preset", ["tiny", "tiny_clip", "vit_b16_224", "vit_v68"])
def test_oracle_matches_transformers_golden(preset, golden_dir):
    g, tok, desc, desc_l2 = _run(preset, golden_dir)
    # fp32 vs fp32, different op order: 2e-4 abs on O(1) activations
    np.testing.assert_allclose(tok[:, :4], g["tokens_head"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(tok[:, -2:], g["tokens_tail"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(desc, g["desc"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(desc_l2, g["desc_l2"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(desc_l2, axis=1), 1.0, atol=1e-5)


def test_l2_normalize_zero_row_is_left_alone():
    x = torch.tensor([[3.0, 4.0], [0.0, 0.0]])
    y = vit_oracle.l2_normalize(x).numpy()
    np.testing.assert_allclose(y, [[0.6, 0.8], [0.0, 0.0]], atol=1e-7)
