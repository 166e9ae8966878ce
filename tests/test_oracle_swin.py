"""The Swin-V2 oracle against golden vectors produced by transformers.Swinv2Model
(tests/golden/gen_swin_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import swin_oracle
from tools import synth
from vsc_hip.swin_config import get_swin_config


@pytest.mark.parametrize("preset", ["tiny_swin", "tiny_swin_w8", "swinv2_base_256", "tiny_swin_w24"])
def test_swin_oracle_matches_transformers_golden(preset, golden_dir):
    g = np.load(os.path.join(golden_dir, f"swin_{preset}.npz"))
    cfg = get_swin_config(preset)
    w = {k: torch.from_numpy(v) for k, v in synth.swin_weights(int(g["weights_seed"]), cfg).items()}
    x = torch.from_numpy(synth.swin_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg))
    with torch.no_grad():
        tok = swin_oracle.encode_tokens(w, cfg, x).numpy()
        desc = swin_oracle.descriptors(w, cfg, x, l2=False).numpy()
        desc_l2 = swin_oracle.descriptors(w, cfg, x, l2=True).numpy()
    np.testing.assert_allclose(tok[:, :4], g["tokens_head"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(tok[:, -2:], g["tokens_tail"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(desc, g["desc"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(desc_l2, g["desc_l2"], atol=3e-5, rtol=0)


def test_shift_mask_and_bias_shapes():
    m = swin_oracle.shift_mask(32, 16, 8)
    assert m.shape == (4, 256, 256) and (m[0] == 0).all() and (m[3] == -100).any()
    assert swin_oracle.relative_position_index(8).max() == 15 * 15 - 1
    assert swin_oracle.relative_coords_table(16, 12).abs().max() > 1.0   # window 16 on a table pretrained at 12
