"""Golden vector for `vit_v68` = timm ViT-B/32-384 backbone + the SSCD head (build container only):

    python tests/golden/gen_sscd_golden.py

The reference builds the model as SSCDModel(name="vit_base_patch32_384", pool="gem", pool_param=3., dims=(768, 512),
use_classify=False, add_head=True) (train/train_v68/torch2scripts.py:14): a timm ViT returning all tokens after its
final LayerNorm (global_pool='', num_classes=0; sscd.py:78) and `embeddings` = Sequential(GlobalGeMPool2d(3., (768, 512)),
nn.Linear(2048, 512)) (sscd.py:88-94).  timm is absent from this image, so the TOKENS come from transformers.ViTModel
configured as ViT-B/32-384 with timm's LayerNorm eps 1e-6 (the same pre-LN / exact-GELU / fused-qkv graph; HF's ViT is the
port of that very timm model), and the HEAD is the reference's own class, instantiated from its source
(_reference_classes.py) — the part of vit_v68 no other fixture covers.
Output: tests/golden/vit_vit_v68.npz {tokens_head, tokens_tail, desc, desc_l2}.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "vsc22-submission_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from tools import synth  # noqa: E402
from vsc_hip.config import get_config  # noqa: E402

WEIGHT_SEED, FRAME_SEED = 29, 31


def run(preset, weight_seed, frame_seed, n):
    from transformers import ViTConfig, ViTModel
    from check_golden_against_reference import sscd_head_reference
    from gen_vit_golden import _to_hf_vit_state
    cfg = get_config(preset)
    w = synth.encoder_weights(weight_seed, cfg)
    hf = ViTModel(ViTConfig(hidden_size=cfg.width, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            intermediate_size=cfg.mlp_dim, image_size=cfg.image_size, patch_size=cfg.patch_size,
                            layer_norm_eps=cfg.ln_eps, hidden_act="gelu"), add_pooling_layer=False).eval()
    missing, unexpected = hf.load_state_dict(_to_hf_vit_state(w, cfg), strict=True)
    assert not missing and not unexpected
    x = torch.from_numpy(synth.frames(frame_seed, n, cfg))
    with torch.no_grad():
        tok = hf(x).last_hidden_state
    return tok, sscd_head_reference(tok, w, cfg)


def main(preset="vit_v68", n=3):
    from sklearn.preprocessing import normalize
    tok, desc = run(preset, WEIGHT_SEED, FRAME_SEED, n)
    path = os.path.join(HERE, f"vit_{preset}.npz")
    np.savez_compressed(path, frames_seed=FRAME_SEED, weights_seed=WEIGHT_SEED, n_frames=n, desc=desc.numpy(),
                        desc_l2=normalize(desc.numpy()), tokens_head=tok[:, :4].numpy(), tokens_tail=tok[:, -2:].numpy())
    print(f"{path}: desc {tuple(desc.shape)} |desc| mean {desc.abs().mean():.4f} tok std {tok.std():.3f} "
          f"{os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
