"""Golden vectors for the Swin-V2 encoder (run in the build container):

    python tests/golden/gen_swin_golden.py

transformers.Swinv2Model (third-party port of the Microsoft Swin-V2 code the reference inlines in
train/train_v115/torch2scripts.py) is loaded with the deterministic weights of tools/synth.py --
translated from the reference's parameter names to HF's -- and run on deterministic frames.  The
reference head (norm -> gem -> output_proj, torch2scripts.py:628-630) is applied on HF's
last_hidden_state (HF applies the final LayerNorm itself).
Outputs: tests/golden/swin_<preset>.npz {tokens_head [n,4,C], tokens_tail [n,2,C], desc, desc_l2}.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))

from tools import synth  # noqa: E402
from vsc_hip.swin_config import get_swin_config  # noqa: E402

WEIGHT_SEED, FRAME_SEED = 5, 13


def to_hf(w, cfg):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    st = {"embeddings.patch_embeddings.projection.weight": t(w["patch_embed.proj.weight"]),
          "embeddings.patch_embeddings.projection.bias": t(w["patch_embed.proj.bias"]),
          "embeddings.norm.weight": t(w["patch_embed.norm.weight"]), "embeddings.norm.bias": t(w["patch_embed.norm.bias"]),
          "layernorm.weight": t(w["norm.weight"]), "layernorm.bias": t(w["norm.bias"])}
    for s in range(cfg.stages):
        c = cfg.dim(s)
        for b in range(cfg.depths[s]):
            r, h = f"layers.{s}.blocks.{b}.", f"encoder.layers.{s}.blocks.{b}."
            qkv = w[r + "attn.qkv.weight"]
            st[h + "attention.self.query.weight"], st[h + "attention.self.key.weight"], st[h + "attention.self.value.weight"] = \
                t(qkv[:c]), t(qkv[c:2 * c]), t(qkv[2 * c:])
            st[h + "attention.self.query.bias"], st[h + "attention.self.value.bias"] = t(w[r + "attn.q_bias"]), t(w[r + "attn.v_bias"])
            st[h + "attention.self.logit_scale"] = t(w[r + "attn.logit_scale"])
            st[h + "attention.self.continuous_position_bias_mlp.0.weight"] = t(w[r + "attn.cpb_mlp.0.weight"])
            st[h + "attention.self.continuous_position_bias_mlp.0.bias"] = t(w[r + "attn.cpb_mlp.0.bias"])
            st[h + "attention.self.continuous_position_bias_mlp.2.weight"] = t(w[r + "attn.cpb_mlp.2.weight"])
            for src, dst in (("attn.proj", "attention.output.dense"), ("norm1", "layernorm_before"), ("norm2", "layernorm_after"),
                             ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense")):
                st[h + dst + ".weight"], st[h + dst + ".bias"] = t(w[r + src + ".weight"]), t(w[r + src + ".bias"])
        if s + 1 < cfg.stages:
            r, h = f"layers.{s}.downsample.", f"encoder.layers.{s}.downsample."
            st[h + "reduction.weight"] = t(w[r + "reduction.weight"])
            st[h + "norm.weight"], st[h + "norm.bias"] = t(w[r + "norm.weight"]), t(w[r + "norm.bias"])
    return st


def gen(preset, n):
    from sklearn.preprocessing import normalize
    from transformers import Swinv2Config, Swinv2Model
    cfg = get_swin_config(preset)
    w = synth.swin_weights(WEIGHT_SEED, cfg)
    hf = Swinv2Model(Swinv2Config(image_size=cfg.image_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim,
                                  depths=list(cfg.depths), num_heads=list(cfg.heads), window_size=cfg.window_size,
                                  pretrained_window_sizes=list(cfg.pretrained_window_sizes), mlp_ratio=float(cfg.mlp_ratio),
                                  layer_norm_eps=cfg.ln_eps, hidden_act="gelu"), add_pooling_layer=False).eval()
    res = hf.load_state_dict(to_hf(w, cfg), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    bad = [k for k in res.missing_keys if not any(s in k for s in ("relative_coords_table", "relative_position_index", "key.bias"))]
    assert not bad, bad
    x = torch.from_numpy(synth.swin_frames(FRAME_SEED, n, cfg))
    with torch.no_grad():
        tok = hf(pixel_values=x).last_hidden_state
        pooled = tok.clamp(min=1e-6).pow(cfg.gem_p).mean(dim=1).pow(1.0 / cfg.gem_p)
        desc = pooled @ torch.from_numpy(w["output_proj.weight"]).t() + torch.from_numpy(w["output_proj.bias"])
    path = os.path.join(HERE, f"swin_{preset}.npz")
    np.savez_compressed(path, weights_seed=WEIGHT_SEED, frames_seed=FRAME_SEED, n_frames=n,
                        tokens_head=tok[:, :4].numpy(), tokens_tail=tok[:, -2:].numpy(), desc=desc.numpy(),
                        desc_l2=normalize(desc.numpy()))
    print(f"{path}: tok {tuple(tok.shape)} std {tok.std():.3f} |desc| {desc.abs().mean():.3f} {os.path.getsize(path) / 1024:.1f} KiB")
    if preset == "swinv2_base_256":
        # a second fixture on frames that differ from one another (synth.structured_frames): descriptors far from collinear
        xs = torch.from_numpy(synth.structured_frames(FRAME_SEED, 6, cfg))
        with torch.no_grad():
            ts = hf(pixel_values=xs).last_hidden_state
            ps = ts.clamp(min=1e-6).pow(cfg.gem_p).mean(dim=1).pow(1.0 / cfg.gem_p)
            ds = normalize((ps @ torch.from_numpy(w["output_proj.weight"]).t() + torch.from_numpy(w["output_proj.bias"])).numpy())
        c = (ds @ ds.T)[np.triu_indices(6, 1)]
        np.savez_compressed(os.path.join(HERE, f"swin_{preset}_structured.npz"), weights_seed=WEIGHT_SEED, frames_seed=FRAME_SEED, n_frames=6,
                            desc_l2=ds, cos_min=float(c.min()), cos_mean=float(c.mean()))
        print(f"swin_{preset}_structured.npz: cosine between frames min {c.min():.3f} mean {c.mean():.3f}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen("tiny_swin", 3)
    gen("tiny_swin_w8", 3)
    gen("swinv2_base_256", 2)
    gen("tiny_swin_w24", 2)
    gen("swinv2_large_384", 1)
