"""Generate the encoder golden vectors (run in the build container, NOT on the GPU box).

    python tests/golden/gen_vit_golden.py

The frame -> token backbone of the reference is a third-party module
(transformers.ViTModel, instantiated at
VSC22-Descriptor-Track-1st/train/train_v115/vsc/baseline/model_factory/backbones/vit.py:27-30;
its CLIP tower follows OpenAI CLIP, video/clip.py).  ``transformers`` is installed
in this container, so the vectors below are outputs of that implementation on the
deterministic weights/frames of tools/synth.py.  The reference wrapper itself
(``VIT``) cannot be imported here (its package imports mmcv), so its three
head lines -- gem (vit.py:52-54) and output_proj (vit.py:47-48) -- are applied
verbatim in ``_vit_head`` below.

Outputs (small, committed): tests/golden/vit_<preset>.npz with
  frames_seed, weights_seed, n_frames,
  pooled      [n, D]       GeM/CLS pooled features (before the Linear)
  desc        [n, out]     descriptors before L2 normalisation
  desc_l2     [n, out]     descriptors after sklearn-style L2 normalisation
  tokens_head [n, 4, D]    last_hidden_state[:, :4]
  tokens_tail [n, 2, D]    last_hidden_state[:, -2:]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))

from tools import synth  # noqa: E402
from vsc_hip.config import get_config  # noqa: E402

WEIGHT_SEED = 7
FRAME_SEED = 11


def _to_hf_vit_state(w, cfg):
    d = cfg.width
    st = {
        "embeddings.cls_token": w["cls"].reshape(1, 1, d),
        "embeddings.position_embeddings": w["pos"].reshape(1, -1, d),
        "embeddings.patch_embeddings.projection.weight": w["patch.weight"],
        "embeddings.patch_embeddings.projection.bias": w["patch.bias"],
        "layernorm.weight": w["ln_post.weight"],
        "layernorm.bias": w["ln_post.bias"],
    }
    for i in range(cfg.layers):
        b = f"blocks.{i}."
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            st[f"layers.{i}.attention.{nm}.weight"] = w[b + "qkv.weight"][j * d:(j + 1) * d]
            st[f"layers.{i}.attention.{nm}.bias"] = w[b + "qkv.bias"][j * d:(j + 1) * d]
        for src, dst in (("proj", "attention.o_proj"), ("ln1", "layernorm_before"),
                         ("ln2", "layernorm_after"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            st[f"layers.{i}.{dst}.weight"] = w[b + src + ".weight"]
            st[f"layers.{i}.{dst}.bias"] = w[b + src + ".bias"]
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}


def _vit_head(tokens, w, p):
    gem = tokens.clamp(min=1e-6).pow(p).mean(dim=1).pow(1.0 / p)       # vit.py:52-54
    pool = gem @ torch.from_numpy(w["head.weight"]).t() + torch.from_numpy(w["head.bias"])  # vit.py:47
    return gem, pool


def _l2(x):
    from sklearn.preprocessing import normalize  # what the reference calls on emitted descriptors
    return normalize(x)


def gen_vit(preset, n_frames):
    from transformers import ViTConfig, ViTModel
    cfg = get_config(preset)
    w = synth.encoder_weights(WEIGHT_SEED, cfg)
    hf = ViTModel(ViTConfig(hidden_size=cfg.width, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, intermediate_size=cfg.mlp_dim,
                            image_size=cfg.image_size, patch_size=cfg.patch_size,
                            layer_norm_eps=cfg.ln_eps, hidden_act="gelu"),
                  add_pooling_layer=False).eval()
    missing, unexpected = hf.load_state_dict(_to_hf_vit_state(w, cfg), strict=True)
    assert not missing and not unexpected
    x = torch.from_numpy(synth.frames(FRAME_SEED, n_frames, cfg))
    with torch.no_grad():
        tok = hf(x).last_hidden_state
        gem, desc = _vit_head(tok, w, cfg.gem_p)
    _save(preset, n_frames, tok, gem, desc)
    if preset == "vit_b16_224":
        # a second fixture on frames that differ from one another (synth.structured_frames): descriptors far from collinear
        xs = torch.from_numpy(synth.structured_frames(FRAME_SEED, 6, cfg))
        with torch.no_grad():
            _, ds = _vit_head(hf(xs).last_hidden_state, w, cfg.gem_p)
        d = _l2(ds.numpy())
        c = (d @ d.T)[np.triu_indices(6, 1)]
        np.savez_compressed(os.path.join(HERE, f"vit_{preset}_structured.npz"), frames_seed=FRAME_SEED, weights_seed=WEIGHT_SEED, n_frames=6,
                            desc_l2=d, cos_min=float(c.min()), cos_mean=float(c.mean()))
        print(f"vit_{preset}_structured.npz: cosine between frames min {c.min():.3f} mean {c.mean():.3f}")


def gen_clip(preset, n_frames):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = get_config(preset)
    w = synth.encoder_weights(WEIGHT_SEED, cfg)
    d = cfg.width
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=d, intermediate_size=cfg.mlp_dim,
                                          num_hidden_layers=cfg.layers,
                                          num_attention_heads=cfg.heads,
                                          image_size=cfg.image_size, patch_size=cfg.patch_size,
                                          hidden_act="quick_gelu", layer_norm_eps=cfg.ln_eps)).eval()
    st = {
        "embeddings.class_embedding": w["cls"],
        "embeddings.patch_embedding.weight": w["patch.weight"],
        "embeddings.position_embedding.weight": w["pos"],
        "pre_layrnorm.weight": w["ln_pre.weight"],
        "pre_layrnorm.bias": w["ln_pre.bias"],
        "post_layernorm.weight": w["ln_post.weight"],
        "post_layernorm.bias": w["ln_post.bias"],
    }
    for i in range(cfg.layers):
        b, t = f"blocks.{i}.", f"encoder.layers.{i}."
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            st[t + f"self_attn.{nm}.weight"] = w[b + "qkv.weight"][j * d:(j + 1) * d]
            st[t + f"self_attn.{nm}.bias"] = w[b + "qkv.bias"][j * d:(j + 1) * d]
        for src, dst in (("proj", "self_attn.out_proj"), ("ln1", "layer_norm1"),
                         ("ln2", "layer_norm2"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            st[t + dst + ".weight"] = w[b + src + ".weight"]
            st[t + dst + ".bias"] = w[b + src + ".bias"]
    st = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}
    res = hf.load_state_dict(st, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("position_ids" in k for k in res.missing_keys), res.missing_keys
    x = torch.from_numpy(synth.frames(FRAME_SEED, n_frames, cfg))
    with torch.no_grad():
        out = hf(pixel_values=x)
        cls = out.pooler_output  # post_layernorm(last_hidden_state[:, 0]) == clip.py:158 then [:, 0]
        # clip.py applies ln_post to every token; restate that on HF's pre-LN hidden state
        tok = torch.nn.functional.layer_norm(out.last_hidden_state, (d,),
                                             st["post_layernorm.weight"],
                                             st["post_layernorm.bias"], cfg.ln_eps)
    assert torch.allclose(tok[:, 0], cls, atol=1e-6)
    _save(preset, n_frames, tok, cls, cls)


def _save(preset, n, tok, pooled, desc):
    path = os.path.join(HERE, f"vit_{preset}.npz")
    np.savez_compressed(
        path, frames_seed=FRAME_SEED, weights_seed=WEIGHT_SEED, n_frames=n,
        pooled=pooled.numpy(), desc=desc.numpy(), desc_l2=_l2(desc.numpy()),
        tokens_head=tok[:, :4].numpy(), tokens_tail=tok[:, -2:].numpy())
    print(f"{path}: desc {tuple(desc.shape)} |desc| mean {desc.abs().mean():.4f} "
          f"tok std {tok.std():.3f}  {os.path.getsize(path)/1024:.1f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_vit("tiny", 5)
    gen_clip("tiny_clip", 3)
    gen_vit("vit_b16_224", 3)
