"""Checkpoint-name translation into the canonical weight naming of the HIP encoder.

Canonical names (all float32, PyTorch ``Linear`` layout ``[out, in]``):

    patch.weight [D,C,p,p]  patch.bias [D]   cls [D]   pos [T,D]
    ln_pre.{weight,bias}                      (CLIP only)
    blocks.{i}.ln1.{weight,bias}  blocks.{i}.qkv.{weight [3D,D],bias}
    blocks.{i}.proj.{weight,bias} blocks.{i}.ln2.{weight,bias}
    blocks.{i}.fc1.{weight,bias}  blocks.{i}.fc2.{weight,bias}
    ln_post.{weight,bias}   head.{weight [out,D],bias}
    head_conv.{weight [C,D],bias}             (SSCD head only; then head.weight is [out,C])

Sources understood (model loading is host-side Python, as in the reference):
  * HF ``ViTModel`` state dicts, both the transformers 4.27 naming the reference
    pins (``encoder.layer.N.attention.attention.query``) and the 5.x naming;
    optional ``vit.`` prefix and ``output_proj`` head of the reference ``VIT``
    wrapper (backbones/vit.py:30-31).
  * timm ``VisionTransformer`` (``blocks.N.attn.qkv`` ...), the ``vit_v68`` backbone.
  * the reference's CLIP tower (video/clip.py:98-110: ``conv1``,
    ``class_embedding``, ``transformer.resblocks.N.attn.in_proj_weight`` ...).
"""
from __future__ import annotations

import re

import numpy as np


def _np(x):
    if isinstance(x, np.ndarray):
        return x.astype(np.float32, copy=False)
    return x.detach().cpu().float().numpy()


def canonical_names(cfg) -> list:
    names = ["patch.weight"] + (["patch.bias"] if cfg.patch_bias else []) + ["cls", "pos"]
    if cfg.pre_ln:
        names += ["ln_pre.weight", "ln_pre.bias"]
    for i in range(cfg.layers):
        for part in ("ln1", "qkv", "proj", "ln2", "fc1", "fc2"):
            names += [f"blocks.{i}.{part}.weight", f"blocks.{i}.{part}.bias"]
    names += ["ln_post.weight", "ln_post.bias"]
    if cfg.head_conv_dim:
        names += ["head_conv.weight", "head_conv.bias"]
    if cfg.out_dim:
        names += ["head.weight", "head.bias"]
    return names


def check_complete(weights: dict, cfg) -> None:
    missing = [n for n in canonical_names(cfg) if n not in weights]
    if missing:
        raise KeyError(f"encoder weights missing {len(missing)} tensors, e.g. {missing[:4]}")


_HF_OLD = {
    "attention.attention.query": "q", "attention.attention.key": "k",
    "attention.attention.value": "v", "attention.output.dense": "proj",
    "intermediate.dense": "fc1", "output.dense": "fc2",
    "layernorm_before": "ln1", "layernorm_after": "ln2",
}
_HF_NEW = {
    "attention.q_proj": "q", "attention.k_proj": "k", "attention.v_proj": "v",
    "attention.o_proj": "proj", "mlp.fc1": "fc1", "mlp.fc2": "fc2",
    "layernorm_before": "ln1", "layernorm_after": "ln2",
}


def from_hf_vit(state: dict, cfg) -> dict:
    """HF ViTModel (optionally wrapped by the reference VIT module) -> canonical."""
    out, qkv = {}, {}
    for name, val in state.items():
        name = re.sub(r"^(module\.)?(backbone\.)?(vit\.)?", "", name)
        v = _np(val)
        if name == "embeddings.cls_token":
            out["cls"] = v.reshape(-1)
        elif name == "embeddings.position_embeddings":
            out["pos"] = v.reshape(-1, cfg.width)
        elif name.startswith("embeddings.patch_embeddings.projection."):
            out["patch." + name.rsplit(".", 1)[1]] = v
        elif name.startswith("layernorm."):
            out["ln_post." + name.rsplit(".", 1)[1]] = v
        elif name.startswith("output_proj."):
            out["head." + name.rsplit(".", 1)[1]] = v
        else:
            m = re.match(r"(?:encoder\.layer|layers)\.(\d+)\.(.+)\.(weight|bias)$", name)
            if not m:
                continue  # pooler etc.: not on the descriptor path
            i, mid, kind = int(m.group(1)), m.group(2), m.group(3)
            part = _HF_OLD.get(mid) or _HF_NEW.get(mid)
            if part is None:
                continue
            if part in ("q", "k", "v"):
                qkv[(i, part, kind)] = v
            else:
                out[f"blocks.{i}.{part}.{kind}"] = v
    for i in range(cfg.layers):
        for kind in ("weight", "bias"):
            if (i, "q", kind) in qkv:
                out[f"blocks.{i}.qkv.{kind}"] = np.concatenate(
                    [qkv[(i, "q", kind)], qkv[(i, "k", kind)], qkv[(i, "v", kind)]], axis=0)
    return out


def from_timm_vit(state: dict, cfg) -> dict:
    out = {}
    for name, val in state.items():
        name = re.sub(r"^(module\.)?(backbone\.)?(model\.backbone\.)?", "", name)
        v = _np(val)
        if name == "cls_token":
            out["cls"] = v.reshape(-1)
        elif name == "pos_embed":
            out["pos"] = v.reshape(-1, cfg.width)
        elif name.startswith("patch_embed.proj."):
            out["patch." + name.rsplit(".", 1)[1]] = v
        elif name.startswith("norm."):
            out["ln_post." + name.rsplit(".", 1)[1]] = v
        elif re.match(r"(model\.)?embeddings\.0\.conv\.(weight|bias)$", name):   # sscd.py:31 Conv1d(D, 2048, 1)
            out["head_conv." + name.rsplit(".", 1)[1]] = v.reshape(v.shape[0], -1) if v.ndim == 3 else v
        elif re.match(r"(model\.)?embeddings\.1\.(weight|bias)$", name):          # sscd.py:91 Linear(2048, out)
            out["head." + name.rsplit(".", 1)[1]] = v
        else:
            m = re.match(r"blocks\.(\d+)\.(norm1|attn\.qkv|attn\.proj|norm2|mlp\.fc1|mlp\.fc2)\.(weight|bias)$", name)
            if m:
                part = {"norm1": "ln1", "attn.qkv": "qkv", "attn.proj": "proj", "norm2": "ln2",
                        "mlp.fc1": "fc1", "mlp.fc2": "fc2"}[m.group(2)]
                out[f"blocks.{m.group(1)}.{part}.{m.group(3)}"] = v
    return out


def from_clip_visual(state: dict, cfg) -> dict:
    """The reference CLIPModel naming (video/clip.py)."""
    out = {}
    for name, val in state.items():
        name = re.sub(r"^(module\.)?(visual\.)?", "", name)
        v = _np(val)
        if name == "conv1.weight":
            out["patch.weight"] = v
        elif name == "class_embedding":
            out["cls"] = v.reshape(-1)
        elif name == "positional_embedding":
            out["pos"] = v.reshape(-1, cfg.width)
        elif name.startswith(("ln_pre.", "ln_post.")):
            out[name] = v
        else:
            m = re.match(r"transformer\.resblocks\.(\d+)\.(.+)$", name)
            if not m:
                continue
            i, rest = m.group(1), m.group(2)
            table = {
                "attn.in_proj_weight": "qkv.weight", "attn.in_proj_bias": "qkv.bias",
                "attn.out_proj.weight": "proj.weight", "attn.out_proj.bias": "proj.bias",
                "ln_1.weight": "ln1.weight", "ln_1.bias": "ln1.bias",
                "ln_2.weight": "ln2.weight", "ln_2.bias": "ln2.bias",
                "mlp.c_fc.weight": "fc1.weight", "mlp.c_fc.bias": "fc1.bias",
                "mlp.c_proj.weight": "fc2.weight", "mlp.c_proj.bias": "fc2.bias",
            }
            if rest in table:
                out[f"blocks.{i}.{table[rest]}"] = v
    return out
