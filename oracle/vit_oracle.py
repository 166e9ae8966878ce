"""CPU oracle for the frame -> descriptor encoder.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path (vsc22-submission_amd/) never does.

Plain PyTorch fp32 restatement of the encoder the reference runs per frame:

* backbone = HuggingFace ``ViTModel`` as instantiated by the reference's
  ``VIT`` backbone
  (VSC22-Descriptor-Track-1st/train/train_v115/vsc/baseline/model_factory/backbones/vit.py:27-30)
  -- a third-party dependency (transformers==4.27.0 in /root/reference/dockerfile)
  whose algorithm is restated here: conv patch embedding, [CLS] + learned
  position embedding, N pre-LN blocks (LN -> MHSA -> +res -> LN -> MLP(GELU)
  -> +res), final LayerNorm.
* descriptor head = GeM pooling over the token axis followed by a Linear
  (vit.py:43-54: ``x.clamp(min=eps).pow(p).mean(dim=1).pow(1/p)`` then
  ``output_proj``).
* the CLIP flavour (ln_pre, QuickGELU, bias-free patch conv, CLS readout)
  follows VSC22-Descriptor-Track-1st/train/train_vid_score/video/clip.py:22-161.
* emitted descriptors are L2-normalised row-wise
  (infer/extract_query_feats.py:178 ``normalize(self.single_infer(...))`` and
  vsc/baseline/score_normalization.py:84-88).

Pinning: tests/golden/vit_*.npz were produced by tests/golden/gen_vit_golden.py,
which runs ``transformers.ViTModel`` / ``transformers.CLIPVisionModel`` (the
reference's own third-party backbone implementation) in this container on the
deterministic weights and frames of tools/synth.py; tests/test_oracle_vit.py checks this
restatement against those vectors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _act(x, kind):
    if kind == "gelu":  # HF ViT 'gelu' == exact erf GELU
        return F.gelu(x)
    if kind == "quick_gelu":  # clip.py:22-25
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(kind)


def patchify(frames: torch.Tensor, patch: int) -> torch.Tensor:
    """[N,C,H,W] -> [N, (H/p)*(W/p), C*p*p] with k = c*p*p + py*p + px, i.e. the
    flattening order of a Conv2d weight [D, C, p, p] (stride == kernel)."""
    n, c, h, w = frames.shape
    gh, gw = h // patch, w // patch
    x = frames.reshape(n, c, gh, patch, gw, patch)
    x = x.permute(0, 2, 4, 1, 3, 5)  # n, gh, gw, c, py, px
    return x.reshape(n, gh * gw, c * patch * patch)


def encode_tokens(params: dict, cfg, frames: torch.Tensor) -> torch.Tensor:
    """Backbone forward -> last hidden state [N, T, D] (after the final LN)."""
    p = {k: v.float() for k, v in params.items()}
    frames = frames.float()
    n = frames.shape[0]
    d = cfg.width
    x = patchify(frames, cfg.patch_size) @ p["patch.weight"].reshape(d, -1).t()
    if "patch.bias" in p:
        x = x + p["patch.bias"]
    cls = p["cls"].reshape(1, 1, d).expand(n, 1, d)
    x = torch.cat([cls, x], dim=1) + p["pos"].reshape(1, -1, d)
    if cfg.pre_ln:  # clip.py:152
        x = _ln(x, p["ln_pre.weight"], p["ln_pre.bias"], cfg.ln_eps)
    h = cfg.heads
    dh = d // h
    t = x.shape[1]
    for i in range(cfg.layers):
        b = f"blocks.{i}."
        y = _ln(x, p[b + "ln1.weight"], p[b + "ln1.bias"], cfg.ln_eps)
        qkv = y @ p[b + "qkv.weight"].t() + p[b + "qkv.bias"]
        q, k, v = qkv.reshape(n, t, 3, h, dh).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        a = torch.softmax(s, dim=-1) @ v  # [n,h,t,dh]
        a = a.permute(0, 2, 1, 3).reshape(n, t, d)
        x = x + a @ p[b + "proj.weight"].t() + p[b + "proj.bias"]
        y = _ln(x, p[b + "ln2.weight"], p[b + "ln2.bias"], cfg.ln_eps)
        y = _act(y @ p[b + "fc1.weight"].t() + p[b + "fc1.bias"], cfg.act)
        x = x + y @ p[b + "fc2.weight"].t() + p[b + "fc2.bias"]
    return _ln(x, p["ln_post.weight"], p["ln_post.bias"], cfg.ln_eps)


def gem(tokens: torch.Tensor, p: float, eps: float = 1e-6) -> torch.Tensor:
    """vit.py:52-54."""
    return tokens.clamp(min=eps).pow(p).mean(dim=1).pow(1.0 / p)


def descriptors(params: dict, cfg, frames: torch.Tensor, l2: bool = True) -> torch.Tensor:
    """frames [N,C,H,W] -> descriptors [N, out_dim]."""
    tok = encode_tokens(params, cfg, frames)
    if getattr(cfg, "head_conv_dim", 0):
        # SSCD head, train/train_v68/vsc/baseline/model_factory/backbones/sscd.py:33-42:
        # x.transpose(1,2) -> Conv1d(D, 2048, 1) -> clamp(1e-6).pow(p).mean(tokens).pow(1/p)
        tok = tok @ params["head_conv.weight"].float().t() + params["head_conv.bias"].float()
    if cfg.pool == "gem":
        pooled = gem(tok, cfg.gem_p)
    elif cfg.pool == "cls":  # extract_query_feats.py:171 ``clip_model(...)[:, 0]``
        pooled = tok[:, 0]
    else:
        raise ValueError(cfg.pool)
    if "head.weight" in params:
        pooled = pooled @ params["head.weight"].float().t() + params["head.bias"].float()
    if l2:
        pooled = l2_normalize(pooled)
    return pooled


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """sklearn.preprocessing.normalize(x) (norm='l2', axis=1): rows with zero
    norm are left unchanged (sklearn sets their norm to 1)."""
    nrm = x.float().pow(2).sum(dim=1, keepdim=True).sqrt()
    nrm = torch.where(nrm == 0, torch.ones_like(nrm), nrm)
    return x / nrm
