"""CPU oracle for the video-score model.  TEST INFRASTRUCTURE ONLY (tests/ and smoke() may import it).

Plain PyTorch fp32 restatement of ``MS.forward`` (VSC22-Descriptor-Track-1st/train/train_vid_score/video/model.py:77-99)
as the reference traces it into vsm.torchscript.pt (train_vid_score/torch2scripts.py:17-31) and calls it on the
CLIP [CLS] features of a query video padded to 256 frames (infer/extract_query_feats.py:165-173):

  vision = LayerNorm(Linear(feats))                                model.py:79, :69
  masks  = feats.abs().sum(-1) > 0  (padded frames are all-zero)    :80
  x = BERT([emb(101), vision, emb(102)], attention_mask=[1, 1, masks])   :85-95
  logit = Linear([x[:, 0] | sum(x * mask) / (mask.sum() + 1e-5)])   :96-99

``bert`` is transformers' ``BertModel`` (a third-party dependency, chinese-roberta-wwm-ext-base): its published
algorithm is restated here -- embeddings = inputs_embeds + position_embeddings[0..T) + token_type_embeddings[0],
LayerNorm(eps 1e-12); L post-LN layers: x = LN(x + Wo·MHSA(x)), x = LN(x + W2·GELU(W1·x)), exact erf GELU, additive
key mask of dtype-min on masked positions.  Note the reference quirk kept as is: the mask passed to BERT is
``cat([ones(2), masks])`` -- aligned with [CLS, frame_1, ...], so the two always-valid slots are positions 0 and 1 and the
mask of frame i sits on position i + 1: the last frame's flag lands on SEP.

Pinning: tests/golden/vsm_tiny_vsm.npz comes from tests/golden/gen_vsm_golden.py, which builds transformers.BertModel
with the same deterministic weights and applies MS.forward's own lines; tests/test_oracle_vsm.py checks this file on it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def bert_encoder(p: dict, cfg, embeds: torch.Tensor, key_mask: torch.Tensor, position_ids=None) -> torch.Tensor:
    """embeds [B, T, H] (inputs_embeds), key_mask [B, T] (1 = attend) -> last hidden states [B, T, H]."""
    B, T, H = embeds.shape
    pos = torch.arange(T) if position_ids is None else position_ids
    e = "bert.embeddings."
    x = embeds + p[e + "position_embeddings.weight"][pos] + p[e + "token_type_embeddings.weight"][0]
    x = _ln(x, p[e + "LayerNorm.weight"], p[e + "LayerNorm.bias"], cfg.ln_eps)
    bias = (1.0 - key_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    hd = H // cfg.heads
    for i in range(cfg.layers):
        q = f"bert.encoder.layer.{i}."

        def lin(t, name):
            return F.linear(t, p[q + name + ".weight"], p[q + name + ".bias"])

        def heads(t):
            return t.reshape(B, T, cfg.heads, hd).transpose(1, 2)

        s = heads(lin(x, "attention.self.query")) @ heads(lin(x, "attention.self.key")).transpose(-1, -2) / math.sqrt(hd)
        a = torch.softmax(s + bias, dim=-1) @ heads(lin(x, "attention.self.value"))
        a = a.transpose(1, 2).reshape(B, T, H)
        x = _ln(x + lin(a, "attention.output.dense"), p[q + "attention.output.LayerNorm.weight"],
                p[q + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        h = F.gelu(lin(x, "intermediate.dense"))
        x = _ln(x + lin(h, "output.dense"), p[q + "output.LayerNorm.weight"], p[q + "output.LayerNorm.bias"], cfg.ln_eps)
    return x


def ms_forward(params: dict, cfg, feats: torch.Tensor) -> torch.Tensor:
    """feats [B, max_frames, feat_dim] (zero rows = padding) -> logits [B]; model.py:77-99 line by line."""
    p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in params.items()}
    vision = _ln(F.linear(feats, p["frame_proj.0.weight"], p["frame_proj.0.bias"]), p["frame_proj.1.weight"],
                 p["frame_proj.1.bias"], cfg.proj_ln_eps)
    masks = feats.abs().sum(dim=2).gt(0)
    bz = feats.shape[0]
    emb = p["bert.embeddings.word_embeddings.weight"]
    cls_emb, sep_emb = emb[cfg.cls_id].expand(bz, -1), emb[cfg.sep_id].expand(bz, -1)
    embeds = torch.cat([cls_emb[:, None], vision, sep_emb[:, None]], dim=1)
    masks = torch.cat([torch.ones((bz, 2)), masks.float()], dim=1)
    states = bert_encoder(p, cfg, embeds, masks)
    avg = (states * masks[..., None]).sum(dim=1) / (masks.sum(dim=1, keepdim=True) + 1e-5)
    cat = torch.cat([states[:, 0], avg], dim=1)
    return F.linear(cat, p["output_proj.weight"], p["output_proj.bias"]).squeeze(1)


def video_score(params: dict, cfg, clip_cls: torch.Tensor) -> float:
    """extract_query_feats.py:165-173: CLIP [CLS] features of the first <= 256 frames, zero-padded, -> sigmoid(logit)."""
    f = clip_cls[: cfg.max_frames].float()
    f = F.pad(f, (0, 0, 0, cfg.max_frames - f.shape[0]))
    return float(torch.sigmoid(ms_forward(params, cfg, f[None]))[0])
