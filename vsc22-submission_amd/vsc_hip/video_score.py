"""Video-score gate on the HIP path: ``MS`` of the reference (train/train_vid_score/video/model.py:63-99, traced into
vsm.torchscript.pt; called at infer/extract_query_feats.py:165-173) composed from libvsc_hip.so's building blocks.

The reference pads the CLIP [CLS] features to 256 frames and masks the padding.  A masked token is never a key and
never enters the pooling, so only the visible tokens are run: [CLS, frame_1 .. frame_n, first padded frame] at
positions 0..n+1 (the reference's mask is ``cat([ones(2), frame_mask])`` against [CLS, frames..., SEP], so the slot after
the last real frame is visible and SEP is hidden), or all 258 tokens when the video fills every frame.

Per layer (BERT post-LN): qkv GEMM -> fused MHSA -> proj GEMM (+x, fp32) -> LayerNorm -> fc1 GEMM (+erf GELU) ->
fc2 GEMM (+x, fp32) -> LayerNorm, i.e. vsc_gemm_bf16 / vsc_attention_bf16 / vsc_ln_residual_f32; bf16 operands,
fp32 accumulation, fp32 residual stream, like the frame encoders.  No CPU fallback."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib, ops
from .vsm_config import VsmConfig, get_vsm_config


def compact_tokens(n_frames: int, max_frames: int) -> Tuple[int, bool]:
    """-> (frame rows to project, whether SEP is visible) for a video with n_frames valid frames."""
    if n_frames >= max_frames:
        return max_frames, True
    return n_frames + 1, False


def from_reference_state(state: dict) -> Dict[str, np.ndarray]:
    """MS state dict (``module.`` prefix stripped as torch2scripts.py:21-27 does) -> float32 arrays."""
    out = {}
    for k, v in state.items():
        k = k[len("module."):] if k.startswith("module.") else k
        out[k] = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
    return out


class VideoScoreHead:
    def __init__(self, cfg: VsmConfig | str, weights: Dict[str, np.ndarray]):
        _lib.require_device()
        self.cfg = cfg = get_vsm_config(cfg) if isinstance(cfg, str) else cfg
        if cfg.hidden != cfg.heads * 64:
            raise ValueError(f"head_dim {cfg.hidden // cfg.heads} unsupported: the attention kernel takes 64")
        dev = torch.device("cuda", torch.cuda.current_device())

        def f32(name):
            return torch.from_numpy(np.ascontiguousarray(weights[name], np.float32)).to(dev)

        def bf16(name):
            return f32(name).to(torch.bfloat16).contiguous()

        self.proj_w, self.proj_b = bf16("frame_proj.0.weight"), f32("frame_proj.0.bias")
        self.proj_g, self.proj_beta = f32("frame_proj.1.weight"), f32("frame_proj.1.bias")
        e = "bert.embeddings."
        words = weights[e + "word_embeddings.weight"]
        self.cls_emb = torch.from_numpy(np.ascontiguousarray(words[cfg.cls_id], np.float32)).to(dev)
        self.sep_emb = torch.from_numpy(np.ascontiguousarray(words[cfg.sep_id], np.float32)).to(dev)
        self.pos_type = f32(e + "position_embeddings.weight") + f32(e + "token_type_embeddings.weight")[0]
        self.emb_g, self.emb_b = f32(e + "LayerNorm.weight"), f32(e + "LayerNorm.bias")
        self.layers = []
        for i in range(cfg.layers):
            p = f"bert.encoder.layer.{i}."
            qkv_w = np.concatenate([weights[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], axis=0)
            qkv_b = np.concatenate([weights[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], axis=0)
            self.layers.append({
                "qkv_w": torch.from_numpy(np.ascontiguousarray(qkv_w, np.float32)).to(dev).to(torch.bfloat16),
                "qkv_b": torch.from_numpy(np.ascontiguousarray(qkv_b, np.float32)).to(dev),
                "o_w": bf16(p + "attention.output.dense.weight"), "o_b": f32(p + "attention.output.dense.bias"),
                "ln1_g": f32(p + "attention.output.LayerNorm.weight"), "ln1_b": f32(p + "attention.output.LayerNorm.bias"),
                "fc1_w": bf16(p + "intermediate.dense.weight"), "fc1_b": f32(p + "intermediate.dense.bias"),
                "fc2_w": bf16(p + "output.dense.weight"), "fc2_b": f32(p + "output.dense.bias"),
                "ln2_g": f32(p + "output.LayerNorm.weight"), "ln2_b": f32(p + "output.LayerNorm.bias"),
            })
        self.out_w, self.out_b = f32("output_proj.weight"), f32("output_proj.bias")

    def logit(self, clip_cls: torch.Tensor) -> torch.Tensor:
        """clip_cls [n, feat_dim] (device): the CLIP [CLS] feature of each frame of ONE video -> 0-d logit tensor."""
        cfg = self.cfg
        if clip_cls.dim() != 2 or clip_cls.shape[1] != cfg.feat_dim or clip_cls.shape[0] == 0:
            raise ValueError(f"expected [n >= 1, {cfg.feat_dim}] features, got {tuple(clip_cls.shape)}")
        if not clip_cls.is_cuda:
            raise _lib.HipPathUnavailable("video-score features must be on the GPU (no CPU fallback)")
        rows, with_sep = compact_tokens(clip_cls.shape[0], cfg.max_frames)
        f = clip_cls[: cfg.max_frames].float()
        if rows > f.shape[0]:
            f = torch.cat([f, torch.zeros(rows - f.shape[0], cfg.feat_dim, device=f.device)])
        t = ops.gemm_bf16(f.to(torch.bfloat16), self.proj_w, self.proj_b, epilogue=_lib.EPI_F32)
        vision = ops.layernorm(t, self.proj_g, self.proj_beta, cfg.proj_ln_eps, out_f32=True)
        toks = [self.cls_emb[None], vision] + ([self.sep_emb[None]] if with_sep else [])
        emb = torch.cat(toks) + self.pos_type[: rows + 1 + int(with_sep)]
        T = emb.shape[0]
        x, xb = ops.ln_residual(emb, self.emb_g, self.emb_b, cfg.ln_eps)
        for L in self.layers:
            qkv = ops.gemm_bf16(xb, L["qkv_w"], L["qkv_b"])
            att = ops.attention_bf16(qkv, 1, T, cfg.heads)
            t = ops.gemm_bf16(att, L["o_w"], L["o_b"], epilogue=_lib.EPI_RESADD_F32, aux=x)
            x, xb = ops.ln_residual(t, L["ln1_g"], L["ln1_b"], cfg.ln_eps)
            h = ops.gemm_bf16(xb, L["fc1_w"], L["fc1_b"], epilogue=_lib.EPI_GELU_BF16)
            t = ops.gemm_bf16(h, L["fc2_w"], L["fc2_b"], epilogue=_lib.EPI_RESADD_F32, aux=x)
            x, xb = ops.ln_residual(t, L["ln2_g"], L["ln2_b"], cfg.ln_eps)
        pooled = torch.cat([x[0], x.sum(dim=0) / (T + 1e-5)])
        return (self.out_w[0] * pooled).sum() + self.out_b[0]

    def score(self, clip_cls: torch.Tensor) -> float:
        return float(torch.sigmoid(self.logit(clip_cls)))
