"""Video-score gate on the HIP path: ``MS`` of the reference (train/train_vid_score/video/model.py:63-99, traced into
vsm.torchscript.pt; called at infer/extract_query_feats.py:165-173) composed from libvsc_hip.so's building blocks.

The reference pads the CLIP [CLS] features to 256 frames and masks the padding.  A masked token is never a key and
never enters the pooling, so only the visible tokens are run: [CLS, frame_1 .. frame_n, first padded frame] at
positions 0..n+1 (the reference's mask is ``cat([ones(2), frame_mask])`` against [CLS, frames..., SEP], so the slot after
the last real frame is visible and SEP is hidden), or all 258 tokens when the video fills every frame.

Per layer (BERT post-LN): qkv Linear -> MHSA -> proj Linear (+x) -> LayerNorm -> fc1 Linear (+erf GELU) -> fc2 Linear (+x)
-> LayerNorm.  The head runs in FLOAT32 end to end: its sigmoid is compared with SCORE_THRESHOLD = 0.001
(extract_query_feats.py:53,172-174) and a bf16 pipeline through 12 post-LN layers moved the logit by up to 3e-2 (round 1) --
enough to flip a video that sits near the gate.  It is one video of <= 258 tokens at a time (44 GFLOP), so precision is
free: every Linear is vsc_conv2d_f32 as a 1x1 convolution (exact fp32 MFMA chains, bias / residual / GELU fused), attention
is vsc_attention_f32, LayerNorm is vsc_layernorm_f32 / vsc_ln_residual_f32.  No CPU fallback."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib, cnn, ops
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
        if cfg.hidden % cfg.heads:
            raise ValueError(f"hidden {cfg.hidden} is not a multiple of {cfg.heads} heads")
        dev = torch.device("cuda", torch.cuda.current_device())

        def f32(name):
            return torch.from_numpy(np.ascontiguousarray(weights[name], np.float32)).to(dev)

        def linear(w, b):
            """nn.Linear weight [out, in] + bias -> a packed 1x1 convolution on the fp32 MFMA tiles"""
            return cnn.Conv({"l.weight": np.asarray(w, np.float32)[:, :, None, None], "l.bias": np.asarray(b, np.float32)}, "l", None, 1, dev)

        self.proj = linear(weights["frame_proj.0.weight"], weights["frame_proj.0.bias"])
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
                "qkv": linear(qkv_w, qkv_b),
                "o": linear(weights[p + "attention.output.dense.weight"], weights[p + "attention.output.dense.bias"]),
                "ln1_g": f32(p + "attention.output.LayerNorm.weight"), "ln1_b": f32(p + "attention.output.LayerNorm.bias"),
                "fc1": linear(weights[p + "intermediate.dense.weight"], weights[p + "intermediate.dense.bias"]),
                "fc2": linear(weights[p + "output.dense.weight"], weights[p + "output.dense.bias"]),
                "ln2_g": f32(p + "output.LayerNorm.weight"), "ln2_b": f32(p + "output.LayerNorm.bias"),
            })
        self.out_w, self.out_b = f32("output_proj.weight"), f32("output_proj.bias")

    def logit(self, clip_cls: torch.Tensor) -> torch.Tensor:
        """clip_cls [n, feat_dim] (device): the CLIP [CLS] feature of each frame of ONE video -> 0-d logit tensor."""
        return self.logits([clip_cls])[0]

    def logits(self, videos) -> torch.Tensor:
        """One [n_v, feat_dim] device tensor per video -> [len(videos)] logits.  ALL the videos go through the head together, whatever
        their lengths: every Linear / LayerNorm is row-wise and takes all their tokens in one launch, the attention takes them as
        back-to-back sequences of their own lengths (vsc_attention_f32_varlen) -- the head of one 40-frame video is ~150 launches of a few
        microseconds of work each; a group of 26 query videos spent a quarter of its time issuing them one video after the other, and
        (round 5) a group of videos of 52 DIFFERENT lengths still did, because only equal lengths shared launches.  The rows of a video
        see the same fp32 fma chains whatever else is in the launch: the logits equal the one-at-a-time ones bit for bit."""
        cfg = self.cfg
        if not len(videos):
            return torch.empty(0, device=torch.device("cuda", torch.cuda.current_device()))
        for f in videos:
            if f.dim() != 2 or f.shape[1] != cfg.feat_dim or f.shape[0] == 0:
                raise ValueError(f"expected [n >= 1, {cfg.feat_dim}] features, got {tuple(f.shape)}")
            if not f.is_cuda:
                raise _lib.HipPathUnavailable("video-score features must be on the GPU (no CPU fallback)")
        dev = videos[0].device
        n_of = [min(int(f.shape[0]), cfg.max_frames) for f in videos]
        rows_sep = [compact_tokens(n, cfg.max_frames) for n in n_of]
        T_of = [rows + 1 + int(sep) for rows, sep in rows_sep]

        def lin(layer, x, act=None, residual=None):   # [R, in] -> [R, out] through the NHWC convolution entry point
            r = None if residual is None else residual.reshape(1, x.shape[0], 1, -1)
            return layer(x.contiguous().reshape(1, x.shape[0], 1, x.shape[1]), act=act, residual=r).reshape(x.shape[0], -1)

        # frame rows of every video (+ its one padding row, compact_tokens), projected in one launch
        zero = torch.zeros(1, cfg.feat_dim, device=dev)
        pieces = []
        for f, n, (rows, _) in zip(videos, n_of, rows_sep):
            pieces.append(f[:n].float())
            if rows > n:
                pieces.append(zero)
        vision = ops.layernorm(lin(self.proj, torch.cat(pieces)), self.proj_g, self.proj_beta, cfg.proj_ln_eps, out_f32=True)
        # token rows: [CLS] + frame rows (+ [SEP]) per video, + position / type embeddings of positions 0 .. T_v - 1
        off = np.concatenate([[0], np.cumsum(T_of)]).astype(np.int64)
        total = int(off[-1])
        vis_pos = np.concatenate([off[v] + 1 + np.arange(rows) for v, (rows, _) in enumerate(rows_sep)])
        sep_pos = np.asarray([off[v + 1] - 1 for v, (_, sep) in enumerate(rows_sep) if sep], dtype=np.int64)
        pos_ids = np.concatenate([np.arange(T) for T in T_of])
        idx = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
        emb = torch.empty(total, vision.shape[1], device=dev)
        emb[idx(off[:-1])] = self.cls_emb
        emb[idx(vis_pos)] = vision
        if len(sep_pos):
            emb[idx(sep_pos)] = self.sep_emb
        emb = (emb + self.pos_type[idx(pos_ids)]).contiguous()
        row_off = torch.from_numpy(off.astype(np.int32)).to(dev)
        x, _ = ops.ln_residual(emb, self.emb_g, self.emb_b, cfg.ln_eps)
        for L in self.layers:
            att = self._attention_varlen(lin(L["qkv"], x), row_off, len(videos), max(T_of))
            x, _ = ops.ln_residual(lin(L["o"], att, residual=x), L["ln1_g"], L["ln1_b"], cfg.ln_eps)
            h = lin(L["fc1"], x, act="gelu")
            x, _ = ops.ln_residual(lin(L["fc2"], h, residual=x), L["ln2_g"], L["ln2_b"], cfg.ln_eps)
        # pooling: [CLS] row | mean over the video's token rows -- per token count (the reduction of a [V, T, hidden] block, as for one video)
        out = [None] * len(videos)
        by_T = {}
        for v, T in enumerate(T_of):
            by_T.setdefault(T, []).append(v)
        for T, vs in by_T.items():
            rows_idx = idx(np.concatenate([off[v] + np.arange(T) for v in vs]))
            xt = x[rows_idx].reshape(len(vs), T, -1)
            pooled = torch.cat([xt[:, 0], xt.sum(dim=1) / (T + 1e-5)], dim=1)
            vals = (self.out_w[0] * pooled).sum(dim=1) + self.out_b[0]
            for j, v in enumerate(vs):
                out[v] = vals[j]
        return torch.stack(out)

    def _attention_varlen(self, qkv: torch.Tensor, row_off: torch.Tensor, seqs: int, max_tokens: int) -> torch.Tensor:
        lib = _lib.require_device()
        out = torch.empty((qkv.shape[0], self.cfg.hidden), dtype=torch.float32, device=qkv.device)
        _lib.check(lib.vsc_attention_f32_varlen(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(row_off), seqs, max_tokens, self.cfg.heads,
                                                self.cfg.hidden // self.cfg.heads, _lib.current_stream()))
        return out

    def score(self, clip_cls: torch.Tensor) -> float:
        return float(torch.sigmoid(self.logit(clip_cls)))
