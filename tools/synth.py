"""Deterministic synthetic frames / weights / descriptor banks.

There is no dataset or checkpoint on the GPU box, and torch's RNG streams differ
between CPU and GPU, so everything synthetic is produced by an integer hash
(splitmix64 over the element index) that gives bit-identical float32 values on
any machine.  Golden fixtures (tests/golden) are generated from, and checked
against, exactly these tensors.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 array, uniform in [lo, hi), exactly reproducible."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx ^ _splitmix64(np.full(1, seed, dtype=np.uint64)))
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))  # 24-bit mantissa: exact in f32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(seed: int, shape, std: float = 1.0) -> np.ndarray:
    """Sum of four uniforms, scaled to the requested std (bell-shaped, bounded)."""
    acc = np.zeros(int(np.prod(shape)), dtype=np.float64)
    for j in range(4):
        acc += uniform(seed * 4 + j, (acc.size,)).astype(np.float64)
    # var of one U(-1,1) = 1/3 -> sum of 4 has std sqrt(4/3)
    return (acc * (std / np.sqrt(4.0 / 3.0))).astype(np.float32).reshape(shape)


def frames(seed: int, n: int, cfg) -> np.ndarray:
    """[n, C, H, W] float32 in [-1, 1): what vit_transform (infer/src/transform.py:37-42,
    Normalize(0.5, 0.5)) hands to the model."""
    return uniform(seed, (n, cfg.channels, cfg.image_size, cfg.image_size))


def encoder_weights(seed: int, cfg) -> dict:
    """Random-init weights in the canonical naming of vsc_hip/weights.py.

    The q/k projections are drawn wider than HF's 0.02 so the softmax is not
    near-uniform (a flat softmax would hide attention bugs)."""
    d, m = cfg.width, cfg.mlp_dim
    s = seed * 1000
    w = {}

    def nxt():
        nonlocal s
        s += 1
        return s

    w["patch.weight"] = normalish(nxt(), (d, cfg.channels, cfg.patch_size, cfg.patch_size), 0.03)
    if cfg.patch_bias:
        w["patch.bias"] = normalish(nxt(), (d,), 0.02)
    w["cls"] = normalish(nxt(), (d,), 0.3)
    w["pos"] = normalish(nxt(), (cfg.tokens, d), 0.3)
    if cfg.pre_ln:
        w["ln_pre.weight"] = 1.0 + normalish(nxt(), (d,), 0.05)
        w["ln_pre.bias"] = normalish(nxt(), (d,), 0.05)
    for i in range(cfg.layers):
        b = f"blocks.{i}."
        w[b + "ln1.weight"] = 1.0 + normalish(nxt(), (d,), 0.05)
        w[b + "ln1.bias"] = normalish(nxt(), (d,), 0.05)
        qkv = normalish(nxt(), (3 * d, d), 0.03)
        qkv[: 2 * d] *= 1.5
        w[b + "qkv.weight"] = qkv
        w[b + "qkv.bias"] = normalish(nxt(), (3 * d,), 0.05)
        w[b + "proj.weight"] = normalish(nxt(), (d, d), 0.03)
        w[b + "proj.bias"] = normalish(nxt(), (d,), 0.02)
        w[b + "ln2.weight"] = 1.0 + normalish(nxt(), (d,), 0.05)
        w[b + "ln2.bias"] = normalish(nxt(), (d,), 0.05)
        w[b + "fc1.weight"] = normalish(nxt(), (m, d), 0.03)
        w[b + "fc1.bias"] = normalish(nxt(), (m,), 0.05)
        w[b + "fc2.weight"] = normalish(nxt(), (d, m), 0.02)
        w[b + "fc2.bias"] = normalish(nxt(), (d,), 0.02)
    w["ln_post.weight"] = 1.0 + normalish(nxt(), (d,), 0.05)
    w["ln_post.bias"] = normalish(nxt(), (d,), 0.05)
    if cfg.head_conv_dim:
        w["head_conv.weight"] = normalish(nxt(), (cfg.head_conv_dim, d), 0.05)
        w["head_conv.bias"] = normalish(nxt(), (cfg.head_conv_dim,), 0.3)
    if cfg.out_dim:
        w["head.weight"] = normalish(nxt(), (cfg.out_dim, cfg.head_conv_dim or d), 0.05)
        w["head.bias"] = normalish(nxt(), (cfg.out_dim,), 0.02)
    return w


def vit_outlier_weights(seed: int, cfg) -> dict:
    """`encoder_weights` with what trained ViTs have and random init lacks: ONE residual channel riding at ~100 from the first block on (the
    "massive activation" channel: a projection bias puts it there, nothing takes it out), LayerNorm gains x 20 in a few channels of every
    third block, and an fc1 bias that drives a few hidden units deep into GELU's linear range.  Index arithmetic only."""
    w = {k: v.copy() for k, v in encoder_weights(seed, cfg).items()}
    d, m = cfg.width, cfg.mlp_dim
    w["blocks.0.proj.bias"][5 % d] = 100.0
    for i in range(cfg.layers):
        b = f"blocks.{i}."
        if i % 3 == 0:
            for j in range(3):
                w[b + "ln1.weight"][(7 + 11 * j + 13 * i) % d] *= 20.0
                w[b + "ln2.weight"][(3 + 17 * j + 5 * i) % d] *= 20.0
        if i % 4 == 2:
            w[b + "fc1.bias"][(19 * i) % m] = 40.0
    return w


def descriptor_bank(seed: int, n: int, dim: int = 512, l2: bool = True) -> np.ndarray:
    """[n, dim] float32 bank; rows L2-normalised like emitted descriptors."""
    x = normalish(seed, (n, dim))
    if l2:
        x = x / np.maximum(np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True), 1e-30)
    return np.ascontiguousarray(x, dtype=np.float32)


def swin_weights(seed: int, cfg) -> dict:
    """Random-init Swin-V2 weights in the reference's own state-dict naming
    (train/train_v115/torch2scripts.py: patch_embed.proj, layers.S.blocks.B.attn.qkv, ...)."""
    s = seed * 100000
    w = {}

    def nxt():
        nonlocal s
        s += 1
        return s

    c0 = cfg.embed_dim
    w["patch_embed.proj.weight"] = normalish(nxt(), (c0, cfg.channels, cfg.patch_size, cfg.patch_size), 0.15)
    w["patch_embed.proj.bias"] = normalish(nxt(), (c0,), 0.05)
    w["patch_embed.norm.weight"] = 1.0 + normalish(nxt(), (c0,), 0.05)
    w["patch_embed.norm.bias"] = normalish(nxt(), (c0,), 0.05)
    for st in range(cfg.stages):
        c, h = cfg.dim(st), cfg.heads[st]
        for b in range(cfg.depths[st]):
            p = f"layers.{st}.blocks.{b}."
            w[p + "attn.qkv.weight"] = normalish(nxt(), (3 * c, c), 1.0 / np.sqrt(c))
            w[p + "attn.q_bias"] = normalish(nxt(), (c,), 0.1)
            w[p + "attn.v_bias"] = normalish(nxt(), (c,), 0.1)
            w[p + "attn.logit_scale"] = (np.log(10.0) + normalish(nxt(), (h, 1, 1), 0.3)).astype(np.float32)
            w[p + "attn.cpb_mlp.0.weight"] = normalish(nxt(), (512, 2), 0.5)
            w[p + "attn.cpb_mlp.0.bias"] = normalish(nxt(), (512,), 0.3)
            w[p + "attn.cpb_mlp.2.weight"] = normalish(nxt(), (h, 512), 0.08)
            w[p + "attn.proj.weight"] = normalish(nxt(), (c, c), 1.0 / np.sqrt(c))
            w[p + "attn.proj.bias"] = normalish(nxt(), (c,), 0.05)
            # res-post-norm: the LayerNorm gains scale each residual update (the reference
            # initialises them to 0; trained values are small)
            w[p + "norm1.weight"] = 0.3 + normalish(nxt(), (c,), 0.05)
            w[p + "norm1.bias"] = normalish(nxt(), (c,), 0.05)
            w[p + "mlp.fc1.weight"] = normalish(nxt(), (cfg.mlp_ratio * c, c), 1.0 / np.sqrt(c))
            w[p + "mlp.fc1.bias"] = normalish(nxt(), (cfg.mlp_ratio * c,), 0.1)
            w[p + "mlp.fc2.weight"] = normalish(nxt(), (c, cfg.mlp_ratio * c), 0.5 / np.sqrt(c))
            w[p + "mlp.fc2.bias"] = normalish(nxt(), (c,), 0.05)
            w[p + "norm2.weight"] = 0.3 + normalish(nxt(), (c,), 0.05)
            w[p + "norm2.bias"] = normalish(nxt(), (c,), 0.05)
        if st + 1 < cfg.stages:
            p = f"layers.{st}.downsample."
            w[p + "reduction.weight"] = normalish(nxt(), (2 * c, 4 * c), 0.5 / np.sqrt(c))
            w[p + "norm.weight"] = 1.0 + normalish(nxt(), (2 * c,), 0.05)
            w[p + "norm.bias"] = normalish(nxt(), (2 * c,), 0.05)
    cl = cfg.dim(cfg.stages - 1)
    w["norm.weight"] = 1.0 + normalish(nxt(), (cl,), 0.05)
    w["norm.bias"] = normalish(nxt(), (cl,), 0.05)
    w["output_proj.weight"] = normalish(nxt(), (cfg.out_dim, cl), 0.05)
    w["output_proj.bias"] = normalish(nxt(), (cfg.out_dim,), 0.02)
    return w


def swin_outlier_weights(seed: int, cfg) -> dict:
    """`swin_weights` with what trained checkpoints have and random init lacks: LayerNorm gains with a few channels x 20 (outlier
    channels of the residual updates), ONE residual channel at magnitude ~100 through the whole deepest stage, heads whose logit_scale
    sits at the reference's clamp (exp -> 100, torch2scripts.py:161), and a fc1 bias pushing a few hidden units deep into GELU's linear
    range.  Every choice by index arithmetic: the same dict on every machine."""
    w = {k: v.copy() for k, v in swin_weights(seed, cfg).items()}
    deepest = max(range(cfg.stages), key=lambda st: cfg.depths[st])
    w[f"layers.{deepest}.blocks.0.norm1.bias"][5] = 100.0     # enters the residual stream of the deepest stage and stays for all its blocks
    for st in range(cfg.stages):
        c = cfg.dim(st)
        for b in range(cfg.depths[st]):
            p = f"layers.{st}.blocks.{b}."
            if b % 3 == 0:                                   # every third block: three gains x 20 in each post-norm
                for j in range(3):
                    w[p + "norm1.weight"][(7 + 11 * j + 13 * b) % c] *= 20.0
                    w[p + "norm2.weight"][(3 + 17 * j + 5 * b) % c] *= 20.0
            if b % 2 == 1:                                   # every other block: half of the heads at the clamp
                w[p + "attn.logit_scale"][::2] = np.log(100.0) + 1.0
            if b % 4 == 2:
                w[p + "mlp.fc1.bias"][(19 * b) % (cfg.mlp_ratio * c)] = 40.0
    return w


def swin_frames(seed: int, n: int, cfg) -> np.ndarray:
    return uniform(seed, (n, cfg.channels, cfg.image_size, cfg.image_size))


def structured_frames(seed: int, n: int, cfg) -> np.ndarray:
    """[n, C, H, W] float32 in [-1, 1): frames that DIFFER from one another -- an oriented triangle wave of its own frequency, contrast
    and offset per frame and channel, a soft blob somewhere, a little noise.  Frames of i.i.d. noise (`frames`) all look alike to a
    network: their descriptors are near-collinear (cosine 0.98 between frames for the random-weight ViT-B/16), so a 1e-3 tolerance is
    4 % of the frame-to-frame signal; these give cosines of 0.67-0.9.  Only +, *, abs, floor, min/max in float64 on exactly
    representable grid values and `uniform` draws: the same array on every platform."""
    size, ch = cfg.image_size, cfg.channels
    par = uniform(seed * 7919 + 13, (n, 16)).astype(np.float64)              # per-frame parameters in [-1, 1)
    pch = uniform(seed * 7919 + 14, (n, ch, 2)).astype(np.float64)           # per-channel phase / offset weight
    noise = uniform(seed * 7919 + 15, (n, ch, size, size)).astype(np.float64)
    g = (np.arange(size, dtype=np.float64) + 0.5) / size
    y, x = g[:, None], g[None, :]
    out = np.empty((n, ch, size, size), np.float64)
    for i in range(n):
        k0 = (0.5 + 4.75 * (par[i, 0] + 1.0)) * (1.0 if par[i, 1] >= 0 else -1.0)      # 0.5 .. 10 cycles per image, either orientation
        k1 = (0.5 + 4.75 * (par[i, 2] + 1.0)) * (1.0 if par[i, 3] >= 0 else -1.0)
        amp = 0.15 + 0.4 * (par[i, 4] + 1.0)                                             # contrast 0.15 .. 0.95
        off = 0.4 * par[i, 5]
        cx, cy = 0.5 + 0.3 * par[i, 6], 0.5 + 0.3 * par[i, 7]
        r2 = 0.02 + 0.05 * (par[i, 8] + 1.0)
        blob = np.maximum(0.0, 1.0 - ((x - cx) ** 2 + (y - cy) ** 2) / r2) ** 2 * (0.8 * par[i, 9])
        for c in range(ch):
            t = k0 * x + k1 * y + 0.5 * pch[i, c, 0]
            tri = 4.0 * np.abs(t - np.floor(t + 0.5)) - 1.0                               # triangle wave in [-1, 1]
            out[i, c] = amp * tri + off * (0.65 + 0.35 * pch[i, c, 1]) + blob + 0.05 * noise[i, c]
    return np.minimum(np.maximum(out, -1.0), 0.999).astype(np.float32)


def vsm_weights(seed: int, cfg) -> dict:
    """Random-init video-score model weights in the reference's state-dict naming
    (train/train_vid_score/video/model.py:63-75 ``MS``: frame_proj, bert.*, output_proj)."""
    s = seed * 100000
    w = {}

    def nxt():
        nonlocal s
        s += 1
        return s

    h, m = cfg.hidden, cfg.mlp_dim
    w["frame_proj.0.weight"] = normalish(nxt(), (h, cfg.feat_dim), 1.0 / np.sqrt(cfg.feat_dim))
    w["frame_proj.0.bias"] = normalish(nxt(), (h,), 0.05)
    w["frame_proj.1.weight"] = 1.0 + normalish(nxt(), (h,), 0.05)
    w["frame_proj.1.bias"] = normalish(nxt(), (h,), 0.05)
    e = "bert.embeddings."
    w[e + "word_embeddings.weight"] = normalish(nxt(), (cfg.vocab, h), 0.5)
    w[e + "position_embeddings.weight"] = normalish(nxt(), (cfg.max_position, h), 0.3)
    w[e + "token_type_embeddings.weight"] = normalish(nxt(), (2, h), 0.3)
    w[e + "LayerNorm.weight"] = 1.0 + normalish(nxt(), (h,), 0.05)
    w[e + "LayerNorm.bias"] = normalish(nxt(), (h,), 0.05)
    for i in range(cfg.layers):
        p = f"bert.encoder.layer.{i}."
        for name in ("query", "key", "value"):
            w[p + f"attention.self.{name}.weight"] = normalish(nxt(), (h, h), 1.2 / np.sqrt(h))
            w[p + f"attention.self.{name}.bias"] = normalish(nxt(), (h,), 0.05)
        w[p + "attention.output.dense.weight"] = normalish(nxt(), (h, h), 1.0 / np.sqrt(h))
        w[p + "attention.output.dense.bias"] = normalish(nxt(), (h,), 0.05)
        w[p + "attention.output.LayerNorm.weight"] = 1.0 + normalish(nxt(), (h,), 0.05)
        w[p + "attention.output.LayerNorm.bias"] = normalish(nxt(), (h,), 0.05)
        w[p + "intermediate.dense.weight"] = normalish(nxt(), (m, h), 1.0 / np.sqrt(h))
        w[p + "intermediate.dense.bias"] = normalish(nxt(), (m,), 0.05)
        w[p + "output.dense.weight"] = normalish(nxt(), (h, m), 1.0 / np.sqrt(m))
        w[p + "output.dense.bias"] = normalish(nxt(), (h,), 0.05)
        w[p + "output.LayerNorm.weight"] = 1.0 + normalish(nxt(), (h,), 0.05)
        w[p + "output.LayerNorm.bias"] = normalish(nxt(), (h,), 0.05)
    w["output_proj.weight"] = normalish(nxt(), (1, 2 * h), 1.0 / np.sqrt(2 * h))
    w["output_proj.bias"] = normalish(nxt(), (1,), 0.05)
    return w
