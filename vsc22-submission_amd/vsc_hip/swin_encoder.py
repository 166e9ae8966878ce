"""``SwinHipEncoder``: stands where the reference puts the swinv2_v1xx TorchScript backbones
(``torch.jit.load(...)`` / ``model(flat_frames)``: infer/extract_ref_feats.py:24-27,
infer/src/extractor.py:23).  Weights are given under the reference's own state-dict names
(train/train_v115/torch2scripts.py:677-684: ``module.backbone.`` prefix stripped)."""
from __future__ import annotations

import ctypes
import re

import numpy as np
import torch

from . import _lib
from ._lib import SwinConfigC, check, current_stream, ptr
from .config import aligned_batch
from .swin_config import SwinConfig, get_swin_config


def swin_weight_names(cfg: SwinConfig) -> list:
    names = ["patch_embed.proj.weight", "patch_embed.proj.bias", "patch_embed.norm.weight", "patch_embed.norm.bias"]
    for s in range(cfg.stages):
        for b in range(cfg.depths[s]):
            p = f"layers.{s}.blocks.{b}."
            names += [p + n for n in (
                "attn.qkv.weight", "attn.q_bias", "attn.v_bias", "attn.logit_scale", "attn.cpb_mlp.0.weight",
                "attn.cpb_mlp.0.bias", "attn.cpb_mlp.2.weight", "attn.proj.weight", "attn.proj.bias",
                "norm1.weight", "norm1.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias",
                "norm2.weight", "norm2.bias")]
        if s + 1 < cfg.stages:
            p = f"layers.{s}.downsample."
            names += [p + "reduction.weight", p + "norm.weight", p + "norm.bias"]
    return names + ["norm.weight", "norm.bias", "output_proj.weight", "output_proj.bias"]


def from_reference_state(state: dict) -> dict:
    """Strip ``module.`` / ``backbone.`` prefixes (torch2scripts.py:679-683); buffers are ignored."""
    out = {}
    for k, v in state.items():
        k = re.sub(r"^(module\.)?(backbone\.)?", "", k)
        out[k] = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
    return out


class SwinHipEncoder:
    def __init__(self, cfg: SwinConfig | str, weights: dict, *, max_batch: int = 32, l2_normalize: bool = False,
                 u8_mean=(0.5, 0.5, 0.5), u8_std=(0.5, 0.5, 0.5), precision: str = "bf16"):
        if isinstance(cfg, str):
            cfg = get_swin_config(cfg)
        self.cfg, self.max_batch = cfg, max_batch
        self.u8_mean = (ctypes.c_float * cfg.channels)(*u8_mean[: cfg.channels])   # Normalize() of uint8 inputs
        self.u8_std = (ctypes.c_float * cfg.channels)(*u8_std[: cfg.channels])
        self.precision = precision                  # 16-bit operand type: "bf16" | "fp16" (HipEncoder's docstring)
        self._lib = _lib.require_device(precision)
        names = swin_weight_names(cfg)
        missing = [n for n in names if n not in weights]
        if missing:
            raise KeyError(f"swin weights missing {len(missing)} tensors, e.g. {missing[:4]}")
        pad4 = lambda t: (ctypes.c_int32 * 4)(*(list(t) + [0] * (4 - len(t))))
        c = SwinConfigC(image_size=cfg.image_size, patch_size=cfg.patch_size, channels=cfg.channels,
                        embed_dim=cfg.embed_dim, stages=cfg.stages, depths=pad4(cfg.depths), heads=pad4(cfg.heads),
                        window_size=cfg.window_size, pretrained_window_sizes=pad4(cfg.pretrained_window_sizes),
                        mlp_ratio=cfg.mlp_ratio, out_dim=cfg.out_dim, ln_eps=cfg.ln_eps, gem_p=cfg.gem_p,
                        max_batch=max_batch, l2_normalize=int(l2_normalize))
        handle = ctypes.c_void_p()
        check(self._lib.vsc_swin_create(ctypes.byref(c), ctypes.byref(handle)))
        self._h = handle
        try:
            for name in names:
                arr = weights[name]
                arr = arr.detach().cpu().numpy() if isinstance(arr, torch.Tensor) else arr
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                check(self._lib.vsc_swin_set_weight(self._h, name.encode(), arr.ctypes.data_as(ctypes.c_void_p), arr.size))
            check(self._lib.vsc_swin_finalize(self._h))
        except Exception:
            self.close()
            raise

    def eval(self):
        return self

    def cuda(self, *_a, **_k):
        return self

    def to(self, *_a, **_k):
        return self

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.vsc_swin_workspace_bytes(self._h))

    @property
    def preferred_batch(self) -> int:
        """Frames per call that fill whole rounds of GEMM tiles in the deepest stage (where most of the time goes:
        swinv2_base_256 has 16 x 16 = 256 tokens per frame there -> 256 frames), within max_batch."""
        deepest = max(range(self.cfg.stages), key=lambda s: self.cfg.depths[s])
        return min(self.max_batch, aligned_batch(self.cfg.resolution(deepest) ** 2))

    @property
    def preferred_call(self) -> int:
        """Frames per CALL that keep both lanes busy (the chunks of a call alternate over the encoder's two lanes)."""
        return 2 * self.preferred_batch

    def __call__(self, frames: torch.Tensor, return_tokens: bool = False):
        """frames: float32 [n,C,H,W] normalised, or uint8 [n,H,W,C] decoded (normalisation fused on the GPU)."""
        cfg = self.cfg
        u8 = frames.dtype == torch.uint8
        want = (cfg.image_size, cfg.image_size, cfg.channels) if u8 else (cfg.channels, cfg.image_size, cfg.image_size)
        if frames.dim() != 4 or tuple(frames.shape[1:]) != want:
            raise ValueError(f"expected frames [n,{cfg.channels},{cfg.image_size},{cfg.image_size}] float32 or "
                             f"[n,{cfg.image_size},{cfg.image_size},{cfg.channels}] uint8, got {tuple(frames.shape)} {frames.dtype}")
        if not frames.is_cuda:
            raise _lib.HipPathUnavailable("frames must be on the GPU; there is no CPU path")
        frames = frames.contiguous() if u8 else frames.to(torch.float32).contiguous()
        n = frames.shape[0]
        desc = torch.empty((n, cfg.out_dim), dtype=torch.float32, device=frames.device)
        tokens = None
        if return_tokens:
            if u8:
                raise ValueError("return_tokens is a debug path of the float32 entry point")
            last = cfg.stages - 1
            tokens = torch.empty((n, cfg.resolution(last) ** 2, cfg.dim(last)), dtype=torch.float32, device=frames.device)
        if n and u8:
            check(self._lib.vsc_swin_forward_u8(self._h, ptr(frames), n, self.u8_mean, self.u8_std, ptr(desc), current_stream()))
        elif n:
            check(self._lib.vsc_swin_forward_debug(self._h, ptr(frames), n, ptr(desc), ptr(tokens), current_stream()))
        return (desc, tokens) if return_tokens else desc

    def set_profiling(self, on: bool) -> None:
        """Per-launch HIP events (vsc_swin_set_profiling); chunks then run back to back on the caller's stream."""
        check(self._lib.vsc_swin_set_profiling(self._h, int(on)))

    def profile(self) -> dict:
        """{class name: (ms, launches)} accumulated since profiling was switched on; stage classes are "s<stage>.<kind>"."""
        ms = (ctypes.c_double * _lib.SWIN_PROF_CLASSES)()
        cnt = (ctypes.c_int64 * _lib.SWIN_PROF_CLASSES)()
        check(self._lib.vsc_swin_get_profile(self._h, ms, cnt))
        names = ["patchify", "patch_embed", "pool_head"]
        for s in range(4):
            names += [f"s{s}.{k}" for k in _lib.SWIN_PROF_KINDS]
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(names) if cnt[i]}

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.vsc_swin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
