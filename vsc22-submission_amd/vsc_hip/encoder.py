"""``HipEncoder``: the callable that stands where the reference puts its TorchScript
backbone -- ``model = torch.jit.load(...)``, ``flat_features = model(flat_frames)``
(infer/extract_ref_feats.py:24-27, infer/src/extractor.py:23,
infer/extract_query_feats.py:145-155).  Same call shape: frames [n,3,H,W] float32
on the GPU in, features [n, dim] float32 on the GPU out.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, weights as wnames
from ._lib import EncoderConfigC, check, current_stream, ptr
from .config import EncoderConfig, aligned_batch, get_config


class HipEncoder:
    def __init__(self, cfg: EncoderConfig | str, weights: dict, *, max_batch: int = 128,
                 l2_normalize: bool = False, lanes: int = 2, fuse_ln: int = 0,
                 u8_mean=(0.5, 0.5, 0.5), u8_std=(0.5, 0.5, 0.5), precision: str = "bf16"):
        """precision: the 16-bit type of the MFMA operands (weights and activations between the fp32 residual stream / LayerNorm /
        softmax): "bf16" (libvsc_hip.so, the benchmarked configuration) or "fp16" (libvsc_hip_f16.so: same kernels and speed, 8 x
        smaller rounding; what the infer/ entry points use)."""
        if isinstance(cfg, str):
            cfg = get_config(cfg)
        self.cfg = cfg
        self.max_batch = max_batch
        self.lanes = lanes
        self.l2 = l2_normalize
        # Normalize(mean, std) applied to uint8 [n,H,W,C] inputs inside the patchify kernel (vit_transform: 0.5 / 0.5)
        self.u8_mean = (ctypes.c_float * cfg.channels)(*u8_mean[: cfg.channels])
        self.u8_std = (ctypes.c_float * cfg.channels)(*u8_std[: cfg.channels])
        self.precision = precision
        self._lib = _lib.require_device(precision)
        wnames.check_complete(weights, cfg)
        c = EncoderConfigC(
            image_size=cfg.image_size, patch_size=cfg.patch_size, channels=cfg.channels,
            width=cfg.width, layers=cfg.layers, heads=cfg.heads, mlp_dim=cfg.mlp_dim,
            out_dim=cfg.out_dim, ln_eps=cfg.ln_eps, act={"gelu": 0, "quick_gelu": 1}[cfg.act],
            pre_ln=int(cfg.pre_ln), patch_bias=int(cfg.patch_bias),
            pool={"gem": 0, "cls": 1}[cfg.pool], gem_p=cfg.gem_p, max_batch=max_batch,
            l2_normalize=int(l2_normalize), head_conv_dim=cfg.head_conv_dim, lanes=lanes, fuse_ln=int(fuse_ln))
        handle = ctypes.c_void_p()
        check(self._lib.vsc_encoder_create(ctypes.byref(c), ctypes.byref(handle)))
        self._h = handle
        try:
            for name in wnames.canonical_names(cfg):
                arr = np.ascontiguousarray(
                    weights[name].detach().cpu().numpy() if isinstance(weights[name], torch.Tensor)
                    else weights[name], dtype=np.float32)
                check(self._lib.vsc_encoder_set_weight(
                    self._h, name.encode(), arr.ctypes.data_as(ctypes.c_void_p), arr.size))
            check(self._lib.vsc_encoder_finalize(self._h))
        except Exception:
            self.close()
            raise

    # nn.Module-ish surface the reference call sites use
    def eval(self):
        return self

    def cuda(self, *_a, **_k):
        return self

    def to(self, *_a, **_k):
        return self

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.vsc_encoder_workspace_bytes(self._h))

    @property
    def preferred_batch(self) -> int:
        """Frames per call that fill whole rounds of GEMM tiles (config.aligned_batch), within max_batch."""
        return min(self.max_batch, aligned_batch(self.cfg.tokens))

    @property
    def preferred_call(self) -> int:
        """Frames per CALL that keep every lane busy: the chunks of a call alternate over the encoder's lanes (two by default), so a
        call of one chunk runs on one lane alone (ViT-B/16: 332 frames 25.4 k frames/s, 664 frames 25.7 k; Swin-V2-B 16.1 vs 17.0 k)."""
        return self.preferred_batch * max(int(self.lanes), 1)

    def __call__(self, frames: torch.Tensor, return_tokens: bool = False):
        """frames: float32 [n,C,H,W] already normalised (the reference's tensors), or uint8 [n,H,W,C] decoded frames
        (ToTensor + Normalize(u8_mean, u8_std) then happen on the GPU; bit-identical descriptors, 4x fewer bytes)."""
        assert self._h is not None, "encoder was closed"
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
        desc = torch.empty((n, cfg.desc_dim), dtype=torch.float32, device=frames.device)
        tokens = None
        if return_tokens:
            if u8:
                raise ValueError("return_tokens is a debug path of the float32 entry point")
            tokens = torch.empty((n, cfg.tokens, cfg.width), dtype=torch.float32, device=frames.device)
        if n and u8:
            check(self._lib.vsc_encoder_forward_u8(self._h, ptr(frames), n, self.u8_mean, self.u8_std, ptr(desc), current_stream()))
        elif n:
            check(self._lib.vsc_encoder_forward_debug(self._h, ptr(frames), n, ptr(desc), ptr(tokens),
                                                      current_stream()))
        return (desc, tokens) if return_tokens else desc

    def set_profiling(self, on: bool) -> None:
        check(self._lib.vsc_encoder_set_profiling(self._h, int(on)))

    def get_profile(self) -> dict:
        """{class: (total_ms, launches)} accumulated since set_profiling(True)."""
        ms = (ctypes.c_double * len(_lib.PROF_CLASSES))()
        cnt = (ctypes.c_int64 * len(_lib.PROF_CLASSES))()
        check(self._lib.vsc_encoder_get_profile(self._h, ms, cnt))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.PROF_CLASSES)}

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.vsc_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
