"""Checkpoint -> HIP encoder, shared by the extract_* entry points.

A model is named ``arch:weights_format:checkpoint_path``: arch is a preset of vsc_hip.config (ViT family) or
vsc_hip.swin_config (Swin-V2); weights_format says how the checkpoint's parameters are named."""
from __future__ import annotations

from typing import Tuple

import torch

from vsc_hip import weights as W
from vsc_hip.config import PRESETS as VIT_PRESETS, aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
from vsc_hip.swin_config import SWIN_PRESETS, get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder, from_reference_state

VIT_LOADERS = {"hf_vit": W.from_hf_vit, "timm_vit": W.from_timm_vit, "clip": W.from_clip_visual}
WEIGHT_FORMATS = sorted(VIT_LOADERS) + ["swin_ref"]


def _state_dict(path: str) -> dict:
    """Parameters of a checkpoint: a plain state dict, a training checkpoint ({"state_dict": ...}), a pickled module,
    or the TorchScript archives the reference ships (checkpoints/*.torchscript.pt, written by torch2scripts.py) --
    a traced module keeps the parameter names of the nn.Module it was traced from."""
    try:
        scripted = torch.jit.load(path, map_location="cpu")
    except Exception:  # noqa: BLE001 -- not a TorchScript archive
        scripted = None
    if scripted is not None:
        return dict(scripted.state_dict())
    state = torch.load(path, map_location="cpu")
    if isinstance(state, dict):
        return state.get("state_dict", state)
    return state.state_dict()


DEFAULT_PRECISION = "fp16"   # of the infer/ entry points: the operand type whose end-to-end uAP lies within 1e-3 of the fp32 reference chain
                             # (tests/test_gpu_uap_e2e.py).  "bf16" is the configuration bench.py's headline times.


def load_encoder(arch: str, weights_format: str, checkpoint_path: str, max_batch: int = None, u8_norm=None,
                 precision: str = DEFAULT_PRECISION):
    """-> (encoder, image_size).  ``u8_norm`` = (mean, std) applied to uint8 frames on the GPU (default 0.5 / 0.5, the
    reference's vit_transform; pass dataset.CLIP_MEAN / CLIP_STD for the CLIP tower).  ``max_batch`` = None sizes the
    workspace for the architecture's tile-aligned batch (vsc_hip.config.aligned_batch).  ``precision``: "fp16" | "bf16" operands
    (vsc_hip.encoder.HipEncoder).
    Raises ValueError for an unknown arch / format pairing."""
    kw = {} if u8_norm is None else {"u8_mean": tuple(u8_norm[0]), "u8_std": tuple(u8_norm[1])}
    kw["precision"] = precision
    if arch in SWIN_PRESETS:
        if weights_format != "swin_ref":
            raise ValueError(f"{arch} is a Swin-V2 preset: weights_format must be swin_ref, not {weights_format}")
        cfg = get_swin_config(arch)
        deepest = max(range(cfg.stages), key=lambda st: cfg.depths[st])
        batch = max_batch or aligned_batch(cfg.resolution(deepest) ** 2)
        return SwinHipEncoder(cfg, from_reference_state(_state_dict(checkpoint_path)), max_batch=batch, **kw), cfg.image_size
    if arch in VIT_PRESETS:
        if weights_format not in VIT_LOADERS:
            raise ValueError(f"{arch} is a ViT preset: weights_format must be one of {sorted(VIT_LOADERS)}")
        cfg = get_config(arch)
        batch = max_batch or aligned_batch(cfg.tokens)
        return HipEncoder(cfg, VIT_LOADERS[weights_format](_state_dict(checkpoint_path), cfg), max_batch=batch, **kw), cfg.image_size
    raise ValueError(f"unknown arch {arch!r}; ViT presets {sorted(VIT_PRESETS)}, Swin presets {sorted(SWIN_PRESETS)}")


def parse_model_spec(spec: str) -> Tuple[str, str, str]:
    parts = spec.split(":", 2)
    if len(parts) != 3 or not all(parts):
        raise ValueError(f"model spec {spec!r} is not arch:weights_format:checkpoint_path")
    return parts[0], parts[1], parts[2]
