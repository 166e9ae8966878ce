"""Shape of the video-score model ``MS`` (reference: train/train_vid_score/video/model.py:63-75,
config_vid_score.py): Linear+LN frame projection -> BERT encoder over [CLS, frames, SEP] -> [cls | mean] -> Linear."""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class VsmConfig:
    name: str = "vsm_roberta_base"
    feat_dim: int = 1024          # CLIP ViT-L/14 width (config_vid_score.py feat_dim)
    hidden: int = 768             # bert_dim
    layers: int = 12
    heads: int = 12               # head_dim must be 64 (the attention kernel's only head width)
    mlp_dim: int = 3072
    max_frames: int = 256
    max_position: int = 512
    vocab: int = 21128            # chinese-roberta-wwm-ext
    cls_id: int = 101             # model.py:87
    sep_id: int = 102
    ln_eps: float = 1e-12         # BERT layer_norm_eps
    proj_ln_eps: float = 1e-5     # nn.LayerNorm default of frame_proj.1


VSM_PRESETS = {
    "vsm_roberta_base": VsmConfig(),
    "tiny_vsm": VsmConfig(name="tiny_vsm", feat_dim=64, hidden=128, layers=2, heads=2, mlp_dim=256, max_frames=12,
                          max_position=16, vocab=128),
}


def get_vsm_config(name: str, **overrides) -> VsmConfig:
    cfg = VSM_PRESETS[name]
    return replace(cfg, **overrides) if overrides else cfg
