"""Encoder configurations for the frame -> descriptor path.

Each preset names the reference backbone it stands for:

* ``vit_b16_224``  -- the reference ``VIT`` backbone: HF ``ViTModel`` default
  config + GeM(p=3) + ``output_proj``
  (train/train_v115/vsc/baseline/model_factory/backbones/vit.py:10-54); 512-d
  descriptors as emitted by every model of infer/infer_ref.sh.
* ``vit_b32_384``  -- timm ``vit_base_patch32_384`` backbone of ``vit_v68``
  (train/train_v68/torch2scripts.py:10-15); ``vit_v68`` adds its SSCD head
  (768->2048 Conv1d over tokens, GeM, Linear 2048->512: sscd.py:25-42,88-94).
* ``clip_vit_l14_224`` -- the video-score CLIP tower
  (train/train_vid_score/video/clip.py:82-161, torch2scripts.py:7-9), CLS readout.
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class EncoderConfig:
    name: str = "vit_b16_224"
    image_size: int = 224
    patch_size: int = 16
    channels: int = 3
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    out_dim: int = 512          # 0 = no projection head
    ln_eps: float = 1e-12       # HF ViTConfig.layer_norm_eps
    act: str = "gelu"           # "gelu" | "quick_gelu"
    pre_ln: bool = False        # CLIP ln_pre
    patch_bias: bool = True
    pool: str = "gem"           # "gem" | "cls"
    gem_p: float = 3.0
    head_conv_dim: int = 0      # SSCD head: Conv1d(width, head_conv_dim, 1) before GeM (sscd.py:25-42)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def patch_dim(self) -> int:
        return self.channels * self.patch_size * self.patch_size

    @property
    def desc_dim(self) -> int:
        return self.out_dim if self.out_dim else self.width

    def flops_per_frame(self) -> int:
        """Algorithmic FLOPs (2*MAC) of one frame through the encoder."""
        t, d, m = self.tokens, self.width, self.mlp_dim
        f = 2 * (t - 1) * self.patch_dim * d
        per_layer = 2 * t * d * 3 * d + 2 * 2 * t * t * d + 2 * t * d * d + 2 * 2 * t * d * m
        f += self.layers * per_layer
        if self.head_conv_dim:
            f += 2 * t * d * self.head_conv_dim + 2 * self.head_conv_dim * self.out_dim
        elif self.out_dim:
            f += 2 * d * self.out_dim
        return f


PRESETS = {
    "vit_b16_224": EncoderConfig(),
    "vit_b32_384": EncoderConfig(name="vit_b32_384", image_size=384, patch_size=32,
                                 ln_eps=1e-6, out_dim=0),
    # the full vit_v68 model: backbone + SSCD head (train/train_v68/torch2scripts.py:14,
    # SSCDModel(pool="gem", pool_param=3., dims=(768, 512), add_head=True))
    "vit_v68": EncoderConfig(name="vit_v68", image_size=384, patch_size=32, ln_eps=1e-6, out_dim=512,
                             head_conv_dim=2048),
    "tiny_sscd": EncoderConfig(name="tiny_sscd", image_size=64, patch_size=16, width=128, layers=2, heads=2,
                               mlp_dim=512, out_dim=64, ln_eps=1e-6, head_conv_dim=256),
    "clip_vit_l14_224": EncoderConfig(name="clip_vit_l14_224", patch_size=14, width=1024,
                                      layers=24, heads=16, mlp_dim=4096, out_dim=0,
                                      ln_eps=1e-5, act="quick_gelu", pre_ln=True,
                                      patch_bias=False, pool="cls"),
    # small shapes for parity tests (head_dim stays 64, the only one the kernels take)
    "tiny": EncoderConfig(name="tiny", image_size=64, patch_size=16, width=128, layers=2,
                          heads=2, mlp_dim=512, out_dim=64),
    "tiny_clip": EncoderConfig(name="tiny_clip", image_size=64, patch_size=16, width=128,
                               layers=2, heads=2, mlp_dim=512, out_dim=0, ln_eps=1e-5,
                               act="quick_gelu", pre_ln=True, patch_bias=False, pool="cls"),
}


def get_config(name: str, **overrides) -> EncoderConfig:
    cfg = PRESETS[name]
    return replace(cfg, **overrides) if overrides else cfg


GEMM_ROUND_ROWS = 256 * 256   # 256-row tiles x 256 CUs: a GEMM over this many rows is a whole number of CU rounds


def aligned_batch(tokens_per_frame: int) -> int:
    """Largest frame count whose token rows fit one round of 256-row tiles on the 256 CUs: ViT-B/16 (197 tokens)
    332, ViT-B/32-384 (145) 451, CLIP ViT-L/14 (257) 255.  One frame more starts a 257th row tile, i.e. another
    (almost empty) round of every GEMM: measured 255 vs 256 frames on CLIP-L +3.4 %, 451 vs 256 on ViT-B/32-384 +6.5 %."""
    return max(GEMM_ROUND_ROWS // max(tokens_per_frame, 1), 1)
