"""Swin-Transformer-V2 encoder configurations (the reference's swinv2_v106/v107/v115 models:
train/train_v115/torch2scripts.py:661-676 -- img 256, patch 4, window 16, embed 128,
depths [2,2,18,2], heads [4,8,16,32], pretrained windows [12,12,12,6], GeM(p=3) + Linear 512)."""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
class SwinConfig:
    name: str = "swinv2_base_256"
    image_size: int = 256
    patch_size: int = 4
    channels: int = 3
    embed_dim: int = 128
    depths: Tuple[int, ...] = (2, 2, 18, 2)
    heads: Tuple[int, ...] = (4, 8, 16, 32)
    window_size: int = 16
    pretrained_window_sizes: Tuple[int, ...] = (12, 12, 12, 6)
    mlp_ratio: int = 4
    out_dim: int = 512
    ln_eps: float = 1e-5
    gem_p: float = 3.0

    @property
    def stages(self) -> int:
        return len(self.depths)

    def resolution(self, stage: int) -> int:
        return self.image_size // self.patch_size // (2 ** stage)

    def dim(self, stage: int) -> int:
        return self.embed_dim * 2 ** stage

    def window(self, stage: int) -> int:
        """Effective window of a stage (clipped to the feature map: torch2scripts.py:218-221)."""
        return min(self.window_size, self.resolution(stage))

    def shift(self, stage: int, block: int) -> int:
        if self.resolution(stage) <= self.window_size:
            return 0
        return 0 if block % 2 == 0 else self.window_size // 2

    @property
    def desc_dim(self) -> int:
        return self.out_dim

    @property
    def patch_dim(self) -> int:
        return self.channels * self.patch_size * self.patch_size

    def flops_per_frame(self) -> int:
        f = 2 * self.resolution(0) ** 2 * self.patch_dim * self.embed_dim
        for s in range(self.stages):
            t, c, n = self.resolution(s) ** 2, self.dim(s), self.window(s) ** 2
            f += self.depths[s] * (2 * t * c * 12 * c + 2 * 2 * t * n * c)
            if s + 1 < self.stages:
                f += 2 * (t // 4) * 4 * c * 2 * c
        return f + 2 * self.dim(self.stages - 1) * self.out_dim


SWIN_PRESETS = {
    "swinv2_base_256": SwinConfig(),
    # parity-test sizes: 256-token shifted windows then one full 256-token window
    "tiny_swin": SwinConfig(name="tiny_swin", image_size=128, embed_dim=64, depths=(2, 2), heads=(2, 4),
                            window_size=16, pretrained_window_sizes=(12, 6), out_dim=64),
    # BASELINE.json configs[4] names a Swin-L 384 backbone; the reference ships none (its models are the swinv2_base_256 above).
    # This is Microsoft's Swin-V2-L at 384 (window 24, embed 192, heads 6/12/24/48) under the reference's head: 576-token
    # windows, clipped to 12 x 12 in the last stage.
    "swinv2_large_384": SwinConfig(name="swinv2_large_384", image_size=384, embed_dim=192, depths=(2, 2, 18, 2),
                                   heads=(6, 12, 24, 48), window_size=24, pretrained_window_sizes=(12, 12, 12, 6)),
    # parity-test size for those windows: 576-token shifted windows (2 x 2 of them), one full 24 x 24 window, a clipped 12 x 12 one
    "tiny_swin_w24": SwinConfig(name="tiny_swin_w24", image_size=192, embed_dim=64, depths=(2, 2, 2), heads=(2, 4, 8),
                                window_size=24, pretrained_window_sizes=(12, 12, 6), out_dim=64),
    # 64-token shifted windows, then a clipped 8x8 window
    "tiny_swin_w8": SwinConfig(name="tiny_swin_w8", image_size=64, embed_dim=64, depths=(2, 2), heads=(2, 4),
                               window_size=8, pretrained_window_sizes=(0, 0), out_dim=64),
}


def get_swin_config(name: str, **overrides) -> SwinConfig:
    cfg = SWIN_PRESETS[name]
    return replace(cfg, **overrides) if overrides else cfg
