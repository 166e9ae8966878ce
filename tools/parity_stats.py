"""Error statistics of the end-to-end parity cases (max, mean |d|, signed mean of HIP - reference on L2-normalised descriptors):
the numbers the bounds in tests/test_gpu_encoder.py / tests/test_gpu_swin.py (DESC_MEAN_ATOL, DESC_BIAS_ATOL) are set from.
A max-only bound at 1e-3 lets a 1 % scale error in one layer through (it moves descriptors 3.5e-4); bf16 rounding noise
averages out over a descriptor, a systematic error does not.   python tools/parity_stats.py [> profiles/rNN_parity_stats.txt]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vsc22-submission_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import swin_oracle, vit_oracle  # noqa: E402  (checker only)
from tools import synth  # noqa: E402
from vsc_hip.config import get_config  # noqa: E402
from vsc_hip.encoder import HipEncoder  # noqa: E402
from vsc_hip.swin_config import get_swin_config  # noqa: E402
from vsc_hip.swin_encoder import SwinHipEncoder  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")


def stats(name, out, ref):
    d = out.astype(np.float64) - ref.astype(np.float64)
    print(f"{name:72s} max {np.abs(d).max():.3e}  mean|d| {np.abs(d).mean():.3e}  mean d {d.mean():+.3e}  "
          f"|mean d| per frame max {np.abs(d.mean(1)).max():.3e}", flush=True)


def main():
    for preset in ("tiny", "tiny_clip", "vit_b16_224", "vit_v68"):
        g = np.load(os.path.join(GOLDEN, f"vit_{preset}.npz"))
        cfg = get_config(preset)
        w = synth.encoder_weights(int(g["weights_seed"]), cfg)
        x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
        stats(f"vit golden {preset}", HipEncoder(cfg, w, max_batch=2, l2_normalize=True)(x).cpu().numpy(), g["desc_l2"])
    cfg = get_config("vit_b16_224")
    w = synth.encoder_weights(21, cfg)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    x = torch.from_numpy(synth.frames(22, 6, cfg))
    with torch.no_grad():
        ref = vit_oracle.descriptors(wt, cfg, x).numpy()
    stats("vit fresh inputs vs oracle", HipEncoder(cfg, w, max_batch=8, l2_normalize=True)(x.to(dev)).cpu().numpy(), ref)
    x = torch.from_numpy(synth.frames(23, 700, cfg))
    sample = [0, 1, 331, 332, 500, 663, 664, 699]
    with torch.no_grad():
        ref = vit_oracle.descriptors(wt, cfg, x[sample]).numpy()
    big = HipEncoder(cfg, w, max_batch=332, l2_normalize=True, lanes=2)(x.to(dev)).cpu().numpy()
    stats("vit benchmarked configuration vs oracle (8 frames)", big[sample], ref)
    small = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, lanes=1)(x.to(dev)).cpu().numpy()
    stats("vit benchmarked configuration vs small batches (700)", big, small)

    # what a small systematic error looks like in these statistics (the bounds must catch it): ONE weight tensor of the HIP encoder off
    g = np.load(os.path.join(GOLDEN, "vit_vit_b16_224.npz"))
    cfg = get_config("vit_b16_224")
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    for key, f in (("blocks.5.fc1.weight", 1.01), ("blocks.0.qkv.bias", 1.05), ("blocks.11.proj.weight", 1.01)):
        w = synth.encoder_weights(int(g["weights_seed"]), cfg)
        w[key] = (w[key] * f).astype(np.float32)
        stats(f"vit golden vit_b16_224, {key} x {f}", HipEncoder(cfg, w, max_batch=2, l2_normalize=True)(x).cpu().numpy(), g["desc_l2"])
    g = np.load(os.path.join(GOLDEN, "swin_swinv2_base_256.npz"))
    cfg = get_swin_config("swinv2_base_256")
    x = torch.from_numpy(synth.swin_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    for key, f in (("layers.2.blocks.5.mlp.fc1.weight", 1.01), ("layers.2.blocks.9.attn.proj.weight", 1.01), ("layers.0.blocks.0.mlp.fc2.bias", 1.05)):
        w = synth.swin_weights(int(g["weights_seed"]), cfg)
        w[key] = (w[key] * f).astype(np.float32)
        stats(f"swin golden swinv2_base_256, {key} x {f}", SwinHipEncoder(cfg, w, max_batch=3, l2_normalize=True)(x).cpu().numpy(), g["desc_l2"])

    for preset in ("tiny_swin", "tiny_swin_w8", "swinv2_base_256", "tiny_swin_w24", "swinv2_large_384"):
        g = np.load(os.path.join(GOLDEN, f"swin_{preset}.npz"))
        cfg = get_swin_config(preset)
        w = synth.swin_weights(int(g["weights_seed"]), cfg)
        x = torch.from_numpy(synth.swin_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
        stats(f"swin golden {preset}", SwinHipEncoder(cfg, w, max_batch=3, l2_normalize=True)(x).cpu().numpy(), g["desc_l2"])
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_weights(9, cfg)
    x = torch.from_numpy(synth.swin_frames(10, 520, cfg))
    sample = [0, 255, 256, 400, 511, 512, 519]
    with torch.no_grad():
        ref = swin_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    big = SwinHipEncoder(cfg, w, max_batch=256, l2_normalize=True)(x.to(dev)).cpu().numpy()
    stats("swin benchmarked configuration vs oracle (7 frames)", big[sample], ref)
    small = SwinHipEncoder(cfg, w, max_batch=4, l2_normalize=True)(x.to(dev)).cpu().numpy()
    stats("swin benchmarked configuration vs small batches (520)", big, small)


if __name__ == "__main__":
    main()
