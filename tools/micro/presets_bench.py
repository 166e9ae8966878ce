"""frames/s and model TF/s of every ViT-family preset at its aligned chunk on two lanes (python tools/micro/presets_bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip.config import aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
for name in ("vit_b16_224", "vit_v68", "clip_vit_l14_224"):
    cfg = get_config(name)
    t = (cfg.image_size // cfg.patch_size) ** 2 + 1
    mb = aligned_batch(t)
    enc = HipEncoder(cfg, synth.encoder_weights(3, cfg), max_batch=mb, l2_normalize=True, lanes=2)
    x = torch.from_numpy(synth.frames(1, 8, cfg)).cuda().repeat((2 * mb + 7) // 8, 1, 1, 1)[:2 * mb].contiguous()
    for _ in range(2): enc(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): enc(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    d, L, m = cfg.width, cfg.layers, cfg.mlp_dim
    fl = L * (2 * t * d * (3 * d + d + 2 * m) + 4 * t * t * d) + 2 * (t - 1) * d * 3 * cfg.patch_size ** 2
    print(f"{name}: tokens {t} chunk {mb}: {dt * 1e3:.2f} ms/step, {2 * mb / dt:.0f} frames/s, {fl * 2 * mb / dt / 1e12:.0f} model TF/s", flush=True)
    enc.close(); del enc, x; torch.cuda.empty_cache()
