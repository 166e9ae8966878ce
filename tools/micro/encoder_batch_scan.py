"""frames/s of the ViT-B/16 and Swin-V2-B encoders over call sizes (looking for cliffs in the chunking / tile rules): an encoder built for
its aligned chunk is called with n frames, n from a handful to a few chunks.   (run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip.config import aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder


def rate(enc, x, n):
    xs = x[:n].contiguous()
    for _ in range(2): enc(xs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = max(3, min(30, 4000 // n))
    for _ in range(reps): enc(xs)
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t0)


cfg = get_config("vit_b16_224")
mb = aligned_batch(cfg.tokens)
enc = HipEncoder(cfg, synth.encoder_weights(3, cfg), max_batch=mb, l2_normalize=True, lanes=2)
x = torch.from_numpy(synth.frames(1, 8, cfg)).cuda().repeat(200, 1, 1, 1)
for n in (8, 40, 100, 166, 200, 300, mb, mb + 1, mb + 40, 500, 2 * mb, 2 * mb + 1, 800, 3 * mb, 1000, 4 * mb):
    print(f"vit_b16_224 (chunk {mb}) n = {n:5d}: {rate(enc, x, n):8.0f} frames/s", flush=True)
enc.close(); del enc, x; torch.cuda.empty_cache()
scfg = get_swin_config("swinv2_base_256")
enc = SwinHipEncoder(scfg, synth.swin_weights(5, scfg), max_batch=256, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, scfg)).cuda().repeat(130, 1, 1, 1)
for n in (8, 40, 100, 128, 200, 256, 257, 300, 384, 512, 513, 600, 768, 1024):
    print(f"swinv2_base_256 (chunk 256) n = {n:5d}: {rate(enc, x, n):8.0f} frames/s", flush=True)
