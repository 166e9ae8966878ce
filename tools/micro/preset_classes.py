"""Per-class HIP-event times of one ViT-family preset at its aligned chunk (one stream):  python tools/micro/preset_classes.py clip_vit_l14_224 [frames per call]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip.config import aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
name = sys.argv[1] if len(sys.argv) > 1 else "clip_vit_l14_224"
cfg = get_config(name)
mb = int(sys.argv[2]) if len(sys.argv) > 2 else aligned_batch(cfg.tokens)
enc = HipEncoder(cfg, synth.encoder_weights(3, cfg), max_batch=mb, l2_normalize=True, lanes=2)
x = torch.from_numpy(synth.frames(1, 8, cfg)).cuda().repeat((mb + 7) // 8, 1, 1, 1)[:mb].contiguous()
for _ in range(2): enc(x)
enc.set_profiling(True)
for _ in range(3): enc(x)
torch.cuda.synchronize()
prof = enc.get_profile()
t, d, L, m = cfg.tokens, cfg.width, cfg.layers, cfg.mlp_dim
fl = {"gemm_qkv": 2 * t * d * 3 * d * L, "gemm_proj": 2 * t * d * d * L, "gemm_fc1": 2 * t * d * m * L, "gemm_fc2": 2 * t * d * m * L, "attention": 4 * t * t * d * L}
tot = sum(v[0] for v in prof.values()) / 3
print(f"{name}: tokens {t}, width {d}, chunk {mb}: {tot:.2f} ms per chunk = {mb / tot * 1e3:.0f} frames/s on one lane")
for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if not n: continue
    tf = f"{fl[k] * mb * 3 / (ms * 1e-3) / 1e12:7.0f} TF/s" if k in fl else ""
    print(f"  {k:14s} {ms / 3:8.3f} ms  {n // 3:3d} launches  {ms / n * 1e3:8.1f} us each  {tf}")
