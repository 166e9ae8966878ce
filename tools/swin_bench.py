"""Throughput of the Swin-V2 encoder (run on the GPU box): python tools/swin_bench.py [batch] [steps] [max_batch] [preset]
(batch > max_batch: the chunks alternate over the encoder's two lanes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
import torch

from tools import synth
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
MB = int(sys.argv[3]) if len(sys.argv) > 3 else B
PRESET = sys.argv[4] if len(sys.argv) > 4 else "swinv2_base_256"
cfg = get_swin_config(PRESET)
enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=MB, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
for _ in range(2):
    enc(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = enc(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{PRESET} batch {B}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} frames/s, "
      f"{cfg.flops_per_frame() * B / dt / 1e12:.1f} model TFLOP/s ({cfg.flops_per_frame() / 1e9:.1f} GFLOP/frame)")
