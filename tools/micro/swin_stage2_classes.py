"""Per-class HIP-event times of Swin-V2-B's stage 2 (one stream, every kernel alone) under the switches given in the environment:
      [VSC_SWIN_QKV512=0] python tools/micro/swin_stage2_classes.py [frames] [chunk]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
import torch

from tools import synth
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = get_swin_config("swinv2_base_256")
enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=MB, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
for _ in range(2):
    enc(x)
enc.set_profiling(True)
for _ in range(3):
    enc(x)
prof = enc.profile()
total = sum(ms for ms, _ in prof.values())
for name, (ms, n) in sorted(prof.items()):
    if name.startswith("s2."):
        print(f"{name:14s} {n:4d} launches  {ms / n * 1e3:8.1f} us each  {ms / 3:8.3f} ms per step")
print(f"all classes: {total / 3:.3f} ms per step of {B} frames")
