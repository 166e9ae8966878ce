"""Per-class HIP-event times of a Swin-V2 preset (one stream):  python tools/micro/swin_classes.py swinv2_large_384 [frames] [chunk]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder
name = sys.argv[1] if len(sys.argv) > 1 else "swinv2_large_384"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
MB = int(sys.argv[3]) if len(sys.argv) > 3 else B
cfg = get_swin_config(name)
enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=MB, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
for _ in range(2): enc(x)
enc.set_profiling(True)
for _ in range(3): enc(x)
prof = enc.profile()
tot = sum(ms for ms, _ in prof.values()) / 3
print(f"{name}: {B} frames in chunks of {MB}: {tot:.2f} ms per step on one stream = {B / tot * 1e3:.0f} frames/s")
for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if not n: continue
    st = int(k[1]) if k[0] == "s" and k[1].isdigit() else None
    tf = ""
    if st is not None and k.split(".")[1] in ("qkv", "proj_ln", "fc1", "fc2_ln"):
        C, R = cfg.dim(st), cfg.resolution(st)
        per = {"qkv": 6.0, "proj_ln": 2.0, "fc1": 8.0, "fc2_ln": 8.0}[k.split(".")[1]] * R * R * C * C
        tf = f"{per * B * 3 * cfg.depths[st] / (ms * 1e-3) / 1e12 * (n / (3.0 * cfg.depths[st] * ((B + MB - 1) // MB))) ** 0:7.0f} TF/s (unfused count)"
    print(f"  {k:14s} {ms / 3:8.3f} ms  {n // 3:3d} launches  {ms / n * 1e3:8.1f} us each  {tf}")
