"""A/B of VSC_SWIN_SUB (frames per sub-chunk of the 512-wide stage; 0 = whole chunk): throughput on two lanes and the
per-class launch times of stage 2 on one lane.   python tools/micro/swin_sub_ab.py [subs, default 0,128,64,32] [max_batch 256]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip import _lib
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder

subs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,128,64,32").split(",")]
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = 2 * MB
cfg = get_swin_config("swinv2_base_256")
enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=MB, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
ref = None
for rep in range(2):
    for sub in subs:
        _lib.set_option("VSC_SWIN_SUB", sub if sub else None)
        for _ in range(2):
            out = enc(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = enc(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if ref is None:
            ref = out.clone()
        same = bool((out == ref).all())
        line = f"sub {sub:4d}: {dt * 1e3:7.2f} ms per {B} frames = {B / dt:7.0f} frames/s  bit-identical to the first arm: {same}"
        if rep == 1:
            enc.set_profiling(True)
            enc(x)
            prof = enc.profile()
            enc.set_profiling(False)
            s2 = {k.split('.')[1]: ms / cnt * 1e3 * (MB / (sub if sub and sub < MB else MB)) for k, (ms, cnt) in prof.items() if k.startswith("s2.") and cnt}
            line += "   s2 us per 256 frames and block: " + " ".join(f"{k} {v:.1f}" for k, v in s2.items())
        print(line, flush=True)
_lib.set_option("VSC_SWIN_SUB", None)
