"""Two Swin encoders (two workspaces) on two streams vs one after the other: what a second lane would buy the Swin path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from tools import synth
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = get_swin_config("swinv2_base_256")
w = synth.swin_weights(5, cfg)
encs = [SwinHipEncoder(cfg, w, max_batch=B, l2_normalize=True) for _ in range(2)]
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def step(par):
    if par:
        for e, s in zip(encs, streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                e(x)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
    else:
        for e in encs:
            e(x)
for par in (False, True, False, True):
    for _ in range(2): step(par)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step(par)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{'two streams' if par else 'sequential '}: {dt * 1e3:.2f} ms per 2 x {B} frames, {2 * B / dt:.0f} frames/s", flush=True)
