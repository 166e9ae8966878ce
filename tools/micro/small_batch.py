"""Is a small encoder call launch-bound?  (run on the GPU box)  Host issue time vs wall time vs profiled kernel time of
8- and 32-frame ViT-B/16 calls: 1.60 / 2.54 ms per call, the host needs 0.53 / 0.85 ms to issue the 89 launches, so the
GPU is the limit and a hipGraph would not shorten the call."""
import os, sys, time
sys.path.insert(0, "/root/repo/vsc22-submission_amd")
sys.path.insert(0, "/root/repo")
import torch
from tools import synth
from vsc_hip.config import get_config
from vsc_hip.encoder import HipEncoder
cfg = get_config("vit_b16_224")
w = synth.encoder_weights(7, cfg)
for b in (8, 32):
    enc = HipEncoder(cfg, w, max_batch=b, l2_normalize=True)
    x = torch.from_numpy(synth.frames(1, b, cfg)).cuda()
    for _ in range(5): enc(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): enc(x)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    enc.set_profiling(True)
    for _ in range(10): enc(x)
    torch.cuda.synchronize()
    prof = enc.get_profile()
    ksum = sum(v[0] for v in prof.values()) / 10
    n = sum(v[1] for v in prof.values()) // 10
    print(f"batch {b}: wall {t_all/50*1e3:.3f} ms/step (host issue {t_issue/50*1e3:.3f} ms), kernel sum {ksum:.3f} ms over {n} launches")
    enc.close()
