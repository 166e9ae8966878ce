"""Run-to-run bit equality of the ViT-B/16 encoder as bench.py runs it (332-frame chunks, two lanes) and of the exact search.
(run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tools import synth
from vsc_hip import ops
from vsc_hip.config import get_config
from vsc_hip.encoder import HipEncoder
dev = torch.device("cuda:0")
cfg = get_config("vit_b16_224")
enc = HipEncoder(cfg, synth.encoder_weights(7, cfg), max_batch=332, l2_normalize=True)
x = torch.from_numpy(synth.frames(11, 700, cfg)).to(dev)
outs = [enc(x).cpu().numpy() for _ in range(6)]
print("ViT-B/16, 700 frames: mismatching runs", [i for i in range(1, 6) if not np.array_equal(outs[i], outs[0])], flush=True)
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(1_000_000, 512, generator=g, device=dev); q = torch.randn(70_000, 512, generator=g, device=dev)
ops.l2_normalize_(r); ops.l2_normalize_(q)
res = [tuple(t.cpu() for t in ops.knn_ip(q, r, 100)) for _ in range(4)]
print("kNN 70 000 x 1M top-100: mismatching runs", [i for i in range(1, 4) if not (torch.equal(res[i][0], res[0][0]) and torch.equal(res[i][1], res[0][1]))], flush=True)
