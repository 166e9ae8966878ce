"""Run-to-run equality of Swin-V2-B at 300 frames (256 + 44 on two lanes) under option sets -- which kernel choice is racy?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tools import synth
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder
from vsc_hip import _lib
dev = torch.device("cuda:0")
cfg = get_swin_config("swinv2_base_256")
w = synth.swin_weights(9, cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
x = torch.from_numpy(synth.swin_frames(10, n, cfg)).to(dev)
for opts in ({}, {"VSC_GEMM_LN_V4": "0"}, {"VSC_GEMM_V4": "0"}, {"VSC_SWIN_FUSED_MLP": "0"}, {}):
    for k, v in opts.items(): _lib.set_option(k, v)
    enc = SwinHipEncoder(cfg, w, max_batch=256, l2_normalize=True)
    outs = [enc(x).cpu().numpy() for _ in range(8)]
    bad = [(i, np.flatnonzero(np.abs(outs[i] - outs[0]).max(axis=1) > 0)[:8].tolist(), float(np.abs(outs[i] - outs[0]).max())) for i in range(1, 8) if not np.array_equal(outs[i], outs[0])]
    print(opts, "mismatching runs:", bad, flush=True)
    for k in opts: _lib.set_option(k, None)
    enc.close()
