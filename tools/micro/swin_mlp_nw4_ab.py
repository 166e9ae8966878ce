"""Stage-0 fused block (proj + LN + MLP + LN, C = 128) as two 4-wave workgroups per CU with a start stagger (VSC_SWIN_MLP_NW4=<stagger x 8128
cycles>) against the shipped one 8-wave workgroup per CU: per-launch HIP-event time of the s0.fc2_ln class, frames/s of the whole encoder, and
the descriptors' equality (same arithmetic per row).  Run on the GPU box: python tools/micro/swin_mlp_nw4_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
import torch

from tools import synth
from vsc_hip import _lib
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder

cfg = get_swin_config("swinv2_base_256")
enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=256, l2_normalize=True)
x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).cuda().repeat(64, 1, 1, 1)[:512].contiguous()
base = None
LONG = len(sys.argv) > 1 and sys.argv[1] == "long"     # long: default vs NW4=0 only, 16 steps each, five alternations
for rep in range(5 if LONG else 2):
    for nw4 in (("-1", None) if LONG else ("-1", None, "1", "2", "3", "4", "6")):     # "-1": the 8-wave form; None: the default (two 4-wave workgroups per CU)
        _lib.set_option("VSC_SWIN_MLP_NW4", nw4)
        for _ in range(2):
            out = enc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nst = 16 if LONG else 4
        for _ in range(nst):
            out = enc(x)
        torch.cuda.synchronize()
        fps = nst * 512 / (time.perf_counter() - t0)
        enc.set_profiling(True)
        for _ in range(2):
            enc(x)
        prof = enc.profile()
        enc.set_profiling(False)
        ms, cnt = prof["s0.fc2_ln"]
        if base is None:
            base = out.clone()
        print(f"NW4={nw4}: s0 fused block {1e3 * ms / cnt:7.1f} us per launch ({cnt} launches), encoder {fps:8.0f} frames/s, "
              f"max |d| vs default {float((out - base).abs().max()):.1e}", flush=True)
_lib.set_option("VSC_SWIN_MLP_NW4", None)
