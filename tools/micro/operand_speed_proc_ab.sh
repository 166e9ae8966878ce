#!/bin/bash
# Alternating PROCESSES, one library each: bf16 operands, fp16 operands, and an experimental fp16 build whose ACTIVATIONS keep 7 explicit
# significand bits (make EXTRA=-DVSC_LP_ACT_MANT=7 -> lib/libvsc_hip_f16_act7.so): frames/s of ViT-B/16 (2 x 332 frames per step) and the
# descriptor error against the golden fixture.  Run on the GPU box from the repo root.
for rep in 1 2 3; do
  for v in bf16 fp16 act7; do
    if [ $v = act7 ]; then export VSC_HIP_LIB_F16=$PWD/vsc22-submission_amd/lib/libvsc_hip_f16_act7.so; P=fp16; else unset VSC_HIP_LIB_F16; P=$v; fi
    python - $P $v <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, "vsc22-submission_amd"); sys.path.insert(0, ".")
from tools import synth
from vsc_hip.config import get_config
from vsc_hip.encoder import HipEncoder
p, tag = sys.argv[1], sys.argv[2]
cfg = get_config("vit_b16_224"); w = synth.encoder_weights(7, cfg); dev = torch.device("cuda:0")
enc = HipEncoder(cfg, w, max_batch=332, l2_normalize=True, precision=p)
x = torch.from_numpy(synth.frames(1000, 32, cfg)).to(dev).repeat(21, 1, 1, 1)[:664].contiguous()
for _ in range(3): enc(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): enc(x)
torch.cuda.synchronize(); fps = 30 * 664 / (time.perf_counter() - t0)
g = np.load("tests/golden/vit_vit_b16_224_structured.npz")
d = enc(torch.from_numpy(synth.structured_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)).cpu().numpy()
print(f"{tag}: {fps:8.0f} frames/s; structured fixture max {np.abs(d - g['desc_l2']).max():.2e} mean {np.abs(d - g['desc_l2']).mean():.2e}", flush=True)
PY
  done
done
