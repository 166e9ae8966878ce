"""bf16-operand against fp16-operand library, same box, alternating (bench.py times its fp16 secondary AFTER the bf16 headline: whatever the
chip's clock does over a run is then booked on the operand type): ViT-B/16 2 x 332 frames and Swin-V2-B 2 x 256 frames per step.
    python tools/micro/operand_speed_ab.py [alternations] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
import torch

from tools import synth
from vsc_hip.config import get_config
from vsc_hip.encoder import HipEncoder
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder

alts = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
vcfg, scfg = get_config("vit_b16_224"), get_swin_config("swinv2_base_256")
vw, sw = synth.encoder_weights(7, vcfg), synth.swin_weights(5, scfg)
vx = torch.from_numpy(synth.frames(1000, 32, vcfg)).to(dev).repeat(21, 1, 1, 1)[:664].contiguous()
sx = torch.from_numpy(synth.swin_frames(1, 8, scfg)).to(dev).repeat(64, 1, 1, 1)[:512].contiguous()
enc = {p: (HipEncoder(vcfg, vw, max_batch=332, l2_normalize=True, precision=p), SwinHipEncoder(scfg, sw, max_batch=256, l2_normalize=True, precision=p))
       for p in ("bf16", "fp16")}
tot = {(p, m): [] for p in enc for m in ("vit", "swin")}
for a in range(alts):
    for p in (("bf16", "fp16") if a % 2 == 0 else ("fp16", "bf16")):
        for m, (e, x) in (("vit", (enc[p][0], vx)), ("swin", (enc[p][1], sx))):
            e(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                e(x)
            torch.cuda.synchronize()
            tot[(p, m)].append(steps * x.shape[0] / (time.perf_counter() - t0))
for m in ("vit", "swin"):
    b, f = tot[("bf16", m)], tot[("fp16", m)]
    print(f"{m}: bf16 operands {sum(b) / len(b):8.0f} frames/s ({min(b):.0f} .. {max(b):.0f}), fp16 operands {sum(f) / len(f):8.0f} ({min(f):.0f} .. {max(f):.0f}): "
          f"fp16 / bf16 = {sum(f) / sum(b):.4f}")
