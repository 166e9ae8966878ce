"""vsc_swin_proj_mlp_bf16 at Swin-V2-B's stage-2 shape (256 frames: 65 536 rows x 512): time of the one-launch second half of a block,
next to the MLP-only kernel and the proj_ln GEMM it absorbs.  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import numpy as np
import torch
from vsc_hip import _lib
from vsc_hip._lib import check, ptr, current_stream
lib = _lib.require_device()
dev = torch.device("cuda:0")
m, c = 256 * 256, 512
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(m, c, device=dev, generator=g)
xb = x.to(torch.bfloat16)
att = torch.randn(m, c, device=dev, generator=g).to(torch.bfloat16)
wp = (torch.randn(c, c, device=dev, generator=g) * c ** -0.5).to(torch.bfloat16)
w1 = (torch.randn(4 * c, c, device=dev, generator=g) * c ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(c, 4 * c, device=dev, generator=g) * (4 * c) ** -0.5).to(torch.bfloat16)
vec = lambda n, s=0.1: torch.randn(n, device=dev, generator=g) * s
bp, b1, b2 = vec(c), vec(4 * c), vec(c)
g1, be1, g2, be2 = 0.3 + vec(c, 0.05), vec(c, 0.05), 0.3 + vec(c, 0.05), vec(c, 0.05)


def timed(fn, n=11):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


t = timed(lambda: check(lib.vsc_swin_proj_mlp_bf16(ptr(att), ptr(wp), ptr(bp), ptr(g1), ptr(be1), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(g2), ptr(be2),
                                                   ptr(x), ptr(xb), m, c, 1e-5, current_stream())))
print(f"proj + MLP in one launch : {t[0]:7.1f} us (min {t[1]:.1f})   {(16.0 + 2.0) * m * c * c / t[0] / 1e6:.0f} TF/s")
t2 = timed(lambda: check(lib.vsc_swin_mlp_bf16(ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(g2), ptr(be2), ptr(x), ptr(xb), m, c, 1e-5, current_stream())))
print(f"MLP only                 : {t2[0]:7.1f} us (min {t2[1]:.1f})")
x2, xb2 = torch.empty_like(x), torch.empty_like(xb)
t3 = timed(lambda: check(lib.vsc_gemm_ln_bf16(ptr(att), ptr(wp), ptr(bp), ptr(g1), ptr(be1), ptr(x), ptr(x2), ptr(xb2), m, c, c, 1e-5, current_stream())))
print(f"proj + LayerNorm GEMM    : {t3[0]:7.1f} us (min {t3[1]:.1f})   -> two launches {t2[0] + t3[0]:.1f} us")
