"""vsc_swin_mlp_bf16 at Swin-V2-B's stage-0 / stage-1 shapes (256 frames): fused kernel time, against the byte floor
(x in / out, shadow in / out) and the two-GEMM path's measured times.  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import numpy as np
import torch
from vsc_hip import _lib
from vsc_hip._lib import check, ptr, current_stream
lib = _lib.require_device()
dev = torch.device("cuda:0")
abls = [None] + (sys.argv[1].split(",") if len(sys.argv) > 1 else [])
for abl in abls:
  _lib.set_option("VSC_SWIN_MLP_ABL", abl)
  print("ablation", abl, "(library built with -DVSC_MLP_ABLATION)" if abl else "")
  for m, c in ((256 * 4096, 128), (256 * 1024, 256), (256 * 256, 512)):
      x = torch.randn(m, c, device=dev)
      xb = x.to(torch.bfloat16)
      w1 = (torch.randn(4 * c, c, device=dev) * c ** -0.5).to(torch.bfloat16)
      w2 = (torch.randn(c, 4 * c, device=dev) * (4 * c) ** -0.5).to(torch.bfloat16)   # (timing only: any layout)
      b1, b2 = torch.zeros(4 * c, device=dev), torch.zeros(c, device=dev)
      g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
      ts = []
      for _ in range(7):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          check(lib.vsc_swin_mlp_bf16(ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(g), ptr(b), ptr(x), ptr(xb), m, c, 1e-5, current_stream()))
          e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
      t = sorted(ts)[len(ts) // 2]
      gb = m * c * (4 + 4 + 2 + 2) / 1e9
      print(f"m {m} c {c}: {t:.1f} us   {16.0 * m * c * c / t / 1e6:.0f} TF/s   bytes {gb:.2f} GB = {gb / t * 1e3:.2f} TB/s", flush=True)
