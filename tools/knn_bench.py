"""Exact top-k sweep timing (run on the GPU box): python tools/knn_bench.py [nq] [nr] [k] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import ops

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev)
q = torch.randn(nq, 512, generator=g, device=dev)
ops.l2_normalize_(r)
ops.l2_normalize_(q)
ops.knn_ip(q, r, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    D, I = ops.knn_ip(q, r, k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f"nq={nq} nr={nr} k={k}: {ms:.2f} ms/sweep, {nq * nr / ms / 1e3:.0f} Mpairs/s, {2 * nq * nr * 512 / ms / 1e9:.1f} TFLOP/s-equivalent (2 nq nr d / call time)")
