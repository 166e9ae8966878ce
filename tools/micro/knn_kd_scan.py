"""whole-call rate of the top-k search over k and the descriptor width (65 536 queries x 1M references): looking for cliffs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
lib = _lib.require_device()
dev = torch.device("cuda:0")
nq, nr = 65536, 1_000_000
for d in (64, 128, 256, 384, 512, 513, 768, 1024, 2048):
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.randn(nr, d, generator=g, device=dev); ops.l2_normalize_(r)
    q = torch.randn(nq, d, generator=g, device=dev); ops.l2_normalize_(q)
    for k in ((1, 10, 100, 128, 200, 256, 384, 512, 1000) if d == 512 else (100,)):
        ops.knn_ip(q, r, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): ops.knn_ip(q, r, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        print(f"d = {d:4d}  k = {k:4d}: {ms:8.2f} ms, {2.0 * nq * nr * d / ms / 1e9:7.1f} TFLOP/s-equivalent, path {lib.vsc_knn_last_path()}", flush=True)
    del r, q
