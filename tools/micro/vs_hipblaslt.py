"""Calibration: the library GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, bf16 out, no epilogue) on the encoder's
shapes, next to vsc_gemm_bf16 with the plain bf16 store epilogue.  (run on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import ops

dev = torch.device("cuda:0")
M = 332 * 197
shapes = [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072),
          ("swin s3 qkv", 65536, 1536, 512), ("swin s3 fc1", 65536, 2048, 512), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192)]


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    us_lib = min(timeit(lambda: torch.matmul(a, w.t())), timeit(lambda: torch.matmul(a, wt)))
    us_own = timeit(lambda: ops.gemm_bf16(a, w, None))
    fl = 2.0 * m * n * k
    print(f"{name:12s} M={m} N={n} K={k}: library {us_lib:8.1f} us {fl / us_lib / 1e6:7.1f} TF/s | vsc_gemm_bf16 {us_own:8.1f} us {fl / us_own / 1e6:7.1f} TF/s")
