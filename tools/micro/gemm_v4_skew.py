"""Sweep of the persistent GEMM's start skew (VSC_GEMM_V4_SKEW=cycles,groups) on the encoder's four GEMMs, fused epilogues."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
from vsc_hip import _lib as _vsc_lib
dev = torch.device("cuda:0")
M = 332 * 197
SETTINGS = sys.argv[1:] or ["0,1", "1000,4", "2000,4", "3000,4", "1500,8", "3000,2", "6000,2", "12000,2", "2000,16", "1000,32"]

def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

for name, m, n, k, epi in [("qkv", M, 2304, 768, _lib.EPI_BF16), ("proj", M, 768, 768, _lib.EPI_RESADD_F32),
                           ("fc1", M, 3072, 768, _lib.EPI_GELU_BF16), ("fc2", M, 768, 3072, _lib.EPI_RESADD_F32)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    x = torch.randn(m, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
    res = {}
    for rnd in range(3):
        for st in SETTINGS + ["v3"]:
            _vsc_lib.set_option("VSC_GEMM_V4", "0" if st == "v3" else "1")
            _vsc_lib.set_option("VSC_GEMM_V4_SKEW", "0,1" if st == "v3" else st)
            res.setdefault(st, []).append(timeit(lambda: ops.gemm_bf16(a, w, b, epilogue=epi, aux=x, out=x)))
    print(f"{name:5s}", "  ".join(f"[{st}] {sorted(t)[1]:6.1f}" for st, t in res.items()), flush=True)
