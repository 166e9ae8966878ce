"""Prints the per-tile timeline of one workgroup of the persistent GEMM (library built with -DVSC_GEMM_TIMING)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
from vsc_hip import _lib as _vsc_lib
dev = torch.device("cuda:0")
M = 332 * 197
for name, m, n, k, epi in [("qkv", M, 2304, 768, _lib.EPI_BF16), ("fc2", M, 768, 3072, _lib.EPI_BF16), ("fc2r", M, 768, 3072, _lib.EPI_RESADD_F32),
                           ("fc1", M, 3072, 768, _lib.EPI_GELU_BF16)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    x = torch.randn(m, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
    for _ in range(3):
        ops.gemm_bf16(a, w, None, epilogue=epi, aux=x, out=x)
    torch.cuda.synchronize()
    _vsc_lib.set_option("VSC_GEMM_TIMING_PRINT", "1")
    for _ in range(2):
        ops.gemm_bf16(a, w, None, epilogue=epi, aux=x, out=x)
    _vsc_lib.set_option("VSC_GEMM_TIMING_PRINT", None)
