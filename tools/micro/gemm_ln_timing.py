"""Per-tile timeline of the persistent kernel's LN_RES launch (Swin stage-2 fc2: 65536 x 512 x 2048) from a -DVSC_GEMM_TIMING build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
dev = torch.device("cuda:0")
for name, m, n, k in (("s2.fc2", 65536, 512, 2048), ("s2.proj", 65536, 512, 512)):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias, g, b = torch.randn(n, device=dev) * 0.1, torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev) * 0.1
    x0 = torch.randn(m, n, device=dev)
    _lib.set_option("VSC_GEMM_LN_V4", "1")
    for _ in range(3):
        ops.gemm_ln_bf16(a, w, bias, g, b, 1e-5, x_in=x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_ln_bf16(a, w, bias, g, b, 1e-5, x_in=x0); e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) * 1e3:.1f} us", flush=True)
    _lib.set_option("VSC_GEMM_TIMING_PRINT", "1")
    ops.gemm_ln_bf16(a, w, bias, g, b, 1e-5, x_in=x0)
    torch.cuda.synchronize()
    _lib.set_option("VSC_GEMM_TIMING_PRINT", None)
_lib.set_option("VSC_GEMM_LN_V4", None)
