"""vsc_gemm_ln_bf16 on the Swin-V2-B shapes at 256 frames: row-owning tile kernel (VSC_GEMM_LN_V4=0) vs the persistent kernel's
LN_RES write-out (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
dev = torch.device("cuda:0")
shapes = [("s1.proj", 262144, 256, 256), ("s1.fc2", 262144, 256, 1024), ("s2.proj", 65536, 512, 512), ("s2.fc2", 65536, 512, 2048),
          ("s1.merge", 65536, 512, 1024)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias, g, b = torch.randn(n, device=dev) * 0.1, torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev) * 0.1
    x0 = torch.randn(m, n, device=dev)
    res = {}
    for mode in ("0", "1", "0", "1"):
        _lib.set_option("VSC_GEMM_LN_V4", mode)
        for _ in range(3):
            ops.gemm_ln_bf16(a, w, bias, g, b, 1e-5, x_in=x0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            x, xb = ops.gemm_ln_bf16(a, w, bias, g, b, 1e-5, x_in=x0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res[mode] = x
        print(f"{name} M={m} N={n} K={k} v4={mode}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.1f} TF/s", flush=True)
    print("   max |diff| between the kernels:", float((res["0"] - res["1"]).abs().max()))
_lib.set_option("VSC_GEMM_LN_V4", None)
