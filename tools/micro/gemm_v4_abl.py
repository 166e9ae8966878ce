"""v3 / v4 with and without the write-out's global stores (VSC_GEMM_ABL=4): what the stores cost each kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
dev = torch.device("cuda:0")
M = 332 * 197

def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

for name, m, n, k in [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc2", M, 768, 3072), ("sq8k", 8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for v4 in "01":
            for abl in sys.argv[1:] or ["0", "4"]:
                os.environ["VSC_GEMM_V4"] = v4; os.environ["VSC_GEMM_ABL"] = abl
                res.setdefault((v4, abl), []).append(timeit(lambda: ops.gemm_bf16(a, w, None, epilogue=_lib.EPI_BF16, out=out)))
    print(name, "  ".join(f"v{3 + int(v4)} abl{abl} {sorted(t)[1]:7.1f} us" for (v4, abl), t in res.items()), flush=True)
