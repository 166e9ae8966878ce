"""A/B of the GEMM kernels on the encoder's shapes in ONE process, rounds interleaved (guide rule 24):
v3 (BK = 64 four-phase loop, one tile per workgroup), v4 (the same loop in a persistent kernel) and the library
(torch.matmul -> hipBLASLt) with the plain bf16 epilogue, then the fused epilogues v3 vs v4.  (run on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops
from vsc_hip import _lib as _vsc_lib

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 332
M = B * 197
ROUNDS = int(os.environ.get("ROUNDS", "5"))
shapes = [("qkv", M, 2304, 768, _lib.EPI_BF16), ("proj", M, 768, 768, _lib.EPI_RESADD_F32),
          ("fc1", M, 3072, 768, _lib.EPI_GELU_BF16), ("fc2", M, 768, 3072, _lib.EPI_RESADD_F32),
          ("sq4k", 4096, 4096, 4096, _lib.EPI_BF16), ("sq8k", 8192, 8192, 8192, _lib.EPI_BF16)]


def timeit(fn, it=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def own(v4, a, w, b, epi, aux):
    def f():
        _vsc_lib.set_option("VSC_GEMM_V4", "1" if v4 else "0")
        ops.gemm_bf16(a, w, b, epilogue=epi, aux=aux, out=aux)
    return f


tot = {"v3": 0.0, "v4": 0.0, "lib": 0.0, "v3e": 0.0, "v4e": 0.0}
totf = 0.0
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    b = torch.randn(n, device=dev)
    aux = torch.randn(m, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
    variants = {"v3": own(False, a, w, None, _lib.EPI_BF16, None), "v4": own(True, a, w, None, _lib.EPI_BF16, None),
                "lib": lambda: torch.matmul(a, w.t()), "libT": lambda: torch.matmul(a, wt),
                "v3e": own(False, a, w, b, epi, aux), "v4e": own(True, a, w, b, epi, aux)}
    best = {key: [] for key in variants}
    for _ in range(ROUNDS):
        for key, fn in variants.items():
            best[key].append(timeit(fn))
    med = {key: sorted(v)[len(v) // 2] for key, v in best.items()}
    med["lib"] = min(med["lib"], med.pop("libT"))
    fl = 2.0 * m * n * k
    mult = 12 if name in ("qkv", "proj", "fc1", "fc2") else 0
    totf += fl * mult
    for key in tot:
        tot[key] += med[key] * mult
    print(f"{name:5s} M={m} N={n} K={k}: " + "  ".join(f"{key} {med[key]:7.1f} us {fl / med[key] / 1e6:6.0f} TF" for key in ("v3", "v4", "lib", "v3e", "v4e")), flush=True)
print("weighted ViT-B/16 (12 layers): " + "  ".join(f"{key} {tot[key] / 1e3:.2f} ms {totf / tot[key] / 1e6:.0f} TF" for key in tot))
