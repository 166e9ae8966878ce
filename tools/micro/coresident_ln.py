"""Does a light (<= 48-register, LDS-free) LayerNorm run BESIDE a persistent bf16 GEMM of another stream?  The GEMM holds every CU
with 8 waves x 232 VGPRs, leaving 48 registers per SIMD lane.  Times: GEMMs alone, LayerNorms alone, both on two streams.
    python tools/micro/coresident_ln.py [gemm: qkv|fc1|proj|fc2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import torch
from vsc_hip import _lib, ops

which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
M = 332 * 197
n, k, epi = {"qkv": (2304, 768, _lib.EPI_BF16), "fc1": (3072, 768, _lib.EPI_GELU_BF16), "proj": (768, 768, _lib.EPI_RESADD_F32),
             "fc2": (768, 3072, _lib.EPI_RESADD_F32)}[which]
dev = torch.device("cuda:0")
a = torch.randn(M, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(n, device=dev)
aux = torch.randn(M, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
x = torch.randn(M, 768, device=dev)
g, be = torch.randn(768, device=dev), torch.randn(768, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 48


def gemms():
    with torch.cuda.stream(sa):
        for _ in range(N):
            ops.gemm_bf16(a, w, b, epilogue=epi, aux=aux, out=aux)


def lns():
    with torch.cuda.stream(sb):
        for _ in range(N):
            ops.layernorm(x, g, be, 1e-6)


def timed(*fns):
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6 / N)
    return best


for _ in range(3):   # clocks and allocator warm
    gemms(); lns()
torch.cuda.synchronize()


for light in ("1", "0", "1", "0"):
    _lib.set_option("VSC_LN_LIGHT", None if light == "1" else "0")
    tg, tl, tb = timed(gemms), timed(lns), timed(gemms, lns)
    print(f"{which} LN_LIGHT={light}: gemm alone {tg:.1f} us, layernorm alone {tl:.1f} us, both streams {tb:.1f} us per pair "
          f"(sum {tg + tl:.1f}): hidden {tg + tl - tb:.1f} us", flush=True)
_lib.set_option("VSC_LN_LIGHT", None)
