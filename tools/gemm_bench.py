"""Micro-benchmark of vsc_gemm_bf16 on the encoder's shapes (run on the GPU box).
usage: python tools/gemm_bench.py [batch]   env: VSC_GEMM_V1=1, VSC_GEMM_TILE=128|256, VSC_GEMM_GROUP_N=g"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops

SWIN = len(sys.argv) > 1 and sys.argv[1] == "swin"
CUSTOM = len(sys.argv) > 4 and sys.argv[1] == "shape"
B = 256 if CUSTOM else (int(sys.argv[2 if SWIN else 1]) if len(sys.argv) > (2 if SWIN else 1) else 256)
M = B * 197
dev = torch.device("cuda:0")
shapes = [("qkv", M, 2304, 768, _lib.EPI_BF16), ("proj", M, 768, 768, _lib.EPI_RESADD_F32),
          ("fc1", M, 3072, 768, _lib.EPI_GELU_BF16), ("fc2", M, 768, 3072, _lib.EPI_RESADD_F32),
          ("patch", B * 196, 768, 768, _lib.EPI_BF16),
          ("fc1ng", M, 3072, 768, _lib.EPI_BF16), ("fc2nr", M, 768, 3072, _lib.EPI_BF16),
          ("sq4k", 4096, 4096, 4096, _lib.EPI_BF16), ("sq8k", 8192, 8192, 8192, _lib.EPI_BF16)]
if len(sys.argv) > 4 and sys.argv[1] == "shape":  # python tools/gemm_bench.py shape M N K
    shapes = [("custom", int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), _lib.EPI_BF16)]
    SWIN = False
if SWIN:  # Swin-V2-B/256 stages 1-3 at batch B: (tokens, width) = (B*4096, 128), (B*1024, 256), (B*256, 512)
    shapes = []
    for st, (t, c) in enumerate([(4096, 128), (1024, 256), (256, 512)], 1):
        shapes += [(f"s{st}qkv", B * t, 3 * c, c, _lib.EPI_BF16), (f"s{st}fc1", B * t, 4 * c, c, _lib.EPI_GELU_BF16)]
tot_f = tot_t = 0.0
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    aux = torch.randn(m, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
    out = aux if aux is not None else None
    for _ in range(3):
        ops.gemm_bf16(a, w, b, epilogue=epi, aux=aux, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.gemm_bf16(a, w, b, epilogue=epi, aux=aux, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    fl = 2.0 * m * n * k
    mult = 12 if name in ("qkv", "proj", "fc1", "fc2") else (1 if name == "patch" else 0)
    tot_f += fl * mult
    tot_t += us * mult
    print(f"{name:6s} M={m} N={n} K={k}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
if tot_t:
    print(f"weighted (12 layers): {tot_t / 1e3:.2f} ms per step, {tot_f / tot_t / 1e6:.1f} TF/s")
