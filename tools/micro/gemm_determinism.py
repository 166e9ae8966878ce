"""Repeatability of vsc_gemm_bf16 on the persistent kernel (and equality with the one-tile-per-workgroup kernel) on Swin / ViT shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib
if os.environ.get("VSC_TEST_LIB"):
    _lib.LIB_PATH = os.environ["VSC_TEST_LIB"]
from vsc_hip import ops
dev = torch.device("cuda:0")
shapes = [("s2.qkv", 65536, 1536, 512, _lib.EPI_BF16), ("s2.fc1", 65536, 2048, 512, _lib.EPI_GELU_BF16), ("vit.fc1", 65404, 3072, 768, _lib.EPI_GELU_BF16), ("s3.fc1", 16384, 4096, 1024, _lib.EPI_GELU_BF16), ("clip.fc1", 65535, 4096, 1024, _lib.EPI_QGELU_BF16), ("s0.qkv", 1048576, 384, 128, _lib.EPI_BF16),
          ("s1.qkv", 262144, 768, 256, _lib.EPI_BF16), ("vit.qkv", 65404, 2304, 768, _lib.EPI_BF16), ("vit.proj", 65404, 768, 768, _lib.EPI_RESADD_F32)]
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    aux0 = torch.randn(m, n, device=dev) if epi == _lib.EPI_RESADD_F32 else None
    def run():
        aux = aux0.clone() if aux0 is not None else None
        return ops.gemm_bf16(a, w, b, epilogue=epi, aux=aux, out=aux).clone()
    _lib.set_option("VSC_GEMM_V4", "0")
    ref = run()
    _lib.set_option("VSC_GEMM_V4", None)
    first = run()
    bad = 0
    for _ in range(30):
        o = run()
        bad += int(not torch.equal(o, first))
    d = (first.float() - ref.float()).abs()
    print(f"{name}: persistent vs one-tile kernel max|diff| {float(d.max()):.3e} ({int((d > 0).sum())} elements differ); {bad} of 30 repeats differ from the first", flush=True)
