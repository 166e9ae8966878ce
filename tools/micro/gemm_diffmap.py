import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
dev = torch.device("cuda:0")
m, n, k, epi = 65536, 2048, 512, _lib.EPI_GELU_BF16
torch.manual_seed(0)
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(n, device=dev)
_lib.set_option("VSC_GEMM_V4", "0"); ref = ops.gemm_bf16(a, w, b, epilogue=epi).float(); _lib.set_option("VSC_GEMM_V4", None)
for rep in range(3):
    o = ops.gemm_bf16(a, w, b, epilogue=epi).float()
    bad = (o != ref) | torch.isnan(o)
    idx = bad.nonzero()
    print("bad elements", len(idx), "nan", int(torch.isnan(o).sum()))
    if len(idx):
        r, c = idx[:, 0], idx[:, 1]
        print(" rows%256 hist (top):", torch.bincount(r % 256, minlength=256).topk(8))
        print(" cols%256 hist (top):", torch.bincount(c % 256, minlength=256).topk(8))
        print(" row tiles:", torch.unique(r // 256)[:20].tolist(), " col tiles:", torch.unique(c // 256).tolist())
        print(" sample:", [(int(r[i]), int(c[i]), float(o[r[i], c[i]]), float(ref[r[i], c[i]])) for i in range(0, len(idx), max(1, len(idx) // 8))][:8])
