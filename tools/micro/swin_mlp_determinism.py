"""Run-to-run determinism of vsc_swin_mlp_bf16 on one input (an in-kernel race would show here; a cross-stream one only in
the encoder).  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for m, c in ((256 * 4096, 128), (256 * 1024, 256), (8 * 4096, 128), (1000, 256)):
    x0 = torch.randn(m, c, device=dev)
    w1 = torch.randn(4 * c, c) * c ** -0.5
    w2 = torch.randn(c, 4 * c) * (4 * c) ** -0.5
    b1, b2, g, b = torch.randn(4 * c) * 0.2, torch.randn(c) * 0.2, torch.ones(c), torch.zeros(c)
    outs = []
    for _ in range(6):
        x, xb = ops.swin_mlp_bf16(x0, w1, b1, w2, b2, g, b, 1e-5)
        outs.append(x.clone())
    bad = [(i, int((outs[i] != outs[0]).any(dim=1).sum()), float((outs[i] - outs[0]).abs().max())) for i in range(1, 6) if not torch.equal(outs[i], outs[0])]
    print(f"m {m} c {c}: mismatching runs (run, rows, max diff): {bad}", flush=True)
