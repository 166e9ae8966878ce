import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
from vsc_hip import _lib as _vsc_lib
lib = _lib.require_device()
dev = torch.device("cuda:0")
out = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
def clock(load, label, n_launch):
    for _ in range(5):
        load()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    for _ in range(n_launch):
        load()
    m1.record()
    with torch.cuda.stream(side):
        e0.record()
        _lib.check(lib.vsc_debug_spin_ticks(12_000_000, out.data_ptr(), side.cuda_stream))
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print(f"{label:34s}: {int(out.item()) / us / 1e3:.3f} GHz   {m0.elapsed_time(m1) / n_launch * 1e3:8.1f} us/launch")
a = torch.randn(8192, 4096, device=dev).to(torch.bfloat16)
w = (torch.randn(8192, 4096, device=dev) * 0.05).to(torch.bfloat16)
o = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
for abl, label in ((0, "ours: full"), (1, "ours: no loop DMA"), (2, "ours: no MFMA"), (3, "ours: frag reads only"), (4, "ours: no epilogue stores"), (8, "ours: no frag reads (const ops)")):
    _vsc_lib.set_option("VSC_GEMM_ABL", str(abl))
    clock(lambda: ops.gemm_bf16(a, w, None), label, 60)
_vsc_lib.set_option("VSC_GEMM_ABL", "0")
clock(lambda: torch.matmul(a, w.t(), out=o), "hipBLASLt", 60)
for cfg in "CB":
    _vsc_lib.set_option("VSC_GEMM_CFG", cfg)
    clock(lambda: ops.gemm_bf16(a, w, None), f"ours cfg {cfg}", 60)
