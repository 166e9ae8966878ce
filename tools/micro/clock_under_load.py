"""Shader clock while a kernel stream is busy (run on the GPU box): a one-wave spin of N s_memtime ticks on a second
stream, timed with events, beside back-to-back launches on the first.  python tools/micro/clock_under_load.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops

lib = _lib.require_device()
dev = torch.device("cuda:0")
out = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()


def clock(load, label, n_launch):
    for _ in range(5):
        load()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(n_launch):
        load()
    with torch.cuda.stream(side):
        e0.record()
        _lib.check(lib.vsc_debug_spin_ticks(20_000_000, out.data_ptr(), side.cuda_stream))
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print(f"{label:28s}: {int(out.item())} ticks in {us:9.1f} us -> {int(out.item()) / us / 1e3:.3f} GHz")


a = torch.randn(8192, 4096, device=dev).to(torch.bfloat16)
w = (torch.randn(8192, 4096, device=dev) * 0.05).to(torch.bfloat16)
z = torch.zeros(8192, 4096, device=dev, dtype=torch.bfloat16)
x = torch.randn(65404, 768, device=dev)
g = torch.ones(768, device=dev)
clock(lambda: None, "idle", 1)
clock(lambda: ops.gemm_bf16(a, w, None), "bf16 GEMM, random operands", 60)
clock(lambda: ops.gemm_bf16(z, z, None), "bf16 GEMM, zero operands", 60)
# the library's kernel on the same operands (calibration only): which clock does a 1.5 PF/s GEMM run at?
o = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
clock(lambda: torch.matmul(a, w.t(), out=o), "hipBLASLt, random operands", 60)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("ours", lambda: ops.gemm_bf16(a, w, None)), ("hipBLASLt", lambda: torch.matmul(a, w.t(), out=o))):
    for _ in range(10):
        fn()
    ev0.record()
    for _ in range(60):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 60
    print(f"{name:10s} 8192 x 8192 x 4096: {ms * 1e3:.1f} us  {2 * 8192 * 8192 * 4096 / ms / 1e9:.0f} TF/s")
clock(lambda: ops.layernorm(x, g, g, 1e-6), "layernorm (HBM-bound)", 400)
