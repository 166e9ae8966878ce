"""Throughput AND shader clock of each GEMM tile configuration (VSC_GEMM_CFG, read once per process: one subprocess
per configuration) and of the library's kernel on the same operands, measured separately (events over 100
back-to-back launches; then a cycle-counted spin beside 60 more).   python tools/micro/cfg_clock.py"""
import os
import subprocess
import sys

if len(sys.argv) == 1:
    for cfg in ("A", "C", "B", "lib"):
        env = dict(os.environ)
        if cfg != "lib":
            env["VSC_GEMM_CFG"] = cfg
        subprocess.run([sys.executable, os.path.abspath(__file__), cfg], env=env, check=True)
    sys.exit(0)
CFG = sys.argv[1]

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops

lib = _lib.require_device()
dev = torch.device("cuda:0")
ticks = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()


def measure(fn, flop):
    for _ in range(30):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    for _ in range(30):
        fn()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(60):
        fn()
    with torch.cuda.stream(side):
        s0.record()
        _lib.check(lib.vsc_debug_spin_ticks(int(us * 25 * 1500), ticks.data_ptr(), side.cuda_stream))
        s1.record()
    torch.cuda.synchronize()
    ghz = int(ticks.item()) / (s0.elapsed_time(s1) * 1e6)
    tf = flop / us / 1e6
    return us, tf, ghz, tf / (2500 * ghz / 2.4)


for m, n, k in ((8192, 8192, 4096), (65404, 768, 3072), (65404, 2304, 768), (65404, 3072, 768)):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    fn = (lambda: torch.matmul(a, w.t(), out=o)) if CFG == "lib" else (lambda: ops.gemm_bf16(a, w, None, out=o))
    for rnd in range(2):
        us, tf, ghz, duty = measure(fn, 2.0 * m * n * k)
        print(f"{CFG:4s} M={m} N={n} K={k}: {us:8.1f} us  {tf:7.1f} TF/s  {ghz:.3f} GHz  MFMA duty at that clock {duty:.2f}", flush=True)
    del a, w, o
