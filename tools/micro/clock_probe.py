"""Sample rocm-smi clocks / power while a kernel loops (run on the GPU box):
python tools/micro/clock_probe.py gemm|knn|ln|idle   -- prints sclk / power samples taken during the loop."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dev = torch.device("cuda:0")
stop = False
samples = []


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            samples.append(repr(e))
        time.sleep(0.2)


if mode == "gemm":
    a = torch.randn(8192, 4096, device=dev).to(torch.bfloat16)
    w = (torch.randn(8192, 4096, device=dev) * 0.05).to(torch.bfloat16)
    fn = lambda: ops.gemm_bf16(a, w, None)
    flops = 2 * 8192 * 8192 * 4096
elif mode == "gemm0":
    a = torch.zeros(8192, 4096, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(8192, 4096, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm_bf16(a, w, None)
    flops = 2 * 8192 * 8192 * 4096
elif mode == "knn":
    q = torch.randn(16384, 512, device=dev)
    r = torch.randn(200000, 512, device=dev)
    fn = lambda: ops.knn_ip(q, r, 100)
    flops = 2 * 16384 * 200000 * 512
elif mode == "ln":
    x = torch.randn(65404, 768, device=dev)
    g = torch.ones(768, device=dev)
    fn = lambda: ops.layernorm(x, g, g, 1e-6)
    flops = 0
else:
    fn = lambda: time.sleep(0.01)
    flops = 0
for _ in range(3):
    fn()
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n += 20
dt = time.perf_counter() - t0
stop = True
th.join()
print(f"{mode}: {n} launches in {dt:.2f} s, {dt / n * 1e6:.1f} us each" + (f", {flops * n / dt / 1e12:.0f} TFLOP/s" if flops else ""))
hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[0]
print(hdr)
for s in samples[1:8]:
    print(s)
