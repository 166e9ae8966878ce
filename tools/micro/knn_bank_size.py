"""Does the sweep's memory-side traffic cost time?  The sweep kernel's rate (HIP events, vsc_knn_last_profile) for the same 262 144
queries against banks from 32 Ki rows (32 MiB as bf16: inside the 256-MiB Infinity Cache, re-read from it by every query group) to
2 Mi rows (2 GiB: streamed from HBM once per XCD and query group).  If the rate does not fall with the bank's size, the re-reads
the PMC pass counts (profiles/r05_pmc_knn.json) ride on bandwidth the sweep does not need.   (run on the GPU box)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
lib = _lib.require_device()
dev = torch.device("cuda:0")
nq, k, d = 262144, 100, 512
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(nq, d, generator=g, device=dev)
ops.l2_normalize_(q)
for nr in (32768, 131072, 524288, 1048576, 2097152, 4194304):
    r = torch.randn(nr, d, generator=g, device=dev)
    ops.l2_normalize_(r)
    ops.knn_ip(q, r, k)
    torch.cuda.synchronize()
    lib.vsc_knn_set_profiling(1)
    sweeps = []
    for _ in range(3):
        ops.knn_ip(q, r, k)
        torch.cuda.synchronize()
        ph = (ctypes.c_float * 4)()
        _lib.check(lib.vsc_knn_last_profile(ph))
        sweeps.append(float(ph[1]))
    lib.vsc_knn_set_profiling(0)
    ms = sorted(sweeps)[1]
    print(f"bank {nr:8d} rows ({nr * d * 2 / 2**20:6.0f} MiB bf16): sweep {ms:8.2f} ms = {2.0 * nq * nr * d / ms / 1e9:7.1f} TFLOP/s   (path {lib.vsc_knn_last_path()})")
    del r
