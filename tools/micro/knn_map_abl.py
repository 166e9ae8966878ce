import ctypes, os, sys
sys.path.insert(0, "/root/repo/vsc22-submission_amd")
import torch
from vsc_hip import _lib, ops
lib = _lib.require_device()
dev = torch.device("cuda:0")
nr, k = 1_000_000, 100
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev); ops.l2_normalize_(r)
for nq in (8192, 65536):
    q = torch.randn(nq, 512, generator=g, device=dev); ops.l2_normalize_(q)
    for abl in ("1", "8"):
        for mode in ("0", "1", "0", "1"):
            _lib.set_option("VSC_KNN_XCD_MAP", mode); _lib.set_option("VSC_KNN_ABL", abl)
            ops.knn_ip(q, r, k)
            lib.vsc_knn_set_profiling(1)
            ops.knn_ip(q, r, k)
            ph = (ctypes.c_float * 4)(); _lib.check(lib.vsc_knn_last_profile(ph)); lib.vsc_knn_set_profiling(0)
            print(f"nq={nq} abl={abl} xcd_map={mode}: sweep {ph[1]:.2f} ms ({2*nq*nr*512/ph[1]/1e9:.0f} TF/s)", flush=True)
