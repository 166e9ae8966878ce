"""Sweep-kernel time and append counts against the compaction trigger (VSC_KNN_TRIG), run on the GPU box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
lib = _lib.require_device()
dev = torch.device("cuda:0")
nq, nr, k = 65536, 1_000_000, 100
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev); ops.l2_normalize_(r)
q = torch.randn(nq, 512, generator=g, device=dev); ops.l2_normalize_(q)
_lib.set_option("VSC_KNN_XCD_MAP", sys.argv[1] if len(sys.argv) > 1 else "0")
for trig in sys.argv[2:] or ["100", "64", "150", "200", "400", "100"]:
    _lib.set_option("VSC_KNN_DELTA", trig)
    _lib.set_option("VSC_KNN_ABL", None)
    ops.knn_ip(q, r, k)
    lib.vsc_knn_set_profiling(1)
    best = 1e9
    for _ in range(3):
        ops.knn_ip(q, r, k)
        ph = (ctypes.c_float * 4)(); _lib.check(lib.vsc_knn_last_profile(ph))
        best = min(best, ph[1])
    lib.vsc_knn_set_profiling(0)
    print(f"delta={trig}: sweep {best:.2f} ms ({2*nq*nr*512/best/1e9:.0f} TF/s) path {lib.vsc_knn_last_path()}", flush=True)
    _lib.set_option("VSC_KNN_ABL", "8")
    ops.knn_ip(q, r, k)
