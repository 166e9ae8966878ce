"""Where the similarity sweep's time goes (run on the GPU box): the bf16 pre-filter sweep kernel alone (HIP events of
vsc_knn_set_profiling) with the VSC_KNN_ABL diagnostic switches -- 0 full, 8 full + counters, 1 no filter at all, 2 masks
computed but nothing appended, 4 appends counted but no key stores.  python tools/micro/knn_abl.py [nq] [nr] [k]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = _lib.require_device()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev)
q = torch.randn(nq, 512, generator=g, device=dev)
ops.l2_normalize_(r)
ops.l2_normalize_(q)
ops.knn_ip(q, r, k)
lib.vsc_knn_set_profiling(1)
for abl in sys.argv[4:] or ["0", "8", "1", "2", "4", "0"]:
    _lib.set_option("VSC_KNN_ABL", abl if abl != "0" else None)
    best = None
    for _ in range(3):
        ops.knn_ip(q, r, k)
        ph = (ctypes.c_float * 4)()
        _lib.check(lib.vsc_knn_last_profile(ph))
        best = list(ph) if best is None or ph[1] < best[1] else best
    print(f"abl={abl}: pack {best[0]:.2f} sweep {best[1]:.2f} rescore {best[2]:.2f} merge {best[3]:.2f} ms; sweep "
          f"{2 * nq * nr * 512 / best[1] / 1e9:.0f} TF/s", flush=True)
_lib.set_option("VSC_KNN_ABL", None)
