"""A/B of the sweep's work order (run on the GPU box): XCD-aware (8 query blocks x 4 splits per XCD) vs plain, whole call and
the sweep kernel alone.  python tools/micro/knn_map_ab.py [nq ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import _lib, ops

lib = _lib.require_device()
dev = torch.device("cuda:0")
nr, k = 1_000_000, 100
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev)
ops.l2_normalize_(r)
for nq in [int(a) for a in sys.argv[1:]] or [8192, 65536, 262144]:
    q = torch.randn(nq, 512, generator=g, device=dev)
    ops.l2_normalize_(q)
    res = {}
    for mode in ("0", "1", "0", "1"):
        _lib.set_option("VSC_KNN_XCD_MAP", mode)
        ops.knn_ip(q, r, k)
        lib.vsc_knn_set_profiling(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        D, I = ops.knn_ip(q, r, k)
        e1.record()
        torch.cuda.synchronize()
        ph = (ctypes.c_float * 4)()
        _lib.check(lib.vsc_knn_last_profile(ph))
        lib.vsc_knn_set_profiling(0)
        print(f"nq={nq} xcd_map={mode}: call {e0.elapsed_time(e1):.2f} ms; pack {ph[0]:.2f} sweep {ph[1]:.2f} ({2 * nq * nr * 512 / ph[1] / 1e9:.0f} TF/s) "
              f"rescore(+union) {ph[2]:.2f} merge {ph[3]:.2f}; path {lib.vsc_knn_last_path()}", flush=True)
        res[mode] = (D.clone(), I.clone())
    assert torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][0].view(torch.int32), res["1"][0].view(torch.int32)), "orders disagree"
    del q, res
_lib.set_option("VSC_KNN_XCD_MAP", None)
