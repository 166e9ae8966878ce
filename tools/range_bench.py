"""vsc_range_search_ip_f32 on the exact path (two fp32 sweeps) and on the bf16 pre-filter path (VSC_RANGE_PATH):
python tools/range_bench.py [nq nr d frac]   (run on the GPU box)"""
import os
import sys
import time
from statistics import NormalDist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import ops
from vsc_hip import _lib as _vsc_lib

nq, nr, d = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 1_000_000, 512)))
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda", generator=g), dim=1)
r = torch.nn.functional.normalize(torch.randn(nr, d, device="cuda", generator=g), dim=1)
radius = NormalDist().inv_cdf(1.0 - frac) / d ** 0.5
ref = None
for path in ("exact", "bf16"):
    _vsc_lib.set_option("VSC_RANGE_PATH", path)
    out = ops.range_search_ip(q, r, radius, capacity=1 << 24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ops.range_search_ip(q, r, radius, capacity=1 << 24)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    same = True if ref is None else all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                                    b.view(torch.int32) if b.dtype == torch.float32 else b) for a, b in zip(ref, out))
    ref = ref or out
    print(f"[{path}] range_search {nq} x {nr} x {d}, radius {radius:.4f}: {dt * 1e3:.2f} ms, {nq * nr / dt / 1e6:.0f} Mpairs/s, "
          f"{int(out[0][-1])} hits, identical to exact: {same}", flush=True)
