"""Candidate retrieval (vsc_video_pair_max_f32) at the matching track's scale: n_q query frames against
n_r reference frames, videos of ~`frames` frames.  python tools/pair_max_bench.py [n_q n_r d frac]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from vsc_hip import ops  # noqa: E402


def main():
    n_q, n_r, d = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 1_000_000, 512)))
    frac = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
    frames = 32
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.nn.functional.normalize(torch.randn(n_q, d, device="cuda", generator=g), dim=1)
    r = torch.nn.functional.normalize(torch.randn(n_r, d, device="cuda", generator=g), dim=1)
    qv = (torch.arange(n_q, device="cuda") // frames).int()
    rv = (torch.arange(n_r, device="cuda") // frames).int()
    nqv, nrv = int(qv[-1]) + 1, int(rv[-1]) + 1
    # radius exceeded by ~frac of the pairs: random unit vectors, <q,r> ~ N(0, 1/d)
    from statistics import NormalDist
    thr = NormalDist().inv_cdf(1.0 - frac) / d ** 0.5
    for _ in range(2):
        lims, rvid, sc = ops.video_pair_max(q, qv, nqv, r, rv, nrv, thr, capacity=1 << 24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        lims, rvid, sc = ops.video_pair_max(q, qv, nqv, r, rv, nrv, thr, capacity=1 << 24)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"video_pair_max {n_q} x {n_r} x {d}, {nqv} x {nrv} videos, thr {thr:.4f}: {dt * 1e3:.2f} ms, "
          f"{n_q * n_r / dt / 1e6:.0f} Mpairs/s, {2.0 * n_q * n_r * d / dt / 1e12:.1f} TF/s fp32, "
          f"{int(lims[-1])} video pairs")
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.knn_ip(q, r, 10)
    torch.cuda.synchronize()
    dk = (time.perf_counter() - t0) / reps
    print(f"knn top-10 same shape: {dk * 1e3:.2f} ms ({n_q * n_r / dk / 1e6:.0f} Mpairs/s)")


if __name__ == "__main__":
    main()
