"""vsc_pair_similarity_f32 over pair counts and matrix sizes (banks of 200k x 512 rows each): GFLOP/s of the exact fp32 chains"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import numpy as np, torch
from vsc_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
nb = 200_000
q = torch.randn(nb, 512, generator=g, device=dev); r = torch.randn(nb, 512, generator=g, device=dev)
rng = np.random.default_rng(0)
for n, (a, b) in [(100, (40, 60)), (2000, (40, 60)), (20000, (40, 60)), (20000, (7, 9)), (2000, (200, 300)), (200, (1000, 1500)), (20000, (64, 64))]:
    q0 = rng.integers(0, nb - a, n); r0 = rng.integers(0, nb - b, n)
    pairs = np.stack([q0, np.full(n, a), r0, np.full(n, b)], 1).astype(np.int64)
    ops.pair_similarity(q, r, pairs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ops.pair_similarity(q, r, pairs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    fl = 2.0 * n * a * b * 512
    print(f"{n:6d} pairs of {a:4d} x {b:4d}: {ms:8.2f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   ({n * a * b * 4 / 1e6:.0f} MB out)", flush=True)
