"""The pipelined sharded search's per-shard sweeps on one GPU: 262 144 queries against a 1M-row bank as eight 125k-row shards
(configs[3] at bench size: what a rank sweeps while the other ranks' shards arrive), with the running k-th score carried as the next
shard's floor (vsc_knn_ip_floor_f32) and without, next to one sweep over the whole bank.   (run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
from vsc_hip.distributed import sweep_shards
dev = torch.device("cuda:0")
nq, nr, k, parts = 262144, 1_000_000, 100, 8
g = torch.Generator(device=dev).manual_seed(1)
r = torch.randn(nr, 512, generator=g, device=dev); ops.l2_normalize_(r)
q = torch.randn(nq, 512, generator=g, device=dev); ops.l2_normalize_(q)
cuts = [nr * j // parts for j in range(parts + 1)]


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


t1, (D, I) = timed(lambda: ops.knn_ip(q, r, k))
print(f"one sweep over the bank                 : {t1:8.2f} ms")
for carry in (False, True):
    t, (Ds, Is) = timed(lambda: sweep_shards(q, ((r[cuts[j]:cuts[j + 1]], cuts[j]) for j in range(parts)), k, ops.knn_ip, None, carry))
    same = bool(torch.equal(Is, I) and torch.equal(Ds.view(torch.int32), D.view(torch.int32)))
    print(f"eight shards, floor carried = {str(carry):5s}      : {t:8.2f} ms   ({t / t1:.2f} x)   same bits as one sweep: {same}")
