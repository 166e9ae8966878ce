"""Multi-GPU readiness on ONE GPU: the sharded search of BASELINE.json configs[3] (8M x 512 reference bank assembled by an
RCCL all_gather, local sweep, result gather) under a real one-rank RCCL communicator at the full bank size.  No 1 -> 8
scaling curve exists anywhere in this repository: 8-GPU runs are the driver's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [ROOT, os.path.join(ROOT, "vsc22-submission_amd")]
from oracle import knn_oracle
from vsc_hip import _lib, ops, distributed as vdist
_lib.require_device()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nr, nq, d, k = 8_000_000, 1024, 512, 100
g = torch.Generator(device=dev).manual_seed(3)
refs = torch.empty(nr, d, device=dev)
for c0 in range(0, nr, 1_000_000):                      # generated in 2-GB pieces: randn has no bank-sized temporaries then
    refs[c0:c0 + 1_000_000] = torch.randn(1_000_000, d, generator=g, device=dev)
ops.l2_normalize_(refs)
q = torch.randn(nq, d, generator=g, device=dev)
ops.l2_normalize_(q)
q[:4] = refs[torch.tensor([0, 2_096_895, 2_096_896, nr - 1], device=dev)]   # rows either side of the first split boundary
D, I = vdist.sharded_knn(q, refs, k, always_collective=True)               # all_gather_into_tensor of 16 GB on RCCL, sweep, gather
assert _lib.require_device().vsc_knn_last_path() == 2
D, I = D.cpu().numpy(), I.cpu().numpy()
assert I.shape == (nq, k) and (I[:4, 0] == [0, 2_096_895, 2_096_896, nr - 1]).all()
rows = np.r_[0:4, 500:502, nq - 2:nq]
qh = q[torch.from_numpy(rows).to(dev)].cpu().numpy()
# oracle over the bank in 1M-row pieces (2 GB of host memory at a time), merged on (score desc, id asc): pairs are independent
best = None
for c0 in range(0, nr, 1_000_000):
    Dc, Ic = knn_oracle.knn_ip(qh, refs[c0:c0 + 1_000_000].cpu().numpy(), k)
    Ic = Ic + c0
    if best is None:
        best = (Dc, Ic)
    else:
        Dm, Im = np.concatenate([best[0], Dc], 1), np.concatenate([best[1], Ic], 1)
        order = np.lexsort((Im, -Dm.astype(np.float64)), axis=1)[:, :k]
        best = (np.take_along_axis(Dm, order, 1), np.take_along_axis(Im, order, 1))
assert np.array_equal(I[rows], best[1]), "ids differ from the oracle"
assert np.array_equal(D[rows].view(np.uint32), best[0].view(np.uint32)), "scores are not bit-identical"
assert (D[:, :-1] >= D[:, 1:]).all() and I.min() >= 0 and I.max() < nr
freed = _lib.require_device().vsc_search_release_scratch()
assert freed > 8_000_000 * 512 * 2
dist.destroy_process_group()
print("sharded 8M-row bank ok; scratch released", freed)
"""


def test_sharded_search_full_bank_one_rank_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + SCRIPT], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "sharded 8M-row bank ok" in out.stdout


def test_knn_merge_parts_equals_one_search_over_the_whole_bank():
    """vsc_knn_merge_parts_f32: a 300k-row bank swept as five ragged shards (one smaller than k, duplicated rows across shards so that
    tied scores meet in the merge) with their id offsets and merged == one vsc_knn_ip_f32 call over the whole bank, bit for bit."""
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vsc22-submission_amd")]
    from vsc_hip import _lib, ops
    _lib.require_device()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    nr, nq, d, k = 300_000, 777, 128, 40
    refs = torch.randn(nr, d, generator=g, device=dev)
    q = torch.randn(nq, d, generator=g, device=dev)
    refs[150_000] = refs[7]
    refs[299_999] = refs[7]
    q[0] = refs[7]
    D, I = ops.knn_ip(q, refs, k)
    cuts = [0, 100_000, 100_020, 180_000, 299_990, nr]          # shards of 100 000, 20 (< k), 79 980, 119 990, 10 rows
    parts = [ops.knn_ip(q, refs[a:b].contiguous(), k, ref_id_offset=a) for a, b in zip(cuts[:-1], cuts[1:])]
    Dm, Im = ops.knn_merge_parts(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(Im, I) and torch.equal(Dm.view(torch.int32), D.view(torch.int32))
    assert I[0, :3].tolist() == [7, 150_000, 299_999]


PIPE_SCRIPT = r"""
import os, sys
import torch, torch.distributed as dist
sys.path[:0] = [ROOT, os.path.join(ROOT, "vsc22-submission_amd")]
from vsc_hip import _lib, ops, distributed as vdist
_lib.require_device()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(5)
refs = torch.randn(500_000, 512, generator=g, device=dev); ops.l2_normalize_(refs)
q = torch.randn(4096, 512, generator=g, device=dev); ops.l2_normalize_(q)
noise = torch.randn(20_000, 512, generator=g, device=dev); ops.l2_normalize_(noise)
a = vdist.sharded_knn(q, refs, 100, always_collective=True, gather_to=None)
b = vdist.sharded_knn(q, refs, 100, always_collective=True, gather_to=None, pipelined=True)      # RCCL broadcast (async) + wait + sweep
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
Dn, In = vdist.sharded_knn_score_normalized(q, refs, noise, 100, beta=1.2, nk=3, always_collective=True, gather_to=None, pipelined=True)
bias = vdist.score_norm_bias(q, noise, 1.2, 3)
D1, I1 = ops.knn_ip(torch.cat([q, bias], 1).contiguous(), torch.cat([refs, torch.ones_like(refs[:, :1])], 1).contiguous(), 100)
assert torch.equal(In, I1) and torch.equal(Dn, D1)
# the bias is a per-query constant: the ranking of a query's references is the un-normalised one, the scores are shifted by it
assert torch.equal(In, a[1]) or (In != a[1]).float().mean() < 1e-3
assert torch.allclose(Dn, a[0] + bias, atol=2e-6)
dist.destroy_process_group()
print("pipelined + score-normalised sharded search ok")
"""


def test_pipelined_and_score_normalised_sharded_search_one_rank_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29542", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + PIPE_SCRIPT], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "score-normalised sharded search ok" in out.stdout
