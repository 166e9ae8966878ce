"""Similarity search parity: vsc_knn_ip_f32 must be BIT-EXACT with oracle/knn_oracle.c
(scores as uint32 patterns, ids as int64), including ties, ragged shapes and k > nr."""
import os

import numpy as np
import pytest
import torch

from tools import synth
from vsc_hip import _lib as _vsc_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


def _check(dev, q, r, k):
    from oracle import knn_oracle
    from vsc_hip import ops
    D, I = ops.knn_ip(torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = knn_oracle.knn_ip(q, r, k)
    assert np.array_equal(I, Ir), f"ids differ at {np.argwhere(I != Ir)[:5]}"
    assert np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), "scores are not bit-identical"
    return D, I


@pytest.mark.parametrize("nq,nr,d,k", [
    (3, 5, 3, 2),            # the reference's own unit-test scale (tests/test_index.py)
    (64, 1000, 512, 10),     # BASELINE configs[0]: 64 x 1k cosine top-10
    (130, 1001, 511, 7),     # ragged everything; 511 = dimension after replace_dim
    (1, 70000, 512, 1),      # one query, many ref splits (score normalisation, nk = 1)
    (257, 20000, 512, 100),  # top-100, several compactions
    (5, 40, 16, 64),         # k > nr: padded with (-FLT_MAX, -1)
    (300, 3000, 64, 500),    # k in the 1024-capacity lists
    (9, 5000, 128, 1024),    # largest k (exhaustive_search.py:66)
])
def test_knn_bit_exact(dev, nq, nr, d, k):
    q = synth.descriptor_bank(100 + nq, nq, d)
    r = synth.descriptor_bank(200 + nr, nr, d)
    D, I = _check(dev, q, r, k)
    if k > nr:
        assert (I[:, nr:] == -1).all() and (D[:, nr:] == np.finfo(np.float32).min).all()


@pytest.fixture
def force_prefilter():
    """VSC_KNN_PATH=bf16 for one test (vsc_set_option: the library reads its environment only once per process)."""
    _vsc_lib.set_option("VSC_KNN_PATH", "bf16")
    yield
    _vsc_lib.set_option("VSC_KNN_PATH", None)


def _last_path():
    from vsc_hip import _lib
    return _lib.require_device().vsc_knn_last_path()


@pytest.mark.parametrize("nq,nr,d,k", [
    (3, 5, 3, 2), (64, 1000, 512, 10), (130, 1001, 511, 7), (1, 70000, 512, 1), (257, 20000, 512, 100),
    (5, 40, 16, 64), (300, 3000, 64, 500), (700, 9000, 100, 257), (513, 4097, 512, 33),
    # widths whose packed rows are an even, non-power-of-two number of K-tiles (513 -> 640: the score-normalised search; 384; 577 -> 640)
    # with the re-scoring's tail columns behind its 32-float chunks, and k on the large list form (129 .. 384)
    (300, 20000, 513, 10), (257, 30000, 384, 20), (130, 20000, 577, 5), (200, 40000, 512, 256), (64, 9000, 200, 130),
])
def test_knn_prefilter_path_bit_exact(dev, force_prefilter, nq, nr, d, k):
    """The bf16 pre-filter sweep + exact re-scoring (VSC_KNN_PATH=bf16 forces it at every size) returns the same bits
    as the oracle's fp32 chain: scores as uint32 patterns, ids, tie order, ragged tiles, k > nr padding."""
    q = synth.descriptor_bank(100 + nq, nq, d)
    r = synth.descriptor_bank(200 + nr, nr, d)
    D, I = _check(dev, q, r, k)
    assert _last_path() == 2, "the pre-filter path did not run (or fell back)"
    if k > nr:
        assert (I[:, nr:] == -1).all() and (D[:, nr:] == np.finfo(np.float32).min).all()


def test_knn_prefilter_unnormalised_ties_and_fallback(dev, force_prefilter):
    """Pre-filter path on inputs that stress its error bound and its lists: rows of very different norms, exact
    duplicates across the bank (ties -> lower id first), and a bank with thousands of near-duplicates of a query, whose
    candidate band cannot fit: the device flag must send the call to the exact sweep (path 3), result still exact."""
    rng = np.random.RandomState(5)
    r = synth.descriptor_bank(31, 6000, 96) * rng.uniform(0.01, 30.0, size=(6000, 1)).astype(np.float32)
    q = synth.descriptor_bank(32, 70, 96) * rng.uniform(0.1, 5.0, size=(70, 1)).astype(np.float32)
    r[10] *= 40.0 / np.linalg.norm(r[10])    # the longest row: <r10, r10> beats every other pair of query 0
    r[4000:4100] = r[10]
    r[5999] = r[10]
    q[0] = r[10]
    D, I = _check(dev, q, r, 50)
    assert _last_path() == 2
    assert I[0, 0] == 10 and (I[0, 1:50] == np.arange(4000, 4049)).all()
    # 600 queries (3 blocks) x 524288 refs = 8 tiles per reference split; 4000 consecutive refs glued to query 300 put
    # ~2000 candidates into two bands of block 1 (KEEP = 512): that block alone is redone on the exact sweep
    r2 = synth.descriptor_bank(33, 524288, 32)
    q2 = synth.descriptor_bank(34, 600, 32)
    r2[100000:104000] = q2[300] + 1e-4 * synth.normalish(35, (4000, 32))
    from oracle import knn_oracle
    from vsc_hip import ops
    D, I = ops.knn_ip(torch.from_numpy(q2).to(dev), torch.from_numpy(r2).to(dev), 20)
    assert _last_path() == 3
    sub = np.r_[0:8, 296:304, 592:600]
    Dr, Ir = knn_oracle.knn_ip(q2[sub], r2, 20)
    assert np.array_equal(I.cpu().numpy()[sub], Ir) and np.array_equal(D.cpu().numpy()[sub].view(np.uint32), Dr.view(np.uint32))
    q2[5, 7] = np.inf                                                  # no finite bound: exact sweep for that block
    ops.knn_ip(torch.from_numpy(q2[:64]).to(dev), torch.from_numpy(r2[:8192]).to(dev), 5)
    assert _last_path() == 3


def test_knn_auto_path_and_full_size_properties(dev):
    """BASELINE configs[2] at its full reference-bank size (1M x 512-d, top-100) with 4096 queries, on the path the
    default call takes there (bf16 pre-filter + exact re-scoring): a 64-query subset against the oracle bit for bit,
    self-match, and permutation invariance of every score."""
    from oracle import knn_oracle
    from vsc_hip import ops
    nr, nq, k = 1_000_000, 4096, 100
    r = synth.descriptor_bank(91, nr, 512)
    q = synth.descriptor_bank(92, nq, 512)
    q[:64] = r[np.arange(0, 64) * 15001]
    rt, qt = torch.from_numpy(r).to(dev), torch.from_numpy(q).to(dev)
    D, I = ops.knn_ip(qt, rt, k)
    assert _last_path() == 2
    D, I = D.cpu().numpy(), I.cpu().numpy()
    assert (I[:64, 0] == np.arange(0, 64) * 15001).all()
    sub = np.r_[0:32, nq - 32:nq]
    Dr, Ir = knn_oracle.knn_ip(q[sub], r, k)
    assert np.array_equal(I[sub], Ir) and np.array_equal(D[sub].view(np.uint32), Dr.view(np.uint32))
    perm = np.random.RandomState(1).permutation(nr)
    D2, I2 = ops.knn_ip(qt, rt[torch.from_numpy(perm).to(dev)], k)
    assert np.array_equal(D2.cpu().numpy().view(np.uint32), D.view(np.uint32))
    # same references behind the same scores; two references of one query may tie bit for bit (a few do, among 4096 x
    # 100 fp32 scores), and a tie is ordered by the id inside the bank that was searched: compare per score group
    back = perm[I2.cpu().numpy()]
    same = back == I
    assert same.mean() > 0.999
    for row, col in np.argwhere(~same):
        tie = D[row] == D[row, col]
        if tie.sum() == 1:      # the tie partner is the (k+1)-th pair: only the last slot can differ, at an equal score
            assert col == k - 1
        else:
            assert sorted(back[row, tie]) == sorted(I[row, tie]) or col == k - 1


def _device_bank(dev, seed, n, d=512):
    """L2-normalised random bank made on the GPU (a 1M x 512 bank takes a minute through tools/synth.py on the host); the
    oracle gets the same bits back through a device -> host copy."""
    from vsc_hip import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, generator=g, device=dev)
    ops.l2_normalize_(x)
    return x


def test_knn_in_the_benchmarked_regime_single_split_many_items_per_workgroup(dev):
    """The regime bench.py's 1M x 1M search runs in: nq >= 65536 -> one reference split, and each of the 256 persistent
    workgroups walks several (query block) work items back to back (lists, thresholds and LDS counters re-initialised per
    item).  262 221 queries = 1025 query blocks (the last one 77 rows): 64 queries spread over the first, middle and last
    work items of several workgroups against knn_oracle, bit for bit.  Replaces faiss IndexFlat.search,
    infer/vsc/index.py:167-175."""
    from oracle import knn_oracle
    from vsc_hip import ops
    nr, nq, k = 1_000_000, 262_221, 100
    rt = _device_bank(dev, 5, nr)
    qt = _device_bank(dev, 6, nq)
    D, I = ops.knn_ip(qt, rt, k)
    assert _last_path() == 2
    # workgroup b owns query blocks b, b + 256, ...: block 0 / 512 / 1024 = first / third / last item of workgroup 0
    blocks = [0, 255, 256, 511, 512, 700, 1023, 1024]
    rows = np.concatenate([np.minimum(b * 256 + np.array([0, 1, 37, 76, 100, 128, 200, 255]), nq - 1) for b in blocks])
    rows = np.unique(np.r_[rows, nq - 1])
    r = rt.cpu().numpy()
    Dr, Ir = knn_oracle.knn_ip(qt[torch.from_numpy(rows).to(dev)].cpu().numpy(), r, k)
    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    assert np.array_equal(Ih[rows], Ir), f"ids differ at rows {rows[np.argwhere(Ih[rows] != Ir)[:5, 0]]}"
    assert np.array_equal(Dh[rows].view(np.uint32), Dr.view(np.uint32))
    # whole result: sorted, ids in range, no duplicate id within a row
    assert (Dh[:, :-1] >= Dh[:, 1:]).all() and Ih.min() >= 0 and Ih.max() < nr
    srt = np.sort(Ih[::997], axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()


def test_knn_at_the_full_size_of_configs2(dev):
    """BASELINE.json configs[2] itself: 1 000 000 queries x 1 000 000 references x 512, k = 100 -- the call bench.py times.  Between
    262 221 queries (the test above) and this size the candidate lists grow past 4 GiB (1M lists x 1024 keys x 8 B) and the call splits
    off its last partial round (3 907 query blocks = 15 rounds of 256 + 67: tail balancing): 72 queries from the first / middle / last
    blocks of the main sweep, the 4-GiB crossing of the list array, and the balanced tail, against knn_oracle, bit for bit; whole-result
    invariants over every row."""
    from oracle import knn_oracle
    from vsc_hip import ops
    nr = nq = 1_000_000
    k = 100
    rt = _device_bank(dev, 5, nr)
    qt = _device_bank(dev, 6, nq)
    D, I = ops.knn_ip(qt, rt, k)
    assert _last_path() == 2
    nqb = (nq + 255) // 256                      # 3907 query blocks; the tail call owns blocks 3840 .. 3906
    cross = (4 << 30) // (1024 * 8 * 256)        # first query block whose candidate list lies past byte 2^32 of the list array
    blocks = [0, 1, 255, 256, cross - 1, cross, cross + 1, nqb // 2, 3839, 3840, 3841, nqb - 2, nqb - 1]
    rows = np.unique(np.concatenate([np.minimum(b * 256 + np.array([0, 37, 128, 200, 255]), nq - 1) for b in blocks] + [[nq - 1]]))
    r = rt.cpu().numpy()
    Dr, Ir = knn_oracle.knn_ip(qt[torch.from_numpy(rows).to(dev)].cpu().numpy(), r, k)
    Dh, Ih = D[torch.from_numpy(rows).to(dev)].cpu().numpy(), I[torch.from_numpy(rows).to(dev)].cpu().numpy()
    assert np.array_equal(Ih, Ir), f"ids differ at rows {rows[np.argwhere(Ih != Ir)[:5, 0]]}"
    assert np.array_equal(Dh.view(np.uint32), Dr.view(np.uint32))
    # whole result (on the device: 800 MB of ids): sorted scores, ids in range, no duplicate id within a row
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < nr
    srt = torch.sort(I[::499], dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    del D, I
    torch.cuda.empty_cache()
    from vsc_hip import _lib
    _lib.require_device().vsc_search_release_scratch()


def test_knn_tail_balancing_is_invisible(dev):
    """The query blocks of a call's last partial round are swept as a second, finer-grained sweep (vsc_knn_ip_f32: tail balancing;
    140 000 queries = 547 blocks = two whole rounds of 256 + 35 blocks, which take 7 reference splits each): the same bits as the
    one-sweep form (VSC_KNN_TAIL=0), and the profile adds the two sweeps up."""
    import ctypes
    from vsc_hip import _lib, ops
    lib = _lib.require_device()
    g = torch.Generator(device=dev).manual_seed(12)
    r = torch.randn(300_000, 512, generator=g, device=dev)
    q = torch.randn(140_000, 512, generator=g, device=dev)
    ops.l2_normalize_(r)
    ops.l2_normalize_(q)
    q[139_999] = r[77]                                   # a planted neighbour in the tail part
    lib.vsc_knn_set_profiling(1)
    D, I = ops.knn_ip(q, r, 50)
    ms = (ctypes.c_float * 4)()
    _lib.check(lib.vsc_knn_last_profile(ms))
    lib.vsc_knn_set_profiling(0)
    assert lib.vsc_knn_last_path() == 2 and int(I[139_999, 0]) == 77 and ms[1] > 0
    _lib.set_option("VSC_KNN_TAIL", "0")
    try:
        D0, I0 = ops.knn_ip(q, r, 50)
    finally:
        _lib.set_option("VSC_KNN_TAIL", None)
    assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))


@pytest.mark.parametrize("nq", [12_000, 40_000, 70_000])
def test_knn_split_plans_chosen_by_the_cost_model(dev, nq):
    """sweep_plan picks the number of reference splits by a cost model (rounds of 256 work items x the per-list warm-up): 12 000 queries
    = 47 blocks (XCD-aware order with 4 splits or the plain order with 5, no longer 16), 40 000 = 157 blocks x 3 splits (two
    third-size rounds, no longer 1.23 rounds of halves), 70 000 = 274 blocks = one whole round + an 18-block tail swept as a call of its
    own in 14 splits (no longer two rounds).  Rows from every part of every plan against the oracle, bit for bit, and the tail
    switch gives the same bits."""
    from oracle import knn_oracle
    from vsc_hip import _lib, ops
    nr, k = 300_000, 100
    rt = _device_bank(dev, 21, nr)
    qt = _device_bank(dev, 22, nq)
    qt[nq - 1] = rt[nr - 1]
    qt[0] = rt[123]
    D, I = ops.knn_ip(qt, rt, k)
    assert _last_path() == 2
    rows = np.unique(np.r_[0:6, 255:258, 2047:2050, nq // 2:nq // 2 + 3, 65535:65538, nq - 6:nq])
    rows = rows[rows < nq]
    Dr, Ir = knn_oracle.knn_ip(qt[torch.from_numpy(rows).to(dev)].cpu().numpy(), rt.cpu().numpy(), k)
    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    assert np.array_equal(Ih[rows], Ir) and np.array_equal(Dh[rows].view(np.uint32), Dr.view(np.uint32))
    assert Ih[0, 0] == 123 and Ih[nq - 1, 0] == nr - 1
    assert (Dh[:, :-1] >= Dh[:, 1:]).all() and Ih.min() >= 0 and Ih.max() < nr
    _lib.set_option("VSC_KNN_TAIL", "0")
    _lib.set_option("VSC_KNN_XCD_MAP", "0")
    try:
        D0, I0 = ops.knn_ip(qt, rt, k)
    finally:
        _lib.set_option("VSC_KNN_TAIL", None)
        _lib.set_option("VSC_KNN_XCD_MAP", None)
    assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))


@pytest.mark.parametrize("path", ["bf16", "exact"])
def test_knn_with_a_floor_bit_exact(dev, path):
    """vsc_knn_ip_floor_f32: the k best of {r : <q, r> >= floor[q]} -- on the pre-filter sweep the floor is where a list's threshold
    starts, on the exact sweep it only cuts the tail off -- against the oracle's list cut at the floor, bit for bit; floors at an
    exact score (ties stay in), above every score (empty list), at -FLT_MAX / -inf (no floor)."""
    from oracle import knn_oracle
    from vsc_hip import _lib, ops
    nq, nr, k = 700, 70_000, 50
    rt = _device_bank(dev, 31, nr)
    qt = _device_bank(dev, 32, nq)
    rt[60_000] = rt[5]                                        # a duplicate row: equal scores, two ids
    Dr, Ir = knn_oracle.knn_ip(qt.cpu().numpy(), rt.cpu().numpy(), k)
    floor = Dr[:, k // 2].copy()                              # exactly the 26th score: 26 entries stay (ties included)
    floor[0:8] = Dr[0:8, 0] + 1.0                             # nothing reaches it
    floor[8:16] = np.finfo(np.float32).min
    floor[16:24] = -np.inf
    floor[24:32] = Dr[24:32, k - 1] - 0.25                    # below the whole list
    _lib.set_option("VSC_KNN_PATH", path)
    try:
        D, I = ops.knn_ip(qt, rt, k, floor=torch.from_numpy(floor).to(dev))
        assert _last_path() == (2 if path == "bf16" else 1)
    finally:
        _lib.set_option("VSC_KNN_PATH", None)
    cut = Dr < floor[:, None]
    De, Ie = Dr.copy(), Ir.copy()
    De[cut], Ie[cut] = np.finfo(np.float32).min, -1
    assert np.array_equal(I.cpu().numpy(), Ie) and np.array_equal(D.cpu().numpy().view(np.uint32), De.view(np.uint32))
    assert (Ie[0:8] == -1).all() and (Ie[8:32] >= 0).all() and (Ie[40, : k // 2 + 1] >= 0).all()


def test_knn_shard_by_shard_with_carried_floor_equals_one_sweep(dev):
    """vsc_hip.distributed.sweep_shards (the pipelined sharded search's loop) over eight unequal local shards -- one of them shorter
    than k, one empty, duplicates of one row in three shards -- with the running k-th score carried as the next shard's floor, and
    without: both equal one sweep over the concatenated bank, bit for bit."""
    from vsc_hip import ops
    from vsc_hip.distributed import sweep_shards
    nq, k = 3000, 100
    bank = _device_bank(dev, 41, 400_000)
    qt = _device_bank(dev, 42, nq)
    bank[250_007] = bank[11]
    bank[399_999] = bank[11]
    qt[5] = bank[11]
    cuts = [0, 90_000, 90_040, 90_040, 180_000, 250_000, 250_008, 330_000, 400_000]
    D, I = ops.knn_ip(qt, bank, k)
    for carry in (True, False):
        shards = ((bank[cuts[j]:cuts[j + 1]], cuts[j]) for j in range(8))
        Ds, Is = sweep_shards(qt, shards, k, ops.knn_ip, None, carry)
        assert torch.equal(Is, I) and torch.equal(Ds.view(torch.int32), D.view(torch.int32)), carry
    assert I[5, :3].tolist() == [11, 250_007, 399_999]


def test_knn_bank_beyond_one_buffer_descriptor(dev):
    """A split is addressed through one buffer descriptor (32-bit extent): 8191 tiles = 2 096 896 rows of 512-d bf16.  With
    nq >= 65 281 (256 query blocks -> one split wanted) and 2.2 M references the cap forces two splits (knn.hip,
    max_tiles) -- the regime of BASELINE configs[3]'s 8M-row bank.  32 queries against the oracle bit for bit, among them
    queries whose best matches sit on both sides of the split boundary."""
    from oracle import knn_oracle
    from vsc_hip import ops
    nr, nq, k = 2_200_000, 65_600, 100
    rt = _device_bank(dev, 7, nr)
    qt = _device_bank(dev, 8, nq)
    # plant near-duplicates of 4 queries just before and just after row 2 096 896 (the split boundary)
    edge = 8191 * 256
    planted = torch.tensor([edge - 3, edge - 1, edge, edge + 2, nr - 1], device=dev)
    qt[:4] = rt[planted[:4]]
    qt[nq - 1] = rt[nr - 1]
    D, I = ops.knn_ip(qt, rt, k)
    assert _last_path() == 2
    rows = np.unique(np.r_[0:8, 255:259, 32768:32772, 65279:65283, 65535:65539, nq - 8:nq])
    r = rt.cpu().numpy()
    Dr, Ir = knn_oracle.knn_ip(qt[torch.from_numpy(rows).to(dev)].cpu().numpy(), r, k)
    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    assert np.array_equal(Ih[rows], Ir) and np.array_equal(Dh[rows].view(np.uint32), Dr.view(np.uint32))
    assert (Ih[:4, 0] == planted[:4].cpu().numpy()).all() and Ih[nq - 1, 0] == nr - 1
    assert (Dh[:, :-1] >= Dh[:, 1:]).all() and Ih.min() >= 0 and Ih.max() < nr


@pytest.mark.parametrize("scaled", [False, True])
def test_prefilter_error_bound_is_measured_not_assumed(dev, scaled):
    """The exactness of the pre-filter rests on |s~ - s| <= eps_q for every pair (knn.hip, header of the sweep), which in
    turn assumes that one v_mfma_f32_16x16x32_bf16 accumulation stays within d 2^-23 |q||r| of the real sum.  Measure it:
    s~ = the bf16 MFMA scores of the same main loop (mainloop64.h: vsc_gemm_bf16 with the plain fp32 write-out on the RNE
    bf16 copies the pack kernel makes), s = the exact fp32 chain (vsc_pair_similarity_f32), over all 4096 x 1M pairs;
    eps_q as the kernel computes it (eps2 / 2).  The worst ratio is reported and must stay below 1."""
    from vsc_hip import _lib, ops
    nr, nq, d = 1_000_000, 4096, 512
    r = _device_bank(dev, 11, nr)
    q = _device_bank(dev, 12, nq)
    if scaled:   # un-normalised rows, norms over three decades
        g = torch.Generator(device=dev).manual_seed(13)
        r *= torch.exp(torch.rand(nr, 1, generator=g, device=dev) * 6.9 - 3.45)
        q *= torch.exp(torch.rand(nq, 1, generator=g, device=dev) * 6.9 - 3.45)
    qb, rb = q.to(torch.bfloat16), r.to(torch.bfloat16)            # RNE, like pack_bf16x2
    dq, dr = q - qb.float(), r - rb.float()
    n64 = lambda t: t.double().norm(dim=1)
    rmax, drmax = n64(r).max(), n64(dr).max()
    eps = (1.02 * (n64(dq) * rmax + n64(qb.float()) * drmax) + d * (2.0 ** -22 + 2.0 ** -24) * n64(q) * rmax).float()
    worst, worst_abs = 0.0, 0.0
    chunk = 65536
    for c0 in range(0, nr, chunk):
        rows = min(chunk, nr - c0)
        approx = ops.gemm_bf16(qb, rb[c0:c0 + rows], epilogue=_lib.EPI_F32)
        exact, _ = ops.pair_similarity(q, r, [[0, nq, c0, rows]])
        err = (approx - exact.view(nq, rows)).abs()
        worst = max(worst, float((err / eps[:, None]).max()))
        worst_abs = max(worst_abs, float(err.max()))
        del approx, exact, err
    print(f"pre-filter bound: max |s~ - s| / eps_q = {worst:.4f} (max |s~ - s| = {worst_abs:.3e}) over {nq} x {nr} pairs, scaled={scaled}")
    assert 0.0 < worst < 1.0
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/prefilter_bound_{'scaled' if scaled else 'unit'}.txt", "w") as f:
        f.write(f"max |s~ - s| / eps_q = {worst:.6f}; max |s~ - s| = {worst_abs:.6e}; nq={nq} nr={nr} d={d} scaled={scaled}\n")


def test_knn_ties_rank_lower_index_first(dev):
    r = synth.descriptor_bank(7, 3000, 64)
    r[1500:1600] = r[10]          # 100 exact duplicates of row 10, far away in the bank
    r[2999] = r[10]
    q = np.concatenate([r[10:11], synth.descriptor_bank(8, 4, 64)])
    D, I = _check(dev, q, r, 50)
    assert I[0, 0] == 10 and (I[0, 1:50] == np.arange(1500, 1549)).all()


def test_knn_zero_vectors_and_unnormalised(dev):
    """test_candidates.py's reference vectors: zero rows, un-normalised rows."""
    q = np.eye(3, dtype=np.float32)
    r = np.array([[0, 0, 0], [0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 0, 0],
                  [0, 0, 0], [1, 0, 0], [1, 0, 0],
                  [0, 0, 0], [0, 0, 0.25], [0, 0, 0]], dtype=np.float32)
    D, I = _check(dev, q, r, 3)
    assert I[0, 0] == 6 and D[0, 0] == 1.0 and I[1, 0] == 3 and D[1, 0] == 2.0 and D[2, 0] == 0.25


def test_knn_id_offset_and_empty(dev):
    from vsc_hip import ops
    q = torch.from_numpy(synth.descriptor_bank(1, 4, 32)).to(dev)
    r = torch.from_numpy(synth.descriptor_bank(2, 100, 32)).to(dev)
    _, I0 = ops.knn_ip(q, r, 5)
    _, I1 = ops.knn_ip(q, r, 5, ref_id_offset=1_000_000_000_000)
    assert torch.equal(I1, I0 + 1_000_000_000_000)
    D, I = ops.knn_ip(q[:0], r, 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)
    D, I = ops.knn_ip(q, r[:0], 5)
    assert (I == -1).all()


def test_knn_linearity_property_full_width(dev):
    """Size-independent property at a size the oracle does not visit: the top-1 of a bank
    that contains the query itself is the query (score = |q|^2 chain), and searching a
    permuted bank returns the permuted ids."""
    from vsc_hip import ops
    nr = 200_000
    r = synth.descriptor_bank(77, nr, 512)
    perm = np.random.RandomState(0).permutation(nr)
    qidx = np.arange(0, nr, 997)[:128]
    q = r[qidx]
    rt = torch.from_numpy(r).to(dev)
    D, I = ops.knn_ip(torch.from_numpy(q).to(dev), rt, 4)
    assert (I[:, 0].cpu().numpy() == qidx).all()
    D2, I2 = ops.knn_ip(torch.from_numpy(q).to(dev), rt[torch.from_numpy(perm).to(dev)], 4)
    assert torch.equal(D, D2)
    assert (perm[I2.cpu().numpy()] == I.cpu().numpy()).all()


@pytest.mark.parametrize("nq,nr,d,radius", [(9, 200, 64, 0.1), (130, 5000, 511, 0.08), (1, 70000, 512, 0.12),
                                            (300, 3000, 32, -2.0), (64, 1000, 512, 0.9)])
def test_range_search_bit_exact(dev, nq, nr, d, radius):
    """Every pair above the radius, CSR layout, ascending ref id: identical to the oracle."""
    from oracle import knn_oracle
    from vsc_hip import ops
    q, r = synth.descriptor_bank(300 + nq, nq, d), synth.descriptor_bank(400 + nr, nr, d)
    lims, D, I = ops.range_search_ip(torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), radius, capacity=16)
    lr, Dr, Ir = knn_oracle.range_search_ip(q, r, radius)
    assert np.array_equal(lims.cpu().numpy(), lr)
    assert np.array_equal(I.cpu().numpy(), Ir)
    assert np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))
    if radius < -1.5:
        assert lr[-1] == nq * nr     # everything matches: the dense limit


@pytest.mark.parametrize("nq,nr,d,radius,expect", [
    (130, 5000, 511, 0.08, 2),       # the bit-exact case above, pre-filter path forced: d not a multiple of 64, ragged tiles
    (600, 70000, 512, 0.12, 2),      # several query blocks and reference splits
    (300, 3000, 32, -2.0, 2),        # every pair matches, lists of 3000 / splits stay under the capacity
    (64, 300000, 64, -0.2, 3),       # ~95 % of 64-d pairs above -0.2: 1.2 k survivors per 1280-reference list -> overflow, the exact path answers
])
def test_range_search_prefilter_path_equals_exact(dev, nq, nr, d, radius, expect):
    """vsc_range_search_ip_f32 through the bf16 pre-filter (one fixed-threshold bf16 sweep + exact re-scoring + scan + emit)
    returns the CSR of the exact two-sweep path bit for bit; when a list overflows it hands the call to the exact path."""
    import os
    from vsc_hip import ops, _lib
    q, r = synth.descriptor_bank(300 + nq, nq, d), synth.descriptor_bank(400 + nr, nr, d)
    qt, rt = torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev)
    out = {}
    for path in ("exact", "bf16"):
        _vsc_lib.set_option("VSC_RANGE_PATH", path)
        try:
            out[path] = [t.cpu() for t in ops.range_search_ip(qt, rt, radius, ref_id_offset=7, capacity=16)]
            ran = _lib.require_device().vsc_range_search_last_path()
            count = ops.range_count_ip(qt, rt, radius)
        finally:
            _vsc_lib.set_option("VSC_RANGE_PATH", None)
        assert ran == (1 if path == "exact" else expect)
        assert count == int(out[path][0][-1])
    assert torch.equal(out["exact"][0], out["bf16"][0]) and torch.equal(out["exact"][2], out["bf16"][2])
    assert torch.equal(out["exact"][1].view(torch.int32), out["bf16"][1].view(torch.int32))
    assert int(out["exact"][0][-1]) > 0


def test_video_index_reference_vectors(dev):
    """The reference's own unit tests, run through VideoIndex / CandidateGeneration on the HIP path:
    tests/test_candidates.py (expected candidate list) and tests/test_index.py (self match)."""
    from vsc.candidates import CandidateGeneration, MaxScoreAggregation
    from vsc.index import VideoFeature, VideoIndex
    from vsc.metrics import CandidatePair
    queries = [VideoFeature(video_id=1, feature=np.eye(3, dtype=np.float32), timestamps=np.array([0.0, 1.0, 2.0]))]
    refs = [VideoFeature(video_id=5, feature=np.array([[0, 0, 0], [0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 0, 0]], np.float32),
                         timestamps=np.array([2.0, 4.0, 6.0, 8.0, 10.0])),
            VideoFeature(video_id=8, feature=np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0]], np.float32),
                         timestamps=np.array([[0.0, 5.0], [5.0, 10.0], [10.0, 15.0]])),
            VideoFeature(video_id=10, feature=np.array([[0, 0, 0], [0, 0, 0.25], [0, 0, 0]], np.float32),
                         timestamps=np.array([0.0, 0.1, 0.2]))]
    cands = CandidateGeneration(refs, MaxScoreAggregation()).query(queries, 2 * 3)
    assert cands == [CandidatePair(query_id=1, ref_id=5, score=2.0), CandidatePair(query_id=1, ref_id=8, score=1.0),
                     CandidatePair(query_id=1, ref_id=10, score=0.25)]

    from oracle import knn_oracle
    feats = np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[11, 12, 13], [14, 15, 16], [17, 18, 19]],
                      [[111, 112, 113], [114, 115, 116], [117, 118, 119]]], np.float32)
    feats = knn_oracle.l2_normalize(feats.reshape(9, 3)).reshape(3, 3, 3)
    mk = lambda pre: [VideoFeature(video_id=f"{pre}{i:06d}", feature=f, timestamps=np.arange(3, dtype=np.float32))
                      for i, f in enumerate(feats)]
    for gk in (1, -1):
        idx = VideoIndex(3)
        idx.add(mk("R"))
        for res in idx.search(mk("Q"), gk):
            assert res.query_id[1:] == res.ref_id[1:]


def test_global_threshold_search_is_the_global_top_k(dev):
    """_global_threshold_knn_search == the global_k best pairs of the full score matrix, including
    the case where one query row owns more winners than the per-row probe (range-sweep path)."""
    from oracle import knn_oracle
    from vsc.index import VideoFeature, VideoIndex
    import vsc.index as vi
    r = synth.descriptor_bank(5, 4000, 32)
    q = synth.descriptor_bank(6, 40, 32)
    r[100:1400] = q[3] + 0.01 * synth.normalish(7, (1300, 32))   # 1300 refs glued to query row 3
    idx = VideoIndex(32)
    idx.add([VideoFeature("R1", np.arange(4000.0), r)])
    S = knn_oracle.ip_matrix(q, r)
    for gk, probe in ((50, 1024), (1200, 64)):
        old, vi.MAX_K = vi.MAX_K, probe
        try:
            hits = idx._global_threshold_knn_search(q, gk)
        finally:
            vi.MAX_K = old
        flat = np.argsort(-S.ravel().astype(np.float64), kind="stable")[:gk]
        want = sorted(((int(f // 4000), int(f % 4000)) for f in flat))
        assert sorted((i, j) for i, j, _ in hits) == want
        assert all(s == S[i, j] for i, j, s in hits)


def test_candidate_generation_fast_path_equals_the_object_path(dev):
    """CandidateGeneration.query under MaxScoreAggregation (what sscd_baseline.search runs) builds its list from flat hit arrays
    (VideoIndex.search_pair_maxima); a subclass of the aggregation takes the reference's object path (a PairMatch per frame hit,
    grouped, aggregated, stably sorted).  Same pairs, same scores, same order -- with a query video glued to a reference video (one
    row owning far more winners than the adaptive probe holds), exact duplicates across videos (tied maxima) and several
    global_k, one of them larger than the number of pairs."""
    from vsc.candidates import CandidateGeneration, MaxScoreAggregation
    from vsc.index import VideoFeature

    class SameMax(MaxScoreAggregation):      # not `type(...) is MaxScoreAggregation`: the object path
        pass

    d = 64
    rb = synth.descriptor_bank(51, 40 * 30, d)
    qb = synth.descriptor_bank(52, 12 * 10, d)
    rb[300:330] = qb[25] + 0.01 * synth.normalish(53, (30, d))    # reference video 10 glued to one frame of query video 2
    rb[600] = rb[30]                                               # duplicate reference frames in two videos
    qb[77] = rb[30]                                                # ... matched exactly by a query frame (tied maxima 1.0-ish)
    qb[5] = rb[900]
    refs = [VideoFeature(f"R{i:03d}", np.arange(30.0), rb[30 * i:30 * i + 30]) for i in range(40)]
    queries = [VideoFeature(f"Q{i:03d}", np.arange(10.0), qb[10 * i:10 * i + 10]) for i in range(12)]
    fast, slow = CandidateGeneration(refs, MaxScoreAggregation()), CandidateGeneration(refs, SameMax())
    for gk in (1, 40, 700, 5000, 10 ** 7):
        a, b = fast.query(queries, gk), slow.query(queries, gk)
        assert [(c.query_id, c.ref_id, c.score) for c in a] == [(c.query_id, c.ref_id, c.score) for c in b], gk
        assert len(a) > 0 and all(x.score >= y.score for x, y in zip(a, a[1:]))
        assert fast.query(queries, gk, limit=7) == a[:7] and slow.query(queries, gk, limit=7) == b[:7]
    # two VideoFeatures under ONE video id (a video stored in two pieces): the object path groups by id -- so must the default path
    split = queries[:3] + [VideoFeature("Q002", np.arange(10.0), qb[30:40])] + queries[4:]
    a, b = fast.query(split, 700), slow.query(split, 700)
    assert [(c.query_id, c.ref_id, c.score) for c in a] == [(c.query_id, c.ref_id, c.score) for c in b]
    assert len({(c.query_id, c.ref_id) for c in a}) == len(a)
    # an L2 index: the first hit of a pair in best-first order is its SMALLEST distance, MaxScoreAggregation keeps the largest
    import vsc.index as vi
    fast2, slow2 = CandidateGeneration(refs[:6], MaxScoreAggregation()), CandidateGeneration(refs[:6], SameMax())
    for cg in (fast2, slow2):
        cg.index = vi.VideoIndex(d, metric=vi.METRIC_L2)
        cg.index.add(refs[:6])
    a, b = fast2.query(queries[:4], 300), slow2.query(queries[:4], 300)
    assert [(c.query_id, c.ref_id) for c in a] == [(c.query_id, c.ref_id) for c in b]
    np.testing.assert_allclose([c.score for c in a], [c.score for c in b], rtol=0, atol=0)


def test_global_threshold_search_when_the_probe_is_smaller_than_global_k(dev):
    """nq * k' < global_k <= nq * nr (few query rows, the per-row probe capped at MAX_K): the reference returns
    min(global_k, nq * nr) pairs (index.py:145-165); the probe alone would truncate to nq * k'."""
    from oracle import knn_oracle
    from vsc.index import VideoFeature, VideoIndex
    import vsc.index as vi
    r = synth.descriptor_bank(15, 3000, 48)
    for nq, gk, probe in ((1, 1500, 1024), (3, 700, 128), (2, 5999, 64), (2, 10000, 64)):
        q = synth.descriptor_bank(16 + nq, nq, 48)
        idx = VideoIndex(48)
        idx.add([VideoFeature("R1", np.arange(3000.0), r)])
        S = knn_oracle.ip_matrix(q, r)
        old, vi.MAX_K = vi.MAX_K, probe
        try:
            hits = idx._global_threshold_knn_search(q, gk)
        finally:
            vi.MAX_K = old
        n = min(gk, nq * 3000)
        assert len(hits) == n
        flat = np.argsort(-S.ravel().astype(np.float64), kind="stable")[:n]
        assert sorted((i, j) for i, j, _ in hits) == sorted((int(f // 3000), int(f % 3000)) for f in flat)
        assert all(s == S[i, j] for i, j, s in hits)
        assert all(a[2] >= b[2] for a, b in zip(hits, hits[1:]))


def test_flat_l2_index(dev):
    """METRIC_L2 (the reference's tests/test_index.py builds VideoIndex(3, "Flat", faiss.METRIC_L2) on UN-normalised
    vectors): exact squared distances ascending, ids of a brute-force float64 scan, through search, range_search and
    both VideoIndex search modes."""
    from vsc.index import METRIC_L2, FlatIPBank, VideoFeature, VideoIndex
    rng = np.random.RandomState(3)
    r = (rng.randn(500, 24) * 3).astype(np.float32)
    q = (rng.randn(17, 24) * 3).astype(np.float32)
    q[5] = r[77]
    bank = FlatIPBank(24, METRIC_L2)
    bank.add(r[:200])
    bank.add(r[200:])
    D, I = bank.search(q, 9)
    d64 = ((q[:, None, :].astype(np.float64) - r[None].astype(np.float64)) ** 2).sum(-1)
    want = np.argsort(d64, axis=1, kind="stable")[:, :9]
    assert np.array_equal(I, want) and I[5, 0] == 77 and D[5, 0] == 0.0
    assert np.allclose(D, np.take_along_axis(d64, want, 1), rtol=1e-5, atol=1e-5) and (np.diff(D, axis=1) >= 0).all()
    rows, ids, dist = bank.range_search(q, 120.0)
    assert sorted(zip(rows.tolist(), ids.tolist())) == sorted(map(tuple, np.argwhere(d64 < 120.0).tolist()))
    assert np.allclose(dist, d64[rows, ids], rtol=1e-5, atol=1e-5)
    assert bank.range_count(q, 120.0) == len(rows)

    feats = np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[11, 12, 13], [14, 15, 16], [17, 18, 19]],
                      [[111, 112, 113], [114, 115, 116], [117, 118, 119]]], np.float32)   # tests/test_index.py, verbatim data
    mk = lambda pre: [VideoFeature(video_id=f"{pre}{i:06d}", feature=f, timestamps=np.arange(3, dtype=np.float32))
                      for i, f in enumerate(feats)]
    for gk in (1, -1, 4):
        idx = VideoIndex(3, "Flat", METRIC_L2)
        idx.add(mk("R"))
        res = idx.search(mk("Q"), gk)
        assert res and all(x.query_id[1:] == x.ref_id[1:] for x in res)
    hits = idx._global_threshold_knn_search(feats.reshape(9, 3), 9)
    assert sorted((i, j) for i, j, _ in hits) == [(i, i) for i in range(9)] and all(s == 0.0 for _, _, s in hits)


def test_eval_entry_point_end_to_end(dev, tmp_path):
    """`python -m vsc.baseline.sscd_baseline` (what infer/eval.sh runs): .npz in, candidates.csv out;
    planted copies are the top candidates and uAP is 1.0; score-normalised run agrees with the
    oracle's statement of score_normalize."""
    import vsc.baseline.sscd_baseline as entry
    from oracle import knn_oracle
    from vsc.baseline import score_normalization as sn
    from vsc.index import VideoFeature
    from vsc.metrics import CandidatePair
    from vsc.storage import load_features, store_features
    rs = np.random.RandomState(3)
    dim = 64
    refs = [VideoFeature(f"R{i:06d}", np.arange(12.0), synth.descriptor_bank(500 + i, 12, dim)) for i in range(40)]
    noise = [VideoFeature(f"R{i:06d}", np.arange(10.0), synth.descriptor_bank(900 + i, 10, dim)) for i in range(100, 130)]
    queries = []
    for i in range(10):
        f = synth.descriptor_bank(700 + i, 8, dim)
        if i < 4:   # queries 0..3 copy frames of refs 5, 6, 7, 8
            f[2:6] = refs[5 + i].feature[3:7] + 0.01 * rs.randn(4, dim).astype(np.float32)
        queries.append(VideoFeature(f"Q{i:06d}", np.arange(8.0), f))
    store_features(tmp_path / "q.npz", queries)
    store_features(tmp_path / "r.npz", refs)
    store_features(tmp_path / "n.npz", noise)
    gt = tmp_path / "gt.csv"
    gt.write_text("query_id,ref_id,query_start,query_end,ref_start,ref_end\n" +
                  "".join(f"Q{i:06d},R{5 + i:06d},2,6,3,7\n" for i in range(4)))
    args = entry.build_parser().parse_args(["--query_features", str(tmp_path / "q.npz"), "--ref_features",
                                            str(tmp_path / "r.npz"), "--output_path", str(tmp_path / "out"),
                                            "--ground_truth", str(gt), "--overwrite"])
    entry.main(args)
    cands = CandidatePair.read_csv(tmp_path / "out" / "candidates.csv")
    assert {(c.query_id, c.ref_id) for c in cands[:4]} == {(f"Q{i:06d}", f"R{5 + i:06d}") for i in range(4)}
    assert all(a.score >= b.score for a, b in zip(cands, cands[1:]))
    assert entry.micro_average_precision(entry.read_ground_truth_pairs(str(gt)), cands) == 1.0

    # score normalisation: compare with a numpy/oracle restatement of score_normalization.py:34-110
    q2, r2 = sn.score_normalize(queries, refs, noise, beta=1.2)
    bank = np.concatenate([n.feature for n in noise])
    dim_drop = int(bank.var(axis=0).argmin())
    drop = lambda x: knn_oracle.l2_normalize(np.delete(x, dim_drop, axis=1))
    nb = drop(bank)
    for q, qn in zip(queries, q2):
        f = drop(q.feature)
        D, _ = knn_oracle.knn_ip(f, nb, 1)
        want = np.concatenate([f, -1.2 * D[:, :1]], axis=1)
        np.testing.assert_allclose(qn.feature, want, rtol=0, atol=2e-6)
    assert all(r.feature.shape[1] == dim and (r.feature[:, -1] == 1).all() for r in r2)
    # the videos are normalised a block at a time (normalize_videos): the same bits as video by video, for any block size, empty videos included
    plus_empty = queries + [VideoFeature("Q000099", np.arange(0.0), np.zeros((0, dim), np.float32))]
    one_by_one = sn.transform_features(plus_empty[:-1], sn.normalize)
    for rows in (1, 13, 1 << 20):
        blocks = sn.normalize_videos(plus_empty, block_rows=rows)
        assert len(blocks) == len(plus_empty) and blocks[-1].feature.shape == (0, dim)
        assert all(np.array_equal(a.feature, b.feature) for a, b in zip(blocks, one_by_one))
    with pytest.raises(Exception, match="against VSC rules"):
        sn.score_normalize(queries, refs, refs)


def test_query_postprocess_on_device(dev):
    """Frame de-duplication with the similarity matrix and normalisation computed by the HIP ops."""
    from src.query_postprocess import HipOps, select_frames
    base = synth.normalish(3, (40, 64))
    frames = np.concatenate([base, base[:10] * 1.5, base[20:25] + 1e-4])   # scaled copies are duplicates in cosine
    keep = select_frames(frames, HipOps)
    S = HipOps.self_similarity(HipOps.normalize(frames))
    from oracle import knn_oracle
    assert np.array_equal(S.view(np.uint32), knn_oracle.ip_matrix(knn_oracle.l2_normalize(frames), knn_oracle.l2_normalize(frames)).view(np.uint32)) or \
        np.allclose(S, knn_oracle.ip_matrix(knn_oracle.l2_normalize(frames), knn_oracle.l2_normalize(frames)), atol=2e-6)
    assert len(keep) == 40 and len(set(keep)) == 40


def test_hip_pca_on_device(dev):
    """PCA apply (2048 -> 512 in the reference) through vsc_pair_similarity_f32 vs float64."""
    from src.query_postprocess import HipPCA

    class Fitted:
        mean_ = synth.normalish(31, (256,)) * 0.1
        components_ = synth.normalish(32, (64, 256)) / 16.0
        whiten = True
        explained_variance_ = np.linspace(2.0, 0.1, 64)

    x = synth.normalish(33, (130, 256))
    got = HipPCA(Fitted).transform(x)
    want = ((x.astype(np.float64) - Fitted.mean_) @ Fitted.components_.astype(np.float64).T) / np.sqrt(Fitted.explained_variance_)
    assert got.shape == (130, 64) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_concat_pca_sn_entry_point(dev, tmp_path):
    """infer/concat_pca_sn.py mirror end to end on small synthetic per-model .npz files: merged descriptors equal the
    numpy statement (normalise, concatenate, sklearn PCA.transform), score-normalised files carry the extra column."""
    import argparse
    import concat_pca_sn as C
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import normalize
    from vsc.index import VideoFeature
    from vsc.storage import load_features, store_features
    models, dims = ["m_a", "m_b"], [24, 40]
    vids = {"train_refs": ["R100001", "R100002", "R100003"], "test_refs": ["R200001", "R200002"]}
    raw = {}
    for mi, (m, d) in enumerate(zip(models, dims)):
        os.makedirs(tmp_path / m)
        for si, (name, ids) in enumerate(vids.items()):
            feats = [VideoFeature(video_id=v, timestamps=np.arange(6 + vi, dtype=np.float64),
                                  feature=synth.normalish(1000 * mi + 100 * si + vi, (6 + vi, d)) * (1 + vi)) for vi, v in enumerate(ids)]
            raw[(m, name)] = feats
            store_features(str(tmp_path / m / f"{name}.npz"), feats)
    args = argparse.Namespace(root=str(tmp_path), models=models, pca_model=str(tmp_path / "pca.pkl"), fit_pca=True, dim=16)
    C.main(args)
    import pickle
    fitted = pickle.load(open(tmp_path / "pca.pkl", "rb"))
    for name, ids in vids.items():
        merged = {vf.video_id: vf for vf in load_features(str(tmp_path / f"{name}.npz"))}
        assert sorted(merged) == sorted(ids)
        for vi, v in enumerate(ids):
            cat = np.concatenate([normalize(raw[(m, name)][vi].feature) for m in models], axis=1)
            np.testing.assert_allclose(merged[v].feature, fitted.transform(cat), rtol=1e-4, atol=2e-5)
        sn = load_features(str(tmp_path / f"{name}_sn.npz"))
        assert all(vf.feature.shape[1] == 16 for vf in sn)   # one low-variance dim replaced by the bias column
    # the videos of a block go through the normalisation and the PCA together: the same bits as one video at a time, for any block size
    from src.query_postprocess import HipPCA
    pca = HipPCA(fitted)
    paths = [str(tmp_path / m / "train_refs.npz") for m in models]
    whole = C.merge_set(paths, pca.transform)
    for rows in (1, 7, 14):
        part = C.merge_set(paths, pca.transform, block_rows=rows)
        assert [v.video_id for v in part] == [v.video_id for v in whole]
        assert all(np.array_equal(a.feature, b.feature) and np.array_equal(a.timestamps, b.timestamps) for a, b in zip(part, whole))
    per_video = [pca.transform(np.concatenate([C.HipOps.normalize(raw[(m, "train_refs")][vi].feature) for m in models], axis=1)) for vi in range(3)]
    assert all(np.array_equal(a.feature, b) for a, b in zip(whole, per_video))


def test_search_scratch_release(dev):
    """vsc_search_release_scratch frees the grow-only search scratch of the device; the next call allocates again and returns
    the same bits."""
    from vsc_hip import _lib, ops
    lib = _lib.require_device()
    q = torch.from_numpy(synth.descriptor_bank(1, 300, 128)).to(dev)
    r = torch.from_numpy(synth.descriptor_bank(2, 70000, 128)).to(dev)
    D0, I0 = ops.knn_ip(q, r, 20)
    freed = lib.vsc_search_release_scratch()
    assert freed >= 70000 * 128 * 2
    assert lib.vsc_search_release_scratch() == 0
    D1, I1 = ops.knn_ip(q, r, 20)
    assert torch.equal(I0, I1) and torch.equal(D0.view(torch.int32), D1.view(torch.int32))
