"""Similarity search parity: vsc_knn_ip_f32 must be BIT-EXACT with oracle/knn_oracle.c
(scores as uint32 patterns, ids as int64), including ties, ragged shapes and k > nr."""
import numpy as np
import pytest
import torch

from src import synth

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
