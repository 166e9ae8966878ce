"""world_size-2 gloo runs (CPU) of the multi-GPU host logic: frame sharding, the bank all_gather
(ragged shards) and the sharded search, with the oracle standing in for the HIP sweep through
the documented test hook."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tools import synth
from vsc_hip import distributed as vdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_knn(q, r, k):
    from oracle import knn_oracle
    D, I = knn_oracle.knn_ip(q.numpy(), r.numpy(), k)
    return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        refs = torch.from_numpy(synth.descriptor_bank(1, 301, 64))      # 301 refs: ragged shards 151 / 150
        qs = torch.from_numpy(synth.descriptor_bank(2, 37, 64))
        rlo, rhi = vdist.shard_bounds(301, rank, world_size)
        qlo, qhi = vdist.shard_bounds(37, rank, world_size)
        bank, offsets = vdist.all_gather_rows(refs[rlo:rhi])
        assert torch.equal(bank, refs) and offsets.tolist() == [0, 151, 301]
        D, I = vdist.sharded_knn(qs[qlo:qhi], refs[rlo:rhi], 5, knn=_oracle_knn)
        if rank == 0:
            np.savez(os.path.join(out_dir, "res.npz"), D=D.numpy(), I=I.numpy())
        else:
            assert D is None and I is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker8(rank, world_size, port, out_dir):
    """8 ranks, 5-row and 1003-row banks: empty shards, ragged shards, and the result gather of the sharded search."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        for n in (5, 1003):
            refs = torch.from_numpy(synth.descriptor_bank(3, n, 32))
            lo, hi = vdist.shard_bounds(n, rank, world_size)
            bank, offsets = vdist.all_gather_rows(refs[lo:hi])
            assert torch.equal(bank, refs)
            assert offsets.tolist() == [vdist.shard_bounds(n, r, world_size)[0] for r in range(world_size)] + [n]
        qs = torch.from_numpy(synth.descriptor_bank(4, 11, 32))          # 11 queries over 8 ranks: 2, 2, 2, 1, 1, 1, 1, 1
        qlo, qhi = vdist.shard_bounds(11, rank, world_size)
        D, I = vdist.sharded_knn(qs[qlo:qhi], refs[lo:hi], 7, knn=_oracle_knn)
        if rank == 0:
            np.savez(os.path.join(out_dir, "res8.npz"), D=D.numpy(), I=I.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker8_one_owner(rank, world_size, port, out_dir):
    """Every rank's reference shard is EMPTY except rank 5's (one process encoded every reference video); queries on ranks 0
    and 7 only.  The padded all_gather, the id numbering (position in rank order) and the result gather must still hold."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        refs = torch.from_numpy(synth.descriptor_bank(5, 77, 32))
        qs = torch.from_numpy(synth.descriptor_bank(6, 9, 32))
        mine = refs if rank == 5 else refs[:0]
        bank, offsets = vdist.all_gather_rows(mine)
        assert torch.equal(bank, refs) and offsets.tolist() == [0] * 6 + [77, 77, 77]
        q_mine = qs[:4] if rank == 0 else (qs[4:] if rank == 7 else qs[:0])
        D, I = vdist.sharded_knn(q_mine, mine, 6, knn=lambda q, r, k: _oracle_knn(q, r, k) if len(q) else
                                 (torch.empty((0, k)), torch.empty((0, k), dtype=torch.int64)))
        if rank == 0:
            np.savez(os.path.join(out_dir, "res8b.npz"), D=D.numpy(), I=I.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_search_world8_single_owner_of_the_bank(tmp_path):
    port = _free_port()
    mp.spawn(_worker8_one_owner, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    got = np.load(tmp_path / "res8b.npz")
    from oracle import knn_oracle
    D, I = knn_oracle.knn_ip(synth.descriptor_bank(6, 9, 32), synth.descriptor_bank(5, 77, 32), 6)
    assert np.array_equal(got["I"], I) and np.array_equal(got["D"], D)


def test_sharded_search_world8_ragged_and_empty_shards(tmp_path):
    """BASELINE.json configs[3]'s host logic at the node's rank count: the bank all_gather with ragged and EMPTY shards
    (fewer videos than ranks) and the sharded search, on gloo."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    got = np.load(tmp_path / "res8.npz")
    from oracle import knn_oracle
    D, I = knn_oracle.knn_ip(synth.descriptor_bank(4, 11, 32), synth.descriptor_bank(3, 1003, 32), 7)
    assert np.array_equal(got["I"], I) and np.array_equal(got["D"], D)


def test_ref_feature_merge_skips_empty_rank_parts(tmp_path):
    """extract_ref_feats.py's rank-0 merge when a rank had nothing to encode (its part holds a (0, 0) block)."""
    from src.extractor import extract_vsc_feat
    ids, feats, stamps = extract_vsc_feat(lambda x: x, [], torch.device("cpu"))
    assert ids == [] and feats.shape[0] == 0 and stamps.shape == (0,)
    np.savez(tmp_path / "p_0.npz", video_ids=np.array(["R1", "R1"]), features=np.ones((2, 4), np.float32), timestamps=np.arange(2))
    np.savez(tmp_path / "p_1.npz", video_ids=ids, features=feats, timestamps=stamps)
    parts = [np.load(tmp_path / f"p_{i}.npz") for i in range(2)]
    parts = [p for p in parts if len(p["features"])] or parts[:1]         # the merge rule of extract_ref_feats.py
    merged = np.concatenate([p["features"] for p in parts])
    assert merged.shape == (2, 4) and np.concatenate([p["video_ids"] for p in parts]).tolist() == ["R1", "R1"]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1000):
        for ws in (1, 2, 3, 8):
            spans = [vdist.shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_search_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "res.npz")
    from oracle import knn_oracle
    D, I = knn_oracle.knn_ip(synth.descriptor_bank(2, 37, 64), synth.descriptor_bank(1, 301, 64), 5)
    assert np.array_equal(got["I"], I) and np.array_equal(got["D"], D)


def test_extractor_loop_with_padding():
    """extract_vsc_feat drops padded frames and keeps video order (reference extractor.py:9-38)."""
    from src.dataset import TensorFrames, collate_fn
    from src.extractor import extract_vsc_feat
    vids = [(torch.full((3, 3, 4, 4), 1.0), "R000001"), (torch.full((5, 3, 4, 4), 2.0), "R000002")]
    loader = torch.utils.data.DataLoader(TensorFrames(vids), batch_size=2, collate_fn=collate_fn)
    model = lambda x: x.mean(dim=(1, 2, 3))[:, None].repeat(1, 8)
    ids, feats, stamps = extract_vsc_feat(model, loader, torch.device("cpu"))
    assert ids == ["R000001"] * 3 + ["R000002"] * 5
    assert stamps.tolist() == [0, 1, 2, 0, 1, 2, 3, 4]
    assert feats.shape == (8, 8) and (feats[:3] == 1).all() and (feats[3:] == 2).all()


def test_vit_transform_matches_reference_definition():
    from PIL import Image
    from src.dataset import vit_transform
    rs = np.random.RandomState(0)
    img = Image.fromarray(rs.randint(0, 255, (30, 50, 3), dtype=np.uint8))
    x = vit_transform(32, 32)(img)
    ref = np.asarray(img.resize((32, 32), Image.BICUBIC), dtype=np.float32).transpose(2, 0, 1) / 255.0
    np.testing.assert_allclose(x.numpy(), (ref - 0.5) / 0.5, atol=1e-6)
    assert x.shape == (3, 32, 32) and x.min() >= -1 and x.max() <= 1
