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
