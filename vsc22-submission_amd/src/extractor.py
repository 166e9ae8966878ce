"""Frame -> descriptor extraction loop (reference: infer/src/extractor.py:9-38).

`model` is anything with the reference's call shape -- here a vsc_hip.encoder.HipEncoder.
A batch is (frames [B,S,3,H,W] padded, mask [B,S], video_ids) exactly as D_vsc.collate_fn
builds it (infer/src/dataset.py:145-153).

The reference encodes loader batch by loader batch (two videos: ~100 frames per call, a synchronous pageable upload in front and a
device -> host copy behind every one).  Here the valid frames of consecutive loader batches are collected, on the host, into groups of
>= `group_frames` frames (4 096: the ragged last chunk of a group is then a percent of it) and go through `src.query_pipeline.encode_group`: pinned staging + a copy stream, encoder calls of the
backbone's aligned chunk, one device -> host copy per group.  A frame's descriptor does not depend on WHICH frames share its call, but
the encoders choose kernels by a call's row count (Swin-V2's 512-wide stage: the one-launch second half from 77 frames per chunk on, GEMM
launches below; the ViT's persistent GEMMs need more tiles than CUs) and the two forms round in different orders: the same frame in a
large and in a small call agrees to rounding-order noise (measured <= 2e-4 with bf16 operands, <= 4e-5 with fp16:
tests/test_gpu_swin.py::test_a_frame_alone_and_inside_a_full_chunk), not bit for bit."""
from __future__ import annotations

from typing import Iterable, List, Tuple

import numpy as np
import torch


def extract_vsc_feat(model, batches: Iterable, device, group_frames: int = 4096) -> Tuple[List[str], np.ndarray, np.ndarray]:
    from src.query_pipeline import encode_group
    feats, vids, stamps = [], [], []
    group, have = [], 0
    pending = None      # the previous group's descriptors, still on the device

    def collect():
        nonlocal pending
        if pending is not None:
            feats.append(torch.cat(pending).cpu().numpy() if len(pending) > 1 else pending[0].cpu().numpy())
            pending = None

    def flush():
        # One group of look-ahead (round 6): a group's uploads and launches are QUEUED here (device tensors back, no wait); its descriptors
        # are copied back when the next group has been queued behind it -- so the loader's next batches are collated while this group is
        # encoded, instead of behind it.  On the CPU (host-logic tests with fake encoders) nothing is asynchronous and nothing changes.
        nonlocal group, have, pending
        if group:
            on_gpu = torch.device(device).type == "cuda"
            outs = encode_group([model], group, device, as_numpy=not on_gpu)[0]
            collect()
            if on_gpu:
                pending = [o for o in outs if o.shape[0] > 0]
            else:
                feats.extend(outs)
        group, have = [], 0

    for frames, mask, video_id in batches:
        mask = mask.bool()
        counts = mask.sum(dim=1).tolist()
        for b, (v, c) in enumerate(zip(video_id, counts)):
            c = int(c)
            vids.extend([v] * c)
            stamps.append(np.arange(c))
            if c:
                f = frames[b]
                group.append(f[:c] if bool(mask[b, :c].all()) else f[mask[b]])   # (the collate pads at the end: a view, no copy)
                have += c
        if have >= group_frames:
            flush()
    flush()
    collect()
    if not stamps:
        return [], np.zeros((0, 0), np.float32), np.zeros((0,), np.int64)
    if not feats:
        return vids, np.zeros((0, 0), np.float32), np.concatenate(stamps)
    return vids, np.concatenate(feats), np.concatenate(stamps)
