"""Frame -> descriptor extraction loop (reference: infer/src/extractor.py:9-38).

`model` is anything with the reference's call shape -- here a vsc_hip.encoder.HipEncoder.
A batch is (frames [B,S,3,H,W] padded, mask [B,S], video_ids) exactly as D_vsc.collate_fn
builds it (infer/src/dataset.py:145-153)."""
from __future__ import annotations

from typing import Iterable, List, Tuple

import numpy as np
import torch


def extract_vsc_feat(model, batches: Iterable, device) -> Tuple[List[str], np.ndarray, np.ndarray]:
    feats, vids, stamps = [], [], []
    for frames, mask, video_id in batches:
        mask = mask.to(device).bool()
        counts = mask.sum(dim=1).tolist()
        flat = frames.to(device)[mask]              # drop the padding frames
        out = model(flat)
        assert out.shape[0] == sum(counts)
        feats.append(out.detach().float().cpu().numpy())
        for v, c in zip(video_id, counts):
            vids.extend([v] * int(c))
            stamps.append(np.arange(int(c)))
    if not feats:
        return [], np.zeros((0, 0), np.float32), np.zeros((0,), np.int64)
    return vids, np.concatenate(feats), np.concatenate(stamps)
