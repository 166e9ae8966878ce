"""Descriptor files: the reference's .npz layout (video_ids / features / timestamps, one row
per frame, rows of a video contiguous) -- infer/vsc/storage.py:14-69.  Files written here
load in the reference and vice versa."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from vsc.index import VideoFeature
from vsc.metrics import Dataset, format_video_id


def store_features(f, features: List[VideoFeature], dataset: Optional[Dataset] = None) -> None:
    ids, feats, stamps = [], [], []
    for vf in features:
        ids.append(np.full(len(vf), format_video_id(vf.video_id, dataset)))
        feats.append(vf.feature)
        stamps.append(vf.timestamps)
    np.savez(f, video_ids=np.concatenate(ids), features=np.concatenate(feats).astype(np.float32),
             timestamps=np.concatenate(stamps))


def same_value_ranges(values):
    """(value, start, end) for each run of equal consecutive values."""
    start = 0
    for i in range(1, len(values) + 1):
        if i == len(values) or values[i] != values[start]:
            yield values[start], start, i
            start = i


def load_features(f, dataset: Optional[Dataset] = None) -> List[VideoFeature]:
    data = np.load(f, allow_pickle=False)
    ids, feats, stamps = data["video_ids"], data["features"].astype(np.float32), data["timestamps"]
    if stamps.shape[0] != feats.shape[0]:
        raise ValueError(f"Expected the same number of timestamps as features: got {stamps.shape[0]} "
                         f"timestamps for {feats.shape[0]} features")
    if not (stamps.ndim == 1 or stamps.shape[1:] == (2,)):
        raise ValueError(f"Unexpected timestamp shape. Got {stamps.shape}")
    return [VideoFeature(video_id=format_video_id(v, dataset), timestamps=stamps[a:b], feature=feats[a:b, :])
            for v, a, b in same_value_ranges(ids)]
