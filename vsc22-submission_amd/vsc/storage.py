"""Descriptor files: the reference's .npz layout (video_ids / features / timestamps, one row
per frame, rows of a video contiguous) -- infer/vsc/storage.py:14-69.  Files written here
load in the reference and vice versa."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from vsc.index import VideoFeature
from vsc.metrics import Dataset, format_video_id


def store_features(f, features: List[VideoFeature], dataset: Optional[Dataset] = None) -> None:
    """One row per frame; the id column repeats each video's (formatted) id over its rows."""
    names = np.array([format_video_id(video.video_id, dataset) for video in features])
    rows = np.array([len(video) for video in features], dtype=np.int64)
    np.savez(f,
             video_ids=np.repeat(names, rows),
             features=np.concatenate([video.feature for video in features]).astype(np.float32),
             timestamps=np.concatenate([video.timestamps for video in features]))


def same_value_ranges(values):
    """(value, start, end) for each run of equal consecutive values (vectorised: banks hold millions of rows)."""
    values = np.asarray(values)
    if values.shape[0] == 0:
        return
    cuts = np.flatnonzero(values[1:] != values[:-1]) + 1
    starts = np.concatenate(([0], cuts))
    ends = np.concatenate((cuts, [values.shape[0]]))
    for a, b in zip(starts.tolist(), ends.tolist()):
        yield values[a], a, b


def load_features(f, dataset: Optional[Dataset] = None) -> List[VideoFeature]:
    with np.load(f, allow_pickle=False) as data:
        ids, feats, stamps = data["video_ids"], data["features"].astype(np.float32), data["timestamps"]
    if len(stamps) != len(feats):
        raise ValueError(f"Expected the same number of timestamps as features: got {len(stamps)} timestamps for "
                         f"{len(feats)} features")
    per_frame_scalar = stamps.ndim == 1
    if not per_frame_scalar and stamps.shape[1:] != (2,):
        raise ValueError(f"Unexpected timestamp shape. Got {stamps.shape}")
    videos = []
    for name, lo, hi in same_value_ranges(ids):
        videos.append(VideoFeature(video_id=format_video_id(name, dataset), timestamps=stamps[lo:hi], feature=feats[lo:hi]))
    return videos
