"""Query-side extraction loop (reference: infer/extract_query_feats.py:163-254, ``Main.process`` / ``Main.run``).

For every query video: each backbone of the ensemble encodes the video's frames at its own input size, the
per-model features are L2-normalised and concatenated, near-duplicate frames are dropped, the fitted PCA maps the
concatenation to the final descriptor; videos the video-score model rejects get one tiny random descriptor.
The per-model features are kept too (the reference stores them per model, :238-245)."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

from src.query_postprocess import HipOps, SCORE_THRESHOLD, process_query_video
from vsc.index import VideoFeature


def encode_frames(model, frames: torch.Tensor, device, chunk: int = 256) -> np.ndarray:
    """``single_infer`` (:150-161): frames of one video through one backbone, ``chunk`` at a time."""
    outs = []
    for lo in range(0, frames.shape[0], chunk):
        out = model(frames[lo:lo + chunk].to(device))
        if out.dim() == 3:
            out = out[:, 0]
        outs.append(out.detach().float().cpu().numpy())
    return np.concatenate(outs, axis=0)


class VideoScorer:
    """The video-score gate (extract_query_feats.py:163-174): CLIP [CLS] features of the first 256 frames ->
    ``MS`` head -> sigmoid.  ``clip`` is a HipEncoder with pool="cls" (clip_vit_l14_224), ``head`` a VideoScoreHead."""

    KEY = "clip"   # frames_by_size key holding the CLIP-normalised frames

    def __init__(self, clip, head, device, chunk: int = 256):
        self.clip, self.head, self.device, self.chunk = clip, head, device, chunk

    def __call__(self, frames: torch.Tensor) -> float:
        n = min(frames.shape[0], self.head.cfg.max_frames)
        feats = [self.clip(frames[lo:min(lo + self.chunk, n)].to(self.device)) for lo in range(0, n, self.chunk)]
        return self.head.score(torch.cat(feats))


def encode_many(model, frames_list: Sequence[torch.Tensor], device, chunk: int = 256) -> List[np.ndarray]:
    """Frames of several videos through one backbone as ONE stream of ``chunk``-frame calls (a 40-frame video alone
    fills a quarter of the chip: 8 / 32 / 64 / 128-frame calls run at 14 / 46 / 74 / 83 % of the large-batch rate),
    split back per video.  The encoders are frame-independent, so this equals per-video ``encode_frames``."""
    lens = [f.shape[0] for f in frames_list]
    out = encode_frames(model, torch.cat(list(frames_list)), device, chunk)
    cuts = np.cumsum(lens)[:-1]
    return np.split(out, cuts)


def _video_groups(videos, min_frames: int):
    """Consecutive videos grouped until a group holds at least ``min_frames`` frames."""
    group, n = [], 0
    for v in videos:
        group.append(v)
        n += len(v[2])
        if n >= min_frames:
            yield group
            group, n = [], 0
    if group:
        yield group


def run_query_videos(videos: Iterable[Tuple[str, Dict[int, torch.Tensor], np.ndarray]], encoders: Sequence[Tuple[object, int]],
                     pca_transform: Callable[[np.ndarray], np.ndarray], video_scores: Dict[str, float], device,
                     ops=HipOps, score_threshold: float = SCORE_THRESHOLD, chunk: int = 256,
                     scorer: Callable[[torch.Tensor], float] = None,
                     group_frames: int = 512) -> Tuple[List[VideoFeature], List[List[VideoFeature]]]:
    """videos yields (video_id, {image_size: frames [S,3,size,size]}, timestamps); encoders = [(model, image_size)].
    The video score comes from ``scorer(frames_by_size[VideoScorer.KEY])`` when a scorer is given (and is recorded
    in ``video_scores``), else from ``video_scores``; a video missing there is treated as accepted (score 1.0).
    Backbones run over groups of consecutive videos (>= ``group_frames`` frames) so their launches stay large.
    -> (final descriptors per video, per-model VideoFeatures per video), in input order."""
    finals, per_model = [], []
    rnd_idx = 0
    for group in _video_groups(videos, group_frames):
        subs_by_model = [encode_many(model, [v[1][size] for v in group], device, chunk) for model, size in encoders]
        for i, (video_id, frames_by_size, timestamps) in enumerate(group):
            subs = [m[i] for m in subs_by_model]
            if scorer is not None:
                video_scores[video_id] = scorer(frames_by_size[VideoScorer.KEY])
            feat, sub_feats, rnd_idx = process_query_video(video_id, subs, np.asarray(timestamps), video_scores.get(video_id, 1.0),
                                                           pca_transform, rnd_idx, ops=ops, score_threshold=score_threshold)
            finals.append(feat)
            per_model.append(sub_feats)
    return finals, per_model
