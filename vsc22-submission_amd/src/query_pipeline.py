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


def run_query_videos(videos: Iterable[Tuple[str, Dict[int, torch.Tensor], np.ndarray]], encoders: Sequence[Tuple[object, int]],
                     pca_transform: Callable[[np.ndarray], np.ndarray], video_scores: Dict[str, float], device,
                     ops=HipOps, score_threshold: float = SCORE_THRESHOLD, chunk: int = 256,
                     scorer: Callable[[torch.Tensor], float] = None) -> Tuple[List[VideoFeature], List[List[VideoFeature]]]:
    """videos yields (video_id, {image_size: frames [S,3,size,size]}, timestamps); encoders = [(model, image_size)].
    The video score comes from ``scorer(frames_by_size[VideoScorer.KEY])`` when a scorer is given (and is recorded
    in ``video_scores``), else from ``video_scores``; a video missing there is treated as accepted (score 1.0).
    -> (final descriptors per video, per-model VideoFeatures per video)."""
    finals, per_model = [], []
    rnd_idx = 0
    for video_id, frames_by_size, timestamps in videos:
        subs = [encode_frames(model, frames_by_size[size], device, chunk) for model, size in encoders]
        if scorer is not None:
            video_scores[video_id] = scorer(frames_by_size[VideoScorer.KEY])
        feat, sub_feats, rnd_idx = process_query_video(video_id, subs, np.asarray(timestamps), video_scores.get(video_id, 1.0),
                                                       pca_transform, rnd_idx, ops=ops, score_threshold=score_threshold)
        finals.append(feat)
        per_model.append(sub_feats)
    return finals, per_model
