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

    def __init__(self, clip, head, device, chunk: int = None):
        self.clip, self.head, self.device, self.chunk = clip, head, device, chunk

    def __call__(self, frames: torch.Tensor) -> float:
        return self.batch([frames])[0]

    def batch(self, frames_list: Sequence[torch.Tensor]) -> List[float]:
        """Scores of several videos: the CLIP tower runs over all their (first 256) frames as one stream of
        ``chunk``-frame calls, the MS head once per video."""
        clipped = [f[: self.head.cfg.max_frames] for f in frames_list]
        feats = encode_group([self.clip], clipped, self.device, self.chunk)[0]
        return [self.head.score(torch.from_numpy(f).to(self.device)) for f in feats]


def encode_many(model, frames_list: Sequence[torch.Tensor], device, chunk: int = None) -> List[np.ndarray]:
    """Frames of several videos through one backbone as ONE stream of ``chunk``-frame calls (a 40-frame video alone
    fills a quarter of the chip: 8 / 32 / 64 / 128-frame calls run at 14 / 46 / 74 / 83 % of the large-batch rate),
    split back per video.  The encoders are frame-independent, so this equals per-video ``encode_frames``."""
    return encode_group([model], frames_list, device, chunk)[0]


def preferred_chunk(models: Sequence, default: int = 256) -> int:
    """Frames per call for backbones that share an upload: the smallest ``preferred_batch`` among them (encoders
    expose the frame count that fills whole rounds of GEMM tiles: 332 / 451 / 255 / 256 for ViT-B/16, ViT-B/32-384,
    CLIP ViT-L/14, Swin-V2-B); ``default`` for models that do not say."""
    return min(int(getattr(m, "preferred_batch", default)) for m in models)


def encode_group(models: Sequence, frames_list: Sequence[torch.Tensor], device, chunk: int = None) -> List[List[np.ndarray]]:
    """Like ``encode_many`` for several backbones that take the SAME frames (the three Swin-V2 models of the ensemble):
    every chunk is uploaded once and goes through all of them.  -> per model, per video arrays.
    ``chunk`` = None: ``preferred_chunk(models)``."""
    chunk = chunk or preferred_chunk(models)
    lens = [f.shape[0] for f in frames_list]
    total = sum(lens)
    outs = [[] for _ in models]
    # walk the videos chunk by chunk without materialising the concatenation on the host
    buf, have = [], 0

    def flush():
        nonlocal buf, have
        if not buf:
            return
        x = torch.cat(buf).to(device, non_blocking=True) if len(buf) > 1 else buf[0].to(device, non_blocking=True)
        for i, model in enumerate(models):
            out = model(x)
            if out.dim() == 3:
                out = out[:, 0]
            outs[i].append(out.detach().float())
        buf, have = [], 0

    for f in frames_list:
        lo = 0
        while lo < f.shape[0]:
            take = min(chunk - have, f.shape[0] - lo)
            buf.append(f[lo:lo + take])
            have += take
            lo += take
            if have == chunk:
                flush()
    flush()
    cuts = np.cumsum(lens)[:-1]
    result = []
    for o in outs:
        full = torch.cat(o).cpu().numpy() if o else np.zeros((0, 0), np.float32)
        assert full.shape[0] == total
        result.append(np.split(full, cuts))
    return result


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
                     ops=HipOps, score_threshold: float = SCORE_THRESHOLD, chunk: int = None,
                     scorer: Callable[[torch.Tensor], float] = None,
                     group_frames: int = 1024) -> Tuple[List[VideoFeature], List[List[VideoFeature]]]:
    """videos yields (video_id, {image_size: frames [S,3,size,size]}, timestamps); encoders = [(model, image_size)].
    The video score comes from ``scorer(frames_by_size[VideoScorer.KEY])`` when a scorer is given (and is recorded
    in ``video_scores``), else from ``video_scores``; a video missing there is treated as accepted (score 1.0).
    Backbones run over groups of consecutive videos (>= ``group_frames`` frames) so their launches stay large.
    -> (final descriptors per video, per-model VideoFeatures per video), in input order."""
    finals, per_model = [], []
    rnd_idx = 0
    for group in _video_groups(videos, group_frames):
        # backbones that share an input size share the upload of every chunk
        subs_by_model = [None] * len(encoders)
        for size in dict.fromkeys(sz for _, sz in encoders):
            idx = [i for i, (_, sz) in enumerate(encoders) if sz == size]
            for i, per_video in zip(idx, encode_group([encoders[i][0] for i in idx], [v[1][size] for v in group], device, chunk)):
                subs_by_model[i] = per_video
        if scorer is not None:
            clip_frames = [v[1][VideoScorer.KEY] for v in group]
            scores = scorer.batch(clip_frames) if hasattr(scorer, "batch") else [scorer(f) for f in clip_frames]
            video_scores.update({v[0]: sc for v, sc in zip(group, scores)})
        for i, (video_id, frames_by_size, timestamps) in enumerate(group):
            subs = [m[i] for m in subs_by_model]
            feat, sub_feats, rnd_idx = process_query_video(video_id, subs, np.asarray(timestamps), video_scores.get(video_id, 1.0),
                                                           pca_transform, rnd_idx, ops=ops, score_threshold=score_threshold)
            finals.append(feat)
            per_model.append(sub_feats)
    return finals, per_model
