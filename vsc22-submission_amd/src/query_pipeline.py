"""Query-side extraction loop (reference: infer/extract_query_feats.py:163-254, ``Main.process`` / ``Main.run``).

For every query video: each backbone of the ensemble encodes the video's frames at its own input size, the
per-model features are L2-normalised and concatenated, near-duplicate frames are dropped, the fitted PCA maps the
concatenation to the final descriptor; videos the video-score model rejects get one tiny random descriptor.
The per-model features are kept too (the reference stores them per model, :238-245)."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

from src.query_postprocess import HipOps, SCORE_THRESHOLD, process_query_group, process_query_video
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
        probs = self.batch_device(frames_list)
        return [] if probs is None else [float(v) for v in probs.cpu().tolist()]                  # one device -> host copy per group

    def batch_device(self, frames_list: Sequence[torch.Tensor]):
        """``batch`` without the copy back: the sigmoid scores as a device tensor (None for an empty list) -- the caller decides when
        to wait for them (run_query_videos: after the NEXT group's launches are queued)."""
        clipped = [f[: self.head.cfg.max_frames] for f in frames_list]
        feats = encode_group([self.clip], clipped, self.device, self.chunk, as_numpy=False)[0]   # stay on the device: the head reads them there
        if not feats:
            return None
        # a video without frames has no score to compute: it is REJECTED (score 0 < any threshold -> the random placeholder descriptor,
        # extract_query_feats.py:218-228) instead of aborting the whole group inside the head
        have = [i for i, f in enumerate(feats) if f.shape[0] > 0]
        probs = torch.zeros(len(feats), device=self.device)
        if have:
            probs[torch.tensor(have, device=self.device)] = torch.sigmoid(self.head.logits([feats[i] for i in have])).reshape(-1).float()
        return probs


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


def preferred_call(models: Sequence, frame_bytes: int, default: int = 256, cap_bytes: int = 512 << 20) -> int:
    """Frames per staged chunk = per encoder call: the encoders' ``preferred_call`` (a chunk for each of their lanes: a call of ONE chunk
    runs on one lane alone, 5 % below what the encoder does with both) while a staging buffer of it stays within ``cap_bytes``."""
    one = preferred_chunk(models, default)
    many = min(int(getattr(m, "preferred_call", getattr(m, "preferred_batch", default))) for m in models)
    return many if many * frame_bytes <= cap_bytes else one


class _Stager:
    """Host -> device staging of frame chunks: two pinned host buffers and two device buffers of one chunk each, and a copy
    stream.  The host gathers chunk k + 1 from the videos' (pageable) tensors straight into a pinned buffer while chunk k is
    encoded, and the copy of chunk k + 1 runs on the copy stream under chunk k's kernels -- `torch.cat(buf).to(device)` from
    pageable memory did both on the critical path (a synchronous copy at pageable speed: as long as the encoders' own time
    at 0.8 MB of uint8 frames per query frame through the reference's ensemble)."""

    _cache = {}

    @classmethod
    def get(cls, device, chunk, shape, dtype):
        key = (str(device), int(chunk), tuple(shape), dtype)
        st = cls._cache.get(key)
        if st is None:
            if len(cls._cache) >= 8:      # a few (input size, dtype) combinations per process; do not grow without bound
                cls._cache.pop(next(iter(cls._cache)))
            st = cls._cache[key] = cls(device, chunk, shape, dtype)
        return st

    def __init__(self, device, chunk, shape, dtype):
        self.pinned = [torch.empty((chunk,) + tuple(shape), dtype=dtype).pin_memory() for _ in range(2)]
        self.dev = [torch.empty((chunk,) + tuple(shape), dtype=dtype, device=device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.copied = [torch.cuda.Event() for _ in range(2)]     # the H2D copy out of pinned[slot] / into dev[slot] is done
        self.consumed = [torch.cuda.Event() for _ in range(2)]   # every encoder is done reading dev[slot]
        self.used = [False, False]

    _pool = None

    def upload(self, slot, pieces):
        """pieces: [(tensor, lo, take)] -> device view of the chunk, ordered behind its copy on the caller's stream"""
        if self.used[slot]:
            self.copied[slot].synchronize()       # (issued two chunks ago: long done) the pinned buffer may be refilled
        # The gather into pinned memory is a plain memcpy of up to a few hundred MB per chunk; one thread moves ~2.9 GB/s here -- 19 k uint8
        # 224 x 224 frames/s, below the ViT-B/16 encoder's 25 k -- so large pieces are cut across four threads (torch releases the GIL in copy_).
        off, jobs = 0, []
        for f, lo, take in pieces:
            nbytes = take * f[0].numel() * f.element_size() if take else 0
            parts = 4 if nbytes >= (8 << 20) and take >= 4 else 1
            step = -(-take // parts)
            for a in range(0, take, step):
                b = min(take, a + step)
                jobs.append((self.pinned[slot][off + a:off + b], f[lo + a:lo + b]))
            off += take
        if len(jobs) > 1 and any(d.numel() * d.element_size() >= (2 << 20) for d, _ in jobs):
            if _Stager._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                _Stager._pool = ThreadPoolExecutor(4, thread_name_prefix="vsc-stage")
            list(_Stager._pool.map(lambda j: j[0].copy_(j[1]), jobs))
        else:
            for d, src in jobs:
                d.copy_(src)
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.copy_stream):
            if self.used[slot]:
                self.copy_stream.wait_event(self.consumed[slot])
            self.dev[slot][:off].copy_(self.pinned[slot][:off], non_blocking=True)
            self.copied[slot].record(self.copy_stream)
        cur.wait_event(self.copied[slot])
        self.used[slot] = True
        return self.dev[slot][:off]

    def done(self, slot):
        self.consumed[slot].record(torch.cuda.current_stream())


def encode_group(models: Sequence, frames_list: Sequence[torch.Tensor], device, chunk: int = None, as_numpy: bool = True):
    """Like ``encode_many`` for several backbones that take the SAME frames (the three Swin-V2 models of the ensemble):
    every chunk is uploaded once and goes through all of them.  -> per model, per video arrays (``as_numpy=False``: device
    tensors -- the video-score head takes them where they are).  ``chunk`` = None: ``preferred_chunk(models)``.
    On a GPU the chunks go through pinned staging buffers and a copy stream (``_Stager``); the per-model outputs of the whole
    group come back in ONE device -> host copy each."""
    if not chunk:
        f0 = frames_list[0] if len(frames_list) else None
        chunk = preferred_call(models, int(f0[0].numel() * f0.element_size()) if f0 is not None and f0.shape[0] else 1)
    lens = [f.shape[0] for f in frames_list]
    total = sum(lens)
    outs = [[] for _ in models]
    staged = torch.device(device).type == "cuda" and total > 0
    st = _Stager.get(device, chunk, frames_list[0].shape[1:], frames_list[0].dtype) if staged else None
    # walk the videos chunk by chunk without materialising the concatenation on the host
    buf, have, k = [], 0, 0

    def flush():
        nonlocal buf, have, k
        if not buf:
            return
        if staged:
            x = st.upload(k & 1, buf)
        else:   # host-logic tests (fake encoders on the CPU); the HIP encoders refuse CPU tensors
            x = torch.cat([f[lo:lo + take] for f, lo, take in buf]).to(device)
        for i, model in enumerate(models):
            out = model(x)
            if out.dim() == 3:
                out = out[:, 0]
            outs[i].append(out.detach().float())
        if staged:
            st.done(k & 1)
        buf, have, k = [], 0, k + 1

    for f in frames_list:
        lo = 0
        while lo < f.shape[0]:
            take = min(chunk - have, f.shape[0] - lo)
            buf.append((f, lo, take))
            have += take
            lo += take
            if have == chunk:
                flush()
    flush()
    cuts = np.cumsum(lens)[:-1]
    result = []
    for o in outs:
        if not o:      # a group without a single frame: one empty block per video, in either form
            result.append(np.split(np.zeros((0, 0), np.float32), cuts) if as_numpy else [torch.empty((0, 0), device=device) for _ in lens])
            continue
        full = torch.cat(o) if len(o) > 1 else o[0]
        assert full.shape[0] == total
        result.append(np.split(full.cpu().numpy(), cuts) if as_numpy else list(torch.split(full, lens)))
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
                     group_frames: int = 4096) -> Tuple[List[VideoFeature], List[List[VideoFeature]]]:
    """videos yields (video_id, {image_size: frames [S,3,size,size]}, timestamps); encoders = [(model, image_size)].
    The video score comes from ``scorer(frames_by_size[VideoScorer.KEY])`` when a scorer is given (and is recorded
    in ``video_scores``), else from ``video_scores``; a video missing there is treated as accepted (score 1.0).
    Backbones run over groups of consecutive videos (>= ``group_frames`` frames) so their launches stay large and the ragged last chunk of a
    group (16 / 138 / 20 frames at 512 / 902 / 510 per call) is paid once per 4 096 frames: 208 videos x 40 frames end to end at 0.904 / 0.966 /
    0.978 / 0.954 of encoder-bound with groups of 1 024 / 2 048 / 4 096 / 8 192 frames (the last: one group, no look-ahead; tools/micro/ensemble_group_scan.sh).
    -> (final descriptors per video, per-model VideoFeatures per video), in input order."""
    finals, per_model = [], []
    rnd_idx = 0
    # on the GPU with the library's own ops the per-video post-processing is batched per group and the features stay on the device
    # in between (process_query_group); anything else (the host-logic tests' numpy ops) takes the reference's per-video steps
    batched = ops is HipOps and torch.device(device).type == "cuda"
    if batched:
        # One group of look-ahead: group g + 1's uploads and launches are queued BEFORE the host waits for group g's results (the video
        # scores' copy back, the similarity blocks, the kept rows) -- the chip used to idle through every group's post-processing and the
        # next group's first gather.  Same calls in the same order per group: same results.
        def launch(group):
            subs_by_model = [None] * len(encoders)
            for size in dict.fromkeys(sz for _, sz in encoders):
                idx = [i for i, (_, sz) in enumerate(encoders) if sz == size]
                for i, per_video in zip(idx, encode_group([encoders[i][0] for i in idx], [v[1][size] for v in group], device, chunk, as_numpy=False)):
                    subs_by_model[i] = per_video
            probs = None
            if scorer is not None:
                clip_frames = [v[1][VideoScorer.KEY] for v in group]
                probs = scorer.batch_device(clip_frames) if hasattr(scorer, "batch_device") else \
                    (scorer.batch(clip_frames) if hasattr(scorer, "batch") else [scorer(f) for f in clip_frames])
            return group, subs_by_model, probs

        def finish(group, subs_by_model, probs):
            nonlocal rnd_idx
            if probs is not None:
                vals = probs.cpu().tolist() if torch.is_tensor(probs) else list(probs)
                video_scores.update({v[0]: float(sc) for v, sc in zip(group, vals)})
            f, pm, rnd_idx = process_query_group([v[0] for v in group], subs_by_model, [np.asarray(v[2]) for v in group],
                                                 [video_scores.get(v[0], 1.0) for v in group], pca_transform, rnd_idx, score_threshold)
            finals.extend(f)
            per_model.extend(pm)

        pending = None
        for group in _video_groups(videos, group_frames):
            cur = launch(group)
            if pending is not None:
                finish(*pending)
            pending = cur
        if pending is not None:
            finish(*pending)
        return finals, per_model
    # anything else (the host-logic tests' numpy ops): the reference's per-video steps
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
