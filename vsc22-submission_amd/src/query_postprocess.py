"""Query-side descriptor post-processing (reference: infer/extract_query_feats.py:163-216,
``Main.process`` after the backbones have run).

Per query video: the per-model frame features are L2-normalised and concatenated (:176-181); if
the video-score model says "may contain a copy" (score >= SCORE_THRESHOLD), near-duplicate frames
are dropped greedily (cosine > FRAME_THRESHOLD against a kept frame, frames visited by descending
mean similarity, :197-207) and the survivors go through the fitted PCA (:210); otherwise the video
gets one tiny random descriptor (:218-228) so it can never match.  The result feeds
``query_score_normalize``.

Host bookkeeping stays numpy as in the reference; the row normalisation and the frame x frame
similarity matrix go through libvsc_hip.so when ``device_ops`` is left at its default.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from vsc.index import VideoFeature

SCORE_THRESHOLD = 0.001   # extract_query_feats.py:53
FRAME_THRESHOLD = 0.975   # :55


class HipOps:
    """normalize / similarity on the GPU (no CPU fallback: raises without a device)."""

    @staticmethod
    def normalize(x: np.ndarray) -> np.ndarray:
        import torch
        from vsc_hip import ops
        return ops.l2_normalize_(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()).cpu().numpy()

    @staticmethod
    def self_similarity(x: np.ndarray) -> np.ndarray:
        """x x^T in the fp32 fma-chain order of the search kernel (one vsc_pair_similarity_f32 launch)."""
        return HipOps.similarity(x, x)

    @staticmethod
    def similarity(a: np.ndarray, b: np.ndarray) -> np.ndarray:
        """a b^T, [len(a), len(b)]"""
        import torch
        from vsc_hip import ops
        ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        tb = ta if b is a else torch.from_numpy(np.ascontiguousarray(b, np.float32)).cuda()
        flat, _ = ops.pair_similarity(ta, tb, np.array([[0, ta.shape[0], 0, tb.shape[0]]], dtype=np.int64))
        return flat.view(ta.shape[0], tb.shape[0]).cpu().numpy()


class HipPCA:
    """``pca_model.transform`` (extract_query_feats.py:203, infer_matching.py:144) on the HIP path: built from the
    fitted sklearn ``PCA`` the reference unpickles (``mean_``, ``components_``, ``whiten``,
    ``explained_variance_``); transform(X) = (X - mean_) @ components_.T [/ sqrt(explained_variance_)]."""

    def __init__(self, fitted, ops=HipOps):
        self.mean_ = np.asarray(fitted.mean_, dtype=np.float32) if getattr(fitted, "mean_", None) is not None else None
        self.components_ = np.ascontiguousarray(fitted.components_, dtype=np.float32)
        self.scale_ = None
        if getattr(fitted, "whiten", False):
            self.scale_ = (1.0 / np.sqrt(np.asarray(fitted.explained_variance_, dtype=np.float64))).astype(np.float32)
        self.ops = ops

    def transform_device(self, x):
        """The same transform on a device tensor [n, d] -> device tensor (process_query_group: the kept rows of a whole group of
        videos in one launch; the same per-row fma chains as ``transform``)."""
        import torch
        from vsc_hip import ops
        if getattr(self, "_dev", None) is None or self._dev[0].device != x.device:
            mean = None if self.mean_ is None else torch.from_numpy(self.mean_).to(x.device)
            scale = None if self.scale_ is None else torch.from_numpy(self.scale_).to(x.device)
            self._dev = (torch.from_numpy(self.components_).to(x.device), mean, scale)
        comps, mean, scale = self._dev
        x = x.float()
        if mean is not None:
            x = x - mean
        x = x.contiguous()
        flat, _ = ops.pair_similarity(x, comps, np.array([[0, x.shape[0], 0, comps.shape[0]]], dtype=np.int64))
        out = flat.view(x.shape[0], comps.shape[0])
        return out * scale if scale is not None else out

    def transform(self, x: np.ndarray) -> np.ndarray:
        x = np.asarray(x, dtype=np.float32)
        if self.mean_ is not None:
            x = x - self.mean_
        out = self.ops.similarity(x, self.components_)
        return out * self.scale_ if self.scale_ is not None else out

    __call__ = transform


def greedy_select(sim: np.ndarray, frame_threshold: float = FRAME_THRESHOLD) -> List[int]:
    """sim: frame x frame similarity of one video with its diagonal removed -> indices kept (:200-207)."""
    removed = set()
    for i in sim.mean(0).argsort()[::-1]:
        if i in removed:
            continue
        removed.update(np.where(sim[i] > frame_threshold)[0].tolist())
    return [i for i in range(len(sim)) if i not in removed]


def select_frames(features: np.ndarray, ops=HipOps, frame_threshold: float = FRAME_THRESHOLD) -> List[int]:
    """Indices kept by the greedy near-duplicate filter (:197-207)."""
    feat = ops.normalize(features)
    sim = ops.self_similarity(feat) - np.eye(len(feat), dtype=np.float32)
    return greedy_select(sim, frame_threshold)


def process_query_video(video_id: str, sub_features: Sequence[np.ndarray], timestamps: np.ndarray, score: float,
                        pca_transform: Callable[[np.ndarray], np.ndarray], rnd_idx: int, ops=HipOps,
                        score_threshold: float = SCORE_THRESHOLD) -> Tuple[VideoFeature, List[VideoFeature], int]:
    """-> (descriptor for the video, per-model VideoFeatures, updated rnd_idx)."""
    subs = [ops.normalize(f) for f in sub_features]
    features = np.concatenate(subs, axis=1)
    ratio = len(features) // len(timestamps)
    stamps = np.asarray(list(timestamps) * ratio) if ratio != 1 else np.asarray(timestamps)
    assert len(stamps) == len(features)
    per_model = [VideoFeature(video_id=video_id, timestamps=stamps, feature=s) for s in subs]
    if score >= score_threshold:
        keep = select_frames(features, ops)
        return VideoFeature(video_id=video_id, timestamps=stamps[keep],
                            feature=pca_transform(features[keep])), per_model, rnd_idx
    rnd_idx += 1
    np.random.seed(rnd_idx)
    rnd = np.random.uniform(-1e-5, 1e-5, size=512).astype(np.float32)
    # the reference's placeholder is one [start, end] row (extract_query_feats.py:214-217); callers that pass 1-D
    # per-frame timestamps get a 1-D placeholder so that store_features can still concatenate all videos
    placeholder = np.array([0, 1])[None, ...] if stamps.ndim == 2 else np.zeros(1, dtype=stamps.dtype)
    return VideoFeature(video_id=video_id, timestamps=placeholder, feature=rnd[None, ...]), per_model, rnd_idx


def process_query_group(video_ids: Sequence[str], subs_by_model: Sequence[Sequence], timestamps: Sequence[np.ndarray], scores: Sequence[float],
                        pca_transform: Callable, rnd_idx: int, score_threshold: float = SCORE_THRESHOLD):
    """``process_query_video`` for a whole group of videos with the device work batched: subs_by_model[i][v] is model i's
    feature tensor of video v ON THE DEVICE (encode_group(as_numpy=False)).  One normalisation launch per model, one launch for
    all the videos' frame x frame similarity blocks, one for the PCA of every kept row, four device -> host copies per group --
    per video that was four normalisations, a similarity and a PCA, each a host -> device -> host round trip.
    Row-wise kernels and per-block fma chains: the results equal the per-video path bit for bit.
    -> ([descriptor per video], [per-model VideoFeatures per video], updated rnd_idx)"""
    import torch
    from vsc_hip import ops
    n_vid = len(video_ids)
    lens = [int(subs_by_model[0][v].shape[0]) for v in range(n_vid)]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    subs_dev = []
    if sum(lens) == 0:      # a group made only of frameless videos: nothing to normalise or compare; every video gets the placeholder below
        subs_dev = [torch.zeros((0, 1), device=per_video[0].device) for per_video in subs_by_model]
    for per_video in ([] if sum(lens) == 0 else subs_by_model):
        per_video = [t for t in per_video if t.shape[0] > 0]     # (an empty block may have width 0: it holds no row to concatenate)
        full = torch.cat(list(per_video)).float().contiguous() if len(per_video) > 1 else per_video[0].float().contiguous().clone()
        subs_dev.append(ops.l2_normalize_(full))
    features = torch.cat(subs_dev, dim=1).contiguous() if len(subs_dev) > 1 else subs_dev[0]
    accepted = [v for v in range(n_vid) if scores[v] >= score_threshold and lens[v] > 0]
    keep_rows, kept_per_video = [], {}
    if accepted:
        feat2 = ops.l2_normalize_(features.clone())
        pairs = np.array([[offs[v], lens[v], offs[v], lens[v]] for v in accepted], dtype=np.int64)
        flat, poff = ops.pair_similarity(feat2, feat2, pairs)
        flat = flat.cpu().numpy()
        for k, v in enumerate(accepted):
            sim = flat[poff[k]:poff[k + 1]].reshape(lens[v], lens[v]) - np.eye(lens[v], dtype=np.float32)
            keep = greedy_select(sim)
            kept_per_video[v] = keep
            keep_rows.extend(int(offs[v]) + i for i in keep)
        idx = torch.from_numpy(np.asarray(keep_rows, dtype=np.int64)).to(features.device)
        kept = features.index_select(0, idx)
        owner = getattr(pca_transform, "__self__", None)
        if isinstance(owner, HipPCA) and owner.ops is HipOps:
            reduced = owner.transform_device(kept).cpu().numpy()
        else:
            reduced = np.asarray(pca_transform(kept.cpu().numpy()))
    subs_host = [t.cpu().numpy() for t in subs_dev]
    finals, per_model, cut = [], [], 0
    for v in range(n_vid):
        ts = np.asarray(timestamps[v])
        ratio = lens[v] // len(ts) if len(ts) else 1
        stamps = np.asarray(list(ts) * ratio) if ratio != 1 else ts
        assert len(stamps) == lens[v]
        per_model.append([VideoFeature(video_id=video_ids[v], timestamps=stamps, feature=h[offs[v]:offs[v + 1]]) for h in subs_host])
        if v in kept_per_video:
            keep = kept_per_video[v]
            finals.append(VideoFeature(video_id=video_ids[v], timestamps=stamps[keep], feature=reduced[cut:cut + len(keep)]))
            cut += len(keep)
            continue
        rnd_idx += 1
        np.random.seed(rnd_idx)
        rnd = np.random.uniform(-1e-5, 1e-5, size=512).astype(np.float32)
        placeholder = np.array([0, 1])[None, ...] if stamps.ndim == 2 else np.zeros(1, dtype=stamps.dtype)
        finals.append(VideoFeature(video_id=video_ids[v], timestamps=placeholder, feature=rnd[None, ...]))
    return finals, per_model, rnd_idx
