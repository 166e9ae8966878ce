"""A deterministic miniature of the descriptor track's data: reference videos, an independent score-normalisation set, and query
videos some of which hold EDITED COPIES of reference frames at graded difficulty -- the input of the end-to-end uAP parity test
(tests/golden/gen_uap_e2e_golden.py makes the fp32 fixture through the reference's own model classes, tests/test_gpu_uap_e2e.py
runs the same bytes through the HIP entry points).

Frames are uint8 [H, W, 3] (what a decoder hands to `vit_transform`, infer/src/transform.py:37-42).  Everything is integer hashing
(`tools.synth.uniform`), +, *, abs, floor, min / max and index gathers in float64: the same bytes on every machine; `fingerprint`
(sha256 over all frames) is stored in the fixture and re-checked where the frames are regenerated.

The edits are the kind the challenge applies, in closed form: zoom into a crop (nearest-neighbour index map), box blur, contrast /
change, additive noise, a blend with a foreign frame.  `level` in (0, 1] scales all of them; easy copies stay the nearest
neighbours of their sources, hard ones sink below unrelated pairs -- which is what makes the uAP land away from 1.0."""
from __future__ import annotations

import hashlib
from typing import Dict, List, Tuple

import numpy as np

from tools import synth

BASE = 256          # frames are drawn at this size; the extractors resize per model (PIL bicubic), as the reference does
FRAMES = 4          # per video
N_REF, N_NORM, N_QUERY, N_POSITIVE = 80, 60, 64, 44


def scenes(seed: int, n: int, size: int = BASE) -> np.ndarray:
    """[n, 3, size, size] float64 in [-1, 1): three oriented triangle waves, four soft blobs and a rectangle per frame, each with its own
    per-channel weights (96 parameters per frame from `tools.synth.uniform`), plus a little pixel noise.  (`synth.structured_frames` is a
    16-parameter family: among a thousand of its frames some pairs are near-identical, which a retrieval fixture cannot use.)"""
    par = synth.uniform(seed * 7919 + 21, (n, 96)).astype(np.float64)
    noise = synth.uniform(seed * 7919 + 22, (n, 3, size, size)).astype(np.float64)
    g = (np.arange(size, dtype=np.float64) + 0.5) / size
    y, x = g[:, None], g[None, :]
    out = np.empty((n, 3, size, size), np.float64)
    for i in range(n):
        p, k = par[i], 0
        img = np.zeros((3, size, size), np.float64)
        for _ in range(3):                                           # waves: 0.5 .. 8 cycles per image, either orientation
            kx = (0.5 + 3.75 * (p[k] + 1.0)) * (1.0 if p[k + 1] >= 0 else -1.0)
            ky = (0.5 + 3.75 * (p[k + 2] + 1.0)) * (1.0 if p[k + 3] >= 0 else -1.0)
            amp = 0.1 + 0.2 * (p[k + 4] + 1.0)
            for c in range(3):
                t = kx * x + ky * y + 0.5 * p[k + 5 + c]
                img[c] += amp * (0.6 + 0.4 * p[k + 8 + c]) * (4.0 * np.abs(t - np.floor(t + 0.5)) - 1.0)
            k += 11
        for _ in range(4):                                           # blobs
            cx, cy, r2 = 0.5 + 0.45 * p[k], 0.5 + 0.45 * p[k + 1], 0.005 + 0.04 * (p[k + 2] + 1.0)
            blob = np.maximum(0.0, 1.0 - ((x - cx) ** 2 + (y - cy) ** 2) / r2) ** 2
            for c in range(3):
                img[c] += 0.9 * p[k + 3 + c] * blob
            k += 6
        x0, y0 = 0.4 * (p[k] + 1.0), 0.4 * (p[k + 1] + 1.0)          # a rectangle
        w, h = 0.15 + 0.2 * (p[k + 2] + 1.0), 0.15 + 0.2 * (p[k + 3] + 1.0)
        rect = ((x >= x0) & (x < x0 + w) & (y >= y0) & (y < y0 + h)).astype(np.float64)
        for c in range(3):
            img[c] += 0.6 * p[k + 4 + c] * rect + 0.3 * p[k + 7 + c]
        out[i] = img + 0.04 * noise[i]
    return np.minimum(np.maximum(out, -1.0), 0.999)


def _zoom(x: np.ndarray, frac: float, ox: float, oy: float) -> np.ndarray:
    s = x.shape[-1]
    win = max(8, int(frac * s))
    x0, y0 = int(ox * (s - win)), int(oy * (s - win))
    idx = (np.arange(s) * win) // s
    return x[:, (y0 + idx)[:, None], (x0 + idx)[None, :]]


def _box_blur(x: np.ndarray, k: int) -> np.ndarray:
    if k <= 1:
        return x
    s, r = x.shape[-1], k // 2
    acc = np.zeros_like(x)
    for dy in range(-r, r + 1):                       # fixed order of explicit adds: reproducible
        iy = np.clip(np.arange(s) + dy, 0, s - 1)
        for dx in range(-r, r + 1):
            ix = np.clip(np.arange(s) + dx, 0, s - 1)
            acc = acc + x[:, iy[:, None], ix[None, :]]
    return acc * (1.0 / (k * k))


def edit(x: np.ndarray, foreign: np.ndarray, level: float, par: np.ndarray, noise: np.ndarray) -> np.ndarray:
    """One edited copy of frame x ([3, S, S] float64); par = numbers in [-1, 1) choosing the crop position."""
    y = _zoom(x, 1.0 - 0.25 * level, 0.5 * (par[0] + 1.0), 0.5 * (par[1] + 1.0))
    y = _box_blur(y, 1 + 2 * int(round(1.5 * level)))
    y = (1.0 - 0.3 * level) * y                                   # (no brightness offset: a random-weight network's descriptor follows the mean grey level more than the content)
    y = (1.0 - 0.3 * level) * y + 0.3 * level * foreign
    y = y + 0.2 * level * noise
    return np.minimum(np.maximum(y, -1.0), 0.999)


def to_u8(x: np.ndarray) -> np.ndarray:
    """[..., 3, S, S] in [-1, 1) -> uint8 [..., S, S, 3]"""
    q = np.floor((x + 1.0) * 127.5 + 0.5)
    return np.moveaxis(np.clip(q, 0, 255).astype(np.uint8), -3, -1)


def make(seed: int = 2022, n_ref: int = None, n_norm: int = None, n_query: int = None, n_positive: int = None) -> Dict:
    """-> {"refs": [(id, u8 [F,S,S,3])], "norm": [...], "queries": [...], "gt": [(query_id, ref_id)], "levels": {query_id: level},
    "fingerprint": hex}.  Ids follow the challenge (Q2xxxxx queries, R2xxxxx references, R1xxxxx the other split's references, which
    the reference normalises scores against: extract_query_feats.py:47-50)."""
    n_ref, n_norm, n_query, n_positive = (n_ref or N_REF), (n_norm or N_NORM), (n_query or N_QUERY), (n_positive or N_POSITIVE)
    assert n_positive <= n_query and n_positive <= n_ref and n_ref % 7 != 0      # (the copy walk q -> 7 q + 3 mod n_ref visits every reference once)
    ref_scenes = scenes(seed * 10 + 1, n_ref * FRAMES).reshape(n_ref, FRAMES, 3, BASE, BASE)
    norm_scenes = scenes(seed * 10 + 2, n_norm * FRAMES).reshape(n_norm, FRAMES, 3, BASE, BASE)
    q_scenes = scenes(seed * 10 + 3, n_query * FRAMES).reshape(n_query, FRAMES, 3, BASE, BASE)
    par = synth.uniform(seed * 10 + 4, (n_query, FRAMES, 6)).astype(np.float64)
    refs = [(f"R2{i:05d}", to_u8(ref_scenes[i])) for i in range(n_ref)]
    norm = [(f"R1{i:05d}", to_u8(norm_scenes[i])) for i in range(n_norm)]
    queries, gt, levels = [], [], {}
    for q in range(n_query):
        qid = f"Q2{q:05d}"
        frames = q_scenes[q].copy()
        if q < n_positive:
            src = (q * 7 + 3) % n_ref                                   # which reference it copies (a permutation walk: 7 and n_ref are coprime)
            level = 0.03 + 0.42 * (q / (n_positive - 1))                # 0.03 (nearly verbatim) .. 0.45 (zoom to 89 %, 3 x 3 blur, 13 % foreign frame, noise)
            for f in (1, 2):                                            # two of the four frames are copies of reference frames f, f + 1
                noise = synth.uniform(seed * 1000 + q * 8 + f, (3, BASE, BASE)).astype(np.float64)
                frames[f] = edit(ref_scenes[src, f], q_scenes[q, f], level, par[q, f], noise)
            gt.append((qid, f"R2{src:05d}"))
            levels[qid] = level
        if q % 5 == 4:
            frames[3] = frames[2]                                       # an exact duplicate frame: the near-duplicate filter must drop one
        queries.append((qid, to_u8(frames)))
    h = hashlib.sha256()
    for group in (refs, norm, queries):
        for _, f in group:
            h.update(f.tobytes())
    return {"refs": refs, "norm": norm, "queries": queries, "gt": gt, "levels": levels, "fingerprint": h.hexdigest()}


def resize_u8(frames: np.ndarray, size: int) -> np.ndarray:
    """uint8 [n, H, W, 3] -> [n, size, size, 3] by PIL's bicubic resize: `Resize([size, size], BICUBIC)` of the reference's transforms
    (extract_query_feats.py:106-129) on a PIL image; identity when the frame already has that size."""
    from PIL import Image
    if frames.shape[1] == size and frames.shape[2] == size:
        return frames
    return np.stack([np.asarray(Image.fromarray(f).resize((size, size), Image.BICUBIC), dtype=np.uint8) for f in frames])


def write_zips(videos: List[Tuple[str, np.ndarray]], prefix: str) -> None:
    """<prefix>/<id[-2:]>/<id>.zip of losslessly stored frames (PNG), the layout `ZipFrames` / `QueryVideos` read
    (infer/src/dataset.py:118-131 reads jpgs the same way; PNG keeps the decoded bytes identical on both sides of the comparison)."""
    import io
    import os
    from zipfile import ZipFile
    from PIL import Image
    for vid, frames in videos:
        os.makedirs(os.path.join(prefix, vid[-2:]), exist_ok=True)
        with ZipFile(os.path.join(prefix, vid[-2:], vid + ".zip"), "w") as z:
            for i, f in enumerate(frames):
                buf = io.BytesIO()
                Image.fromarray(f).save(buf, format="PNG")
                z.writestr(f"{i:05d}.png", buf.getvalue())
