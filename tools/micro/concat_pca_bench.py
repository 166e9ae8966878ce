"""Reference-side merge (concat_pca_sn.merge_set: normalise per model, concatenate, PCA 2048 -> 512) of 4 000 videos x 25 frames x 4 models:
one video per device round trip (block_rows = 1: the per-video form) against blocks of 2^18 frames.  Run on the GPU box."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import numpy as np
import concat_pca_sn as C
from src.query_postprocess import HipPCA
from vsc.index import VideoFeature
from vsc.storage import store_features

rng = np.random.default_rng(0)
nv, nf, d = 4000, 25, 512


class Fitted:
    mean_ = rng.standard_normal(4 * d).astype(np.float32) * 0.01
    components_ = (rng.standard_normal((512, 4 * d)) / 45.0).astype(np.float32)
    whiten = False


with tempfile.TemporaryDirectory() as tmp:
    paths = []
    for m in range(4):
        feats = rng.standard_normal((nv * nf, d), dtype=np.float32)
        p = os.path.join(tmp, f"m{m}.npz")
        store_features(p, [VideoFeature(video_id=f"R{200000 + v}", timestamps=np.arange(nf, dtype=np.float64), feature=feats[v * nf:(v + 1) * nf]) for v in range(nv)])
        paths.append(p)
    pca = HipPCA(Fitted)
    C.merge_set(paths, pca.transform, block_rows=1 << 12)      # warm (library, scratch)
    for name, rows in (("one video per round trip", 1), ("blocks of 2^18 frames", None)):
        t0 = time.perf_counter()
        out = C.merge_set(paths, pca.transform, block_rows=rows)
        dt = time.perf_counter() - t0
        print(f"{name}: {nv} videos x {nf} frames x 4 models in {dt:.2f} s (incl. reading the four .npz files)", flush=True)
        ref = out if rows == 1 else ref
    assert all(np.array_equal(a.feature, b.feature) for a, b in zip(out, ref))
