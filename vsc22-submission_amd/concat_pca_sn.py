"""Reference-side merge of the ensemble's descriptors (reference: infer/concat_pca_sn.py).

    python concat_pca_sn.py --root outputs --models swinv2_v115 swinv2_v107 swinv2_v106 vit_v68 \
        --pca_model ../checkpoints/pca_model.pkl [--fit_pca]

Per feature set (train_refs, test_refs): the per-model descriptors of every video are L2-normalised, concatenated and
mapped to 512-d by the PCA (concat_pca_sn.py:56-68) -> <root>/<set>.npz; then each set is score-normalised against the
other (:71-88) -> <root>/<set>_sn.npz.  Row normalisation and the PCA product run on the GPU (HipOps / HipPCA);
``--fit_pca`` fits the PCA on the train set with sklearn exactly as the reference does (:42-54, offline, CPU) and
pickles it, otherwise the pickle is loaded."""
from __future__ import annotations

import argparse
import os
import pickle

import numpy as np

from src.query_postprocess import HipOps, HipPCA
from vsc.baseline.score_normalization import ref_score_normalize
from vsc.index import VideoFeature
from vsc.storage import load_features, store_features

NK, BETA = 1, 1.2   # concat_pca_sn.py:70-71


def concat_models(per_model_features, ops=HipOps):
    """[{video_id: VideoFeature} per model] -> (video ids in the first model's order, [n_frames, sum dims] per video)."""
    vids = list(per_model_features[0].keys())
    return vids, [np.concatenate([ops.normalize(m[v].feature) for m in per_model_features], axis=1) for v in vids]


BLOCK_ROWS = 1 << 18   # frames per device round trip of merge_set (x 2048 floats of concatenated descriptors = 2 GiB)


def merge_set(paths, pca_transform, ops=HipOps, block_rows: int = None):
    """Per video: normalise every model's rows, concatenate, PCA (concat_pca_sn.py:56-68).  The reference does this video by video on the
    host; video by video through the GPU it was one host -> device -> host round trip per video and model plus one for the PCA -- 200 k of
    them for the track's 40 k reference videos.  Row normalisation and the PCA product are row-wise, so the videos of a BLOCK go through
    them together (one round trip per model and block, one for the PCA) and are cut apart again: the same per-row arithmetic, the same
    bits (tests/test_gpu_knn.py::test_concat_pca_sn_entry_point compares the two forms)."""
    models = [{vf.video_id: vf for vf in load_features(p)} for p in paths]
    vids = list(models[0].keys())
    lens = [len(models[0][v].feature) for v in vids]
    out, lo = [], 0
    limit = block_rows or BLOCK_ROWS
    while lo < len(vids):
        hi, rows = lo, 0
        while hi < len(vids) and (hi == lo or rows + lens[hi] <= limit):
            rows += lens[hi]
            hi += 1
        block = vids[lo:hi]
        cat = np.concatenate([ops.normalize(np.concatenate([m[v].feature for v in block])) for m in models], axis=1)
        reduced = np.asarray(pca_transform(cat))
        cuts = np.cumsum([lens[i] for i in range(lo, hi)])[:-1]
        out.extend(VideoFeature(video_id=v, feature=f, timestamps=models[0][v].timestamps) for v, f in zip(block, np.split(reduced, cuts)))
        lo = hi
    return out


def main(args):
    sets = ["train_refs", "test_refs"]
    path = lambda model, name: os.path.join(args.root, model, f"{name}.npz")
    if args.fit_pca:
        from sklearn.decomposition import PCA
        models = [{vf.video_id: vf for vf in load_features(path(m, sets[0]))} for m in args.models]
        fitted = PCA(n_components=args.dim, random_state=2023).fit(np.concatenate(concat_models(models)[1]))
        with open(args.pca_model, "wb") as f:
            pickle.dump(fitted, f)
    else:
        with open(args.pca_model, "rb") as f:
            fitted = pickle.load(f)
    pca = HipPCA(fitted)
    for name in sets:
        store_features(os.path.join(args.root, f"{name}.npz"), merge_set([path(m, name) for m in args.models], pca.transform))
    for name, other in ((sets[1], sets[0]), (sets[0], sets[1])):
        refs = load_features(os.path.join(args.root, f"{name}.npz"))
        norm = load_features(os.path.join(args.root, f"{other}.npz"))
        store_features(os.path.join(args.root, f"{name}_sn.npz"), ref_score_normalize(refs, norm, nk=NK, beta=BETA))


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--root", default="./outputs")
    ap.add_argument("--models", nargs="+", default=["swinv2_v115", "swinv2_v107", "swinv2_v106", "vit_v68"])
    ap.add_argument("--pca_model", default="../checkpoints/pca_model.pkl")
    ap.add_argument("--fit_pca", action="store_true")
    ap.add_argument("--dim", type=int, default=512)
    return ap


if __name__ == "__main__":
    main(build_parser().parse_args())
