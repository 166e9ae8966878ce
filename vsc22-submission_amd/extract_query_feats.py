"""Query-descriptor extraction with the backbone ensemble (reference: infer/extract_query_feats.py).

    python extract_query_feats.py --split test \
        --models swinv2_base_256:swin_ref:ckpt/swinv2_v115.pth swinv2_base_256:swin_ref:ckpt/swinv2_v107.pth \
                 swinv2_base_256:swin_ref:ckpt/swinv2_v106.pth vit_v68:timm_vit:ckpt/vit_v68.pth \
        --pca_model ckpt/pca_model.pkl --zip_prefix ../data/jpg_zips --input_file ../data/meta/test/query_ids.txt \
        --norm_refs outputs/train_refs.npz --output_dir outputs [--video_scores video_scores.csv]

Outputs, as the reference: ``<output_dir>/<model name>/<split>_query.npz`` per backbone (:238-245) and
``<output_dir>/<split>_query_sn.npz`` after query score normalisation (:247-252).  The video-score gate
(CLIP ViT-L/14 [CLS] features -> ``MS`` head, :163-174) runs on the HIP path when ``--clip_checkpoint`` and
``--vsm_checkpoint`` (the state dicts torch2scripts.py traces) are given; otherwise scores are read from
``--video_scores`` (csv: video_id,score) and every video passes when neither is given."""
from __future__ import annotations

import argparse
import csv
import io
import os
import pickle
from zipfile import ZipFile

import numpy as np
import torch

from src.dataset import CLIP_MEAN, CLIP_STD, clip_transform_u8, vit_transform_u8
from src.matching import calclualte_low_var_dim
from src.model_zoo import DEFAULT_PRECISION, load_encoder, parse_model_spec
from src.query_pipeline import VideoScorer, run_query_videos
from src.query_postprocess import HipPCA, SCORE_THRESHOLD
from vsc.baseline.score_normalization import query_score_normalize
from vsc.metrics import Dataset
from vsc.storage import load_features, store_features

NK, BETA = 1, 1.2  # extract_query_feats.py:56-57


class QueryVideos(torch.utils.data.Dataset):
    """One item = (video_id, {size: uint8 frames [S,size,size,3]}, timestamps): the jpgs of
    <prefix>/<vid[-2:]>/<vid>.zip decoded once and resized per input size (bicubic, as vit_transform / the CLIP
    transform do).  Frames stay uint8 until they are on the GPU: ToTensor + Normalize run inside the encoders'
    patchify kernels.  Decoding is the slow part of the whole pipeline, so main() reads this through a DataLoader with
    worker processes, as the reference does (extract_query_feats.py:138-140)."""

    def __init__(self, video_ids, zip_prefix, sizes, with_clip=False):
        self.zip_prefix = zip_prefix
        self.video_ids = [v for v in video_ids if os.path.exists(self._path(v))]
        self.transforms = {s: vit_transform_u8(s, s) for s in sizes}
        if with_clip:
            self.transforms[VideoScorer.KEY] = clip_transform_u8(224)

    def _path(self, vid):
        return "%s/%s/%s.zip" % (self.zip_prefix, vid[-2:], vid)

    def __len__(self):
        return len(self.video_ids)

    def __getitem__(self, i):
        from PIL import Image
        vid = self.video_ids[i]
        with ZipFile(self._path(vid), "r") as z:
            images = [Image.open(io.BytesIO(z.read(n))).convert("RGB") for n in sorted(z.namelist())]
        # [start, end] seconds per frame at 1 fps, the layout the reference's FFmpeg reader yields (infer/src/dataset.py:97-102):
        # a video rejected by the score gate gets a [1, 2] placeholder, so every video's timestamps must be 2-D for
        # store_features to concatenate them
        n = len(images)
        stamps = np.stack([np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32) + 1.0], axis=1)
        return vid, {s: torch.stack([t(im) for im in images]) for s, t in self.transforms.items()}, stamps


def zip_videos(video_ids, zip_prefix, sizes, with_clip=False, workers=0):
    data = QueryVideos(video_ids, zip_prefix, sizes, with_clip)
    kw = {"prefetch_factor": 4} if workers > 0 else {}
    return torch.utils.data.DataLoader(data, batch_size=1, shuffle=False, num_workers=workers, collate_fn=lambda b: b[0], **kw)


def read_video_scores(path):
    if not path:
        return {}
    with open(path, newline="", encoding="utf-8") as f:
        rows = [r for r in csv.reader(f) if r and r[0] != "video_id"]
    return {r[0]: float(r[1]) for r in rows}


def main(args):
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    specs = [parse_model_spec(s) for s in args.models]
    encoders = [load_encoder(arch, fmt, path, args.max_batch, precision=args.precision) for arch, fmt, path in specs]
    with open(args.pca_model, "rb") as f:
        pca = HipPCA(pickle.load(f))
    with open(args.input_file, encoding="utf-8") as f:
        vids = [x.strip() for x in f if x.strip()]
    scores = read_video_scores(args.video_scores)
    scorer = None
    if args.clip_checkpoint and args.vsm_checkpoint:
        from vsc_hip.video_score import VideoScoreHead, from_reference_state
        clip, _ = load_encoder("clip_vit_l14_224", "clip", args.clip_checkpoint, args.max_batch, u8_norm=(CLIP_MEAN, CLIP_STD),
                               precision=args.precision)
        from src.model_zoo import _state_dict      # a plain / training checkpoint or the TorchScript archive the reference ships (vsm.torchscript.pt)
        scorer = VideoScorer(clip, VideoScoreHead("vsm_roberta_base", from_reference_state(_state_dict(args.vsm_checkpoint))), device)
    videos = zip_videos(vids, args.zip_prefix, sorted({size for _, size in encoders}), with_clip=scorer is not None,
                        workers=args.workers)
    finals, per_model = run_query_videos(videos, encoders, pca.transform, scores, device, score_threshold=args.score_threshold,
                                         scorer=scorer)
    for i, (_, _, path) in enumerate(specs):
        key = os.path.split(path)[-1].split(".")[0]
        os.makedirs(os.path.join(args.output_dir, key), exist_ok=True)
        store_features(os.path.join(args.output_dir, key, f"{args.split}_query.npz"), [sub[i] for sub in per_model])
    if args.norm_refs:
        norm_refs = load_features(args.norm_refs, Dataset.REFS)
        all_scores = {f.video_id: scores.get(f.video_id, 1.0) for f in finals}
        finals = query_score_normalize(finals, norm_refs, all_scores, args.score_threshold, calclualte_low_var_dim(norm_refs),
                                       nk=NK, beta=BETA)
    store_features(os.path.join(args.output_dir, f"{args.split}_query_sn.npz"), finals)


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--split", default="test")
    ap.add_argument("--models", nargs="+", required=True, help="arch:weights_format:checkpoint_path, in concatenation order")
    ap.add_argument("--pca_model", required=True, help="pickled sklearn PCA (mean_, components_, whiten, explained_variance_)")
    ap.add_argument("--zip_prefix", default="")
    ap.add_argument("--input_file", required=True, help="one query video id per line")
    ap.add_argument("--norm_refs", default="", help="score-normalisation reference descriptors (.npz)")
    ap.add_argument("--video_scores", default="", help="csv video_id,score from the video-score model")
    ap.add_argument("--clip_checkpoint", default="", help="CLIP ViT-L/14 visual tower state dict (video-score gate)")
    ap.add_argument("--vsm_checkpoint", default="", help="MS video-score head state dict (epoch_*.pth of train_vid_score)")
    ap.add_argument("--score_threshold", type=float, default=SCORE_THRESHOLD)
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--max_batch", type=int, default=None,
                    help="frames per encoder call; default: each backbone's tile-aligned batch (332 / 451 / 255 / 256)")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["fp16", "bf16"],
                    help="16-bit type of the encoders' MFMA operands (same speed; fp16 = 8 x smaller rounding, DESIGN.md 3a)")
    ap.add_argument("--workers", type=int, default=6, help="decode / resize worker processes (the reference uses 6)")
    return ap


if __name__ == "__main__":
    main(build_parser().parse_args())
