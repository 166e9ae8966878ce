"""End-to-end query-side extraction rate of the reference's ensemble on synthetic videos (run on the GPU box):
3 x Swin-V2-B/256 + vit_v68 (ViT-B/32-384 + SSCD head) + the video-score gate (CLIP ViT-L/14 -> MS head), L2-normalise,
concatenate (2048-d), near-duplicate filter, PCA 2048 -> 512 -- src/query_pipeline.run_query_videos, i.e. what
extract_query_feats.py does per query video.      python tools/ensemble_bench.py [videos] [frames_per_video]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tools import synth
from src.query_pipeline import VideoScorer, run_query_videos
from src.query_postprocess import HipPCA
from vsc_hip.config import aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder
from vsc_hip.video_score import VideoScoreHead
from vsc_hip.vsm_config import get_vsm_config

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n_videos = int(_pos[0]) if len(_pos) > 0 else 32
n_frames = int(_pos[1]) if len(_pos) > 1 else 40
dev = torch.device("cuda:0")
scfg, vcfg, ccfg, mcfg = get_swin_config("swinv2_base_256"), get_config("vit_v68"), get_config("clip_vit_l14_224"), get_vsm_config("vsm_roberta_base")
t0 = time.perf_counter()
swins = [SwinHipEncoder(scfg, synth.swin_weights(40 + i, scfg), max_batch=256) for i in range(3)]   # 256 = its aligned batch
vit = HipEncoder(vcfg, synth.encoder_weights(50, vcfg), max_batch=aligned_batch(vcfg.tokens))
from src.dataset import CLIP_MEAN, CLIP_STD  # noqa: E402
scorer = VideoScorer(HipEncoder(ccfg, synth.encoder_weights(51, ccfg), max_batch=aligned_batch(ccfg.tokens), u8_mean=CLIP_MEAN, u8_std=CLIP_STD),
                     VideoScoreHead(mcfg, synth.vsm_weights(52, mcfg)), dev)
U8 = "--f32" not in sys.argv   # decoded uint8 HWC frames (default) or the reference's fp32 CHW tensors
print(f"ensemble built in {time.perf_counter() - t0:.1f} s")


class Fitted:
    mean_ = synth.normalish(60, (4 * 512,)) * 0.01
    components_ = synth.normalish(61, (512, 4 * 512)) / 45.0
    whiten = False


def videos():
    if U8:
        base = {k: torch.from_numpy(synth.uniform(s, (8, size, size, 3), 0.0, 256.0).astype(np.uint8))
                for k, s, size in ((256, 1, 256), (384, 2, 384), ("clip", 3, 224))}
    else:
        base = {256: torch.from_numpy(synth.swin_frames(1, 8, scfg)), 384: torch.from_numpy(synth.frames(2, 8, vcfg)),
                "clip": torch.from_numpy(synth.frames(3, 8, ccfg))}
    for v in range(n_videos):
        reps = (n_frames + 7) // 8
        yield (f"Q{v:06d}", {k: f.repeat(reps, 1, 1, 1)[:n_frames].clone() for k, f in base.items()}, np.arange(n_frames))


encoders = [(m, 256) for m in swins] + [(vit, 384)]
pca = HipPCA(Fitted)
vids = list(videos())
run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=scorer)   # warm-up: same shapes as the timed run
torch.cuda.synchronize()
t0 = time.perf_counter()
finals, _ = run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=scorer)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
total = n_videos * n_frames
print(f"[{'uint8 HWC' if U8 else 'fp32 CHW'} frames] {n_videos} videos x {n_frames} frames: {dt:.2f} s -> {total / dt:.0f} query frames/s through the whole ensemble "
      f"({finals[0].feature.shape[1]}-d descriptors, {sum(len(f.feature) for f in finals)} frames kept)")

# ---- where the time goes (synchronising wrappers; slower than the run above)
import src.query_pipeline as qp  # noqa: E402

acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return wrap


qp.encode_group = timed("backbones + CLIP tower (uploads included)", qp.encode_group)
qp.process_query_video = timed("per-video post-processing (normalise, de-dup, PCA)", qp.process_query_video)
run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=timed("video-score gate (CLIP + MS head)", scorer))
for k, v in acc.items():
    print(f"  {k:52s} {v:6.2f} s")
