"""End-to-end query-side extraction rate of the reference's ensemble on synthetic videos (run on the GPU box):
3 x Swin-V2-B/256 + vit_v68 (ViT-B/32-384 + SSCD head) + the video-score gate (CLIP ViT-L/14 -> MS head), L2-normalise,
concatenate (2048-d), near-duplicate filter, PCA 2048 -> 512 -- src/query_pipeline.run_query_videos, i.e. what
infer/extract_query_feats.py:143-254 does per query video with the models infer/infer_ref.sh:7 lists.
      python tools/ensemble_bench.py [videos] [frames_per_video] [--f32] [--breakdown] [--ragged]
`measure()` is what bench.py reports as its `ensemble` secondary: the end-to-end rate from uint8 HOST frames, every model's own
rate on device-resident frames, and the encoder-bound rate those imply (1 / sum of the models' per-frame times)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "vsc22-submission_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np
import torch

from tools import synth


class _Fitted:
    mean_ = synth.normalish(60, (4 * 512,)) * 0.01
    components_ = synth.normalish(61, (512, 4 * 512)) / 45.0
    whiten = False


def build(dev, precision="bf16"):
    from src.dataset import CLIP_MEAN, CLIP_STD
    from src.query_pipeline import VideoScorer
    from vsc_hip.config import aligned_batch, get_config
    from vsc_hip.encoder import HipEncoder
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    from vsc_hip.video_score import VideoScoreHead
    from vsc_hip.vsm_config import get_vsm_config
    scfg, vcfg, ccfg, mcfg = get_swin_config("swinv2_base_256"), get_config("vit_v68"), get_config("clip_vit_l14_224"), get_vsm_config("vsm_roberta_base")
    swins = [SwinHipEncoder(scfg, synth.swin_weights(40 + i, scfg), max_batch=256, precision=precision) for i in range(3)]   # 256 = its aligned batch
    vit = HipEncoder(vcfg, synth.encoder_weights(50, vcfg), max_batch=aligned_batch(vcfg.tokens), precision=precision)
    clip = HipEncoder(ccfg, synth.encoder_weights(51, ccfg), max_batch=aligned_batch(ccfg.tokens), u8_mean=CLIP_MEAN, u8_std=CLIP_STD, precision=precision)
    scorer = VideoScorer(clip, VideoScoreHead(mcfg, synth.vsm_weights(52, mcfg)), dev)
    return {"swins": swins, "vit": vit, "clip": clip, "scorer": scorer, "cfgs": (scfg, vcfg, ccfg)}


def videos(models, n_videos, n_frames, u8=True, ragged=False):
    """ragged: video v has 10 + 37 v mod (2 n_frames - 19) frames instead of n_frames each (the same total on average)"""
    scfg, vcfg, ccfg = models["cfgs"]
    if u8:
        base = {k: torch.from_numpy(synth.uniform(s, (8, size, size, 3), 0.0, 256.0).astype(np.uint8))
                for k, s, size in ((256, 1, 256), (384, 2, 384), ("clip", 3, 224))}
    else:
        base = {256: torch.from_numpy(synth.swin_frames(1, 8, scfg)), 384: torch.from_numpy(synth.frames(2, 8, vcfg)),
                "clip": torch.from_numpy(synth.frames(3, 8, ccfg))}
    lens = [10 + (37 * v) % (2 * n_frames - 19) if ragged else n_frames for v in range(n_videos)]
    reps = (max(lens) + 7) // 8
    return [(f"Q{v:06d}", {k: f.repeat(reps, 1, 1, 1)[:n].clone() for k, f in base.items()}, np.arange(n)) for v, n in enumerate(lens)]


def model_rate(model, frames_u8, dev, batch, steps=3):
    """frames/s of one encoder on device-resident uint8 frames, `batch` frames per call"""
    x = frames_u8.to(dev).repeat((batch + frames_u8.shape[0] - 1) // frames_u8.shape[0], 1, 1, 1)[:batch].contiguous()
    model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(x)
    torch.cuda.synchronize()
    return batch * steps / (time.perf_counter() - t0)


def measure(dev, n_videos=52, n_frames=40, u8=True, breakdown=False, ragged=False, group_frames=None, precision=None):
    """precision: the encoders' 16-bit operand type; None = what the extract_* entry points default to (src.model_zoo.DEFAULT_PRECISION)"""
    if precision is None:
        from src.model_zoo import DEFAULT_PRECISION as precision
    from src.query_pipeline import run_query_videos
    from src.query_postprocess import HipPCA
    t0 = time.perf_counter()
    m = build(dev, precision)
    build_s = time.perf_counter() - t0
    encoders = [(s, 256) for s in m["swins"]] + [(m["vit"], 384)]
    pca = HipPCA(_Fitted)
    vids = videos(m, n_videos, n_frames, u8, ragged)
    kw = {} if group_frames is None else {"group_frames": group_frames}
    run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=m["scorer"], **kw)   # warm-up: same shapes as the timed run
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    finals, _ = run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=m["scorer"], **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = sum(len(v[2]) for v in vids)
    out = {"metric": "query frames/s through the reference's ensemble, end to end from uint8 host frames (3 x Swin-V2-B/256 + vit_v68 + CLIP ViT-L/14 "
                     "video-score gate, normalise, concatenate, near-duplicate filter, PCA 2048 -> 512: infer/extract_query_feats.py:143-254, infer/infer_ref.sh:7)",
           "value": round(total / dt, 1), "unit": "frames/s", "operands": precision, "videos": n_videos, "frames_per_video": n_frames, "seconds": round(dt, 3),
           "descriptor_dim": int(finals[0].feature.shape[1]), "frames_kept": int(sum(len(f.feature) for f in finals)),
           "input": "uint8 HWC host tensors (pageable), pinned staging + copy stream" if u8 else "fp32 CHW host tensors", "build_seconds": round(build_s, 1)}
    if u8:
        b = vids[0][1]
        rates = {"swinv2_base_256": model_rate(m["swins"][0], b[256], dev, 512), "vit_v68": model_rate(m["vit"], b[384], dev, 2 * m["vit"].preferred_batch),
                 "clip_vit_l14_224": model_rate(m["clip"], b["clip"], dev, 2 * m["clip"].preferred_batch)}
        per_frame = 3.0 / rates["swinv2_base_256"] + 1.0 / rates["vit_v68"] + 1.0 / rates["clip_vit_l14_224"]
        out["models_frames_per_s"] = {k: round(v, 1) for k, v in rates.items()}
        out["encoder_bound_frames_per_s"] = round(1.0 / per_frame, 1)
        out["fraction_of_encoder_bound"] = round(total / dt * per_frame, 3)
        if not ragged and n_videos >= 8:
            # the same models on videos of DIFFERENT lengths (what query sets are): 10 .. 2 n_frames - 10 frames each
            rv = videos(m, n_videos, n_frames, u8, True)
            run_query_videos(rv, encoders, pca.transform, {}, dev, scorer=m["scorer"], **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_query_videos(rv, encoders, pca.transform, {}, dev, scorer=m["scorer"], **kw)
            torch.cuda.synchronize()
            rdt, rtotal = time.perf_counter() - t0, sum(len(v[2]) for v in rv)
            out["ragged_lengths"] = {"value": round(rtotal / rdt, 1), "unit": "frames/s", "frames": rtotal,
                                     "frames_per_video": [min(len(v[2]) for v in rv), max(len(v[2]) for v in rv)],
                                     "fraction_of_encoder_bound": round(rtotal / rdt * per_frame, 3)}
        out["host_bytes_per_frame"] = int(sum(int(np.prod(t.shape[1:])) for t in b.values()))
    if breakdown:
        import src.query_pipeline as qp
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

        saved = (qp.encode_group, qp.process_query_group, m["scorer"].head.logit)
        qp.encode_group = timed("encode_group: staging + backbones + CLIP tower", qp.encode_group)
        qp.process_query_group = timed("process_query_group: normalise, de-dup, PCA", qp.process_query_group)
        m["scorer"].head.logit = timed("MS head (per video)", m["scorer"].head.logit)
        t0 = time.perf_counter()
        run_query_videos(vids, encoders, pca.transform, {}, dev, scorer=m["scorer"])
        acc["whole run with synchronising wrappers"] = time.perf_counter() - t0
        qp.encode_group, qp.process_query_group, m["scorer"].head.logit = saved
        out["breakdown_seconds"] = {k: round(v, 3) for k, v in acc.items()}
    for e in m["swins"] + [m["vit"], m["clip"]]:
        e.close()
    return out


if __name__ == "__main__":
    _pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    import json
    r = measure(torch.device("cuda:0"), int(_pos[0]) if _pos else 52, int(_pos[1]) if len(_pos) > 1 else 40,
                u8="--f32" not in sys.argv, breakdown="--breakdown" in sys.argv, ragged="--ragged" in sys.argv,
                group_frames=next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--group=")), None))
    print(json.dumps(r, indent=1))
