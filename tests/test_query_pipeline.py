"""Query-side extraction loop (extract_query_feats.py mirror): plumbing on CPU with stand-in encoders, the real
ensemble (one ViT + one Swin-V2 HIP encoder at different input sizes) on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from src import synth  # noqa: E402
from src.model_zoo import load_encoder, parse_model_spec  # noqa: E402
from src.query_pipeline import run_query_videos  # noqa: E402
from src.query_postprocess import HipPCA, process_query_video  # noqa: E402


class _NumpyOps:
    @staticmethod
    def normalize(x):
        x = np.asarray(x, np.float32)
        return x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)

    @staticmethod
    def self_similarity(x):
        return x @ x.T

    @staticmethod
    def similarity(a, b):
        return a @ b.T


class _FakeEncoder:
    """frames [S,3,H,W] -> [S, dim] deterministic features (call shape of the HIP encoders)."""

    def __init__(self, dim, seed, tokens=False):
        self.proj = torch.from_numpy(synth.normalish(seed, (3, dim)))
        self.tokens = tokens

    def __call__(self, frames):
        f = frames.float().mean(dim=(2, 3)) @ self.proj + frames.float().flatten(1)[:, :1]
        return f[:, None, :].repeat(1, 2, 1) if self.tokens else f


def _videos(sizes, lens):
    out = []
    for i, n in enumerate(lens):
        frames = {s: torch.from_numpy(synth.uniform(100 * i + s, (n, 3, s, s))) for s in sizes}
        out.append((f"Q{i:06d}", frames, np.arange(n)))
    return out


def test_run_query_videos_composes_the_reference_steps():
    vids = _videos((8, 12), [9, 5, 7])
    enc = [(_FakeEncoder(6, 1), 8), (_FakeEncoder(4, 2, tokens=True), 12)]   # the second returns [S, T, D]: row 0 is taken
    pca = lambda x: x[:, :5] * 2.0
    scores = {"Q000001": 0.0}  # rejected; the others default to accepted
    finals, per_model = run_query_videos(vids, enc, pca, scores, torch.device("cpu"), ops=_NumpyOps, chunk=4)
    assert [f.video_id for f in finals] == ["Q000000", "Q000001", "Q000002"] and len(per_model) == 3
    rnd = 0
    for (vid, frames, stamps), got, subs in zip(vids, finals, per_model):
        raw = [enc[0][0](frames[8]).numpy(), enc[1][0](frames[12])[:, 0].numpy()]
        want, want_subs, rnd = process_query_video(vid, raw, stamps, scores.get(vid, 1.0), pca, rnd, ops=_NumpyOps)
        assert np.array_equal(got.feature, want.feature) and np.array_equal(got.timestamps, want.timestamps)
        assert len(subs) == 2 and all(np.array_equal(a.feature, b.feature) for a, b in zip(subs, want_subs))
    assert finals[1].feature.shape == (1, 512) and np.abs(finals[1].feature).max() <= 1e-5   # the rejected video
    assert finals[0].feature.shape[1] == 5


def test_model_spec_and_zoo_errors(tmp_path):
    assert parse_model_spec("vit_v68:timm_vit:/x/y:z.pth") == ("vit_v68", "timm_vit", "/x/y:z.pth")
    with pytest.raises(ValueError, match="arch:weights_format:checkpoint_path"):
        parse_model_spec("vit_v68:timm_vit")
    with pytest.raises(ValueError, match="unknown arch"):
        load_encoder("resnet50", "hf_vit", "nowhere.pth", 8)
    with pytest.raises(ValueError, match="swin_ref"):
        load_encoder("tiny_swin", "hf_vit", "nowhere.pth", 8)
    with pytest.raises(ValueError, match="ViT preset"):
        load_encoder("tiny", "swin_ref", "nowhere.pth", 8)


def test_video_scores_csv(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
    import extract_query_feats as E
    p = tmp_path / "s.csv"
    p.write_text("video_id,score\nQ000001,0.25\nQ000002,1e-4\n")
    assert E.read_video_scores(str(p)) == {"Q000001": 0.25, "Q000002": 1e-4} and E.read_video_scores("") == {}
    a = E.build_parser().parse_args(["--models", "tiny:hf_vit:a.pth", "--pca_model", "p.pkl", "--input_file", "q.txt"])
    assert a.split == "test" and a.score_threshold == 0.001 and a.max_batch == 256


@pytest.mark.gpu
def test_ensemble_on_hip_encoders():
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    from src.query_postprocess import HipOps
    dev = torch.device("cuda", 0)
    vcfg, scfg = get_config("tiny"), get_swin_config("tiny_swin")
    vit = HipEncoder(vcfg, synth.encoder_weights(3, vcfg), max_batch=8)
    swin = SwinHipEncoder(scfg, synth.swin_weights(4, scfg), max_batch=8)
    sizes = (vcfg.image_size, scfg.image_size)
    vids = []
    for i, n in enumerate([11, 4]):
        vids.append((f"Q{i:06d}", {vcfg.image_size: torch.from_numpy(synth.frames(10 + i, n, vcfg)),
                                   scfg.image_size: torch.from_numpy(synth.swin_frames(20 + i, n, scfg))}, np.arange(n)))

    class Fitted:
        mean_ = None
        components_ = synth.normalish(5, (16, vcfg.out_dim + scfg.out_dim)) / 4.0
        whiten = False

    pca = HipPCA(Fitted)
    finals, per_model = run_query_videos(vids, [(vit, sizes[0]), (swin, sizes[1])], pca.transform, {}, dev, chunk=8)
    for (vid, frames, stamps), got, subs in zip(vids, finals, per_model):
        a = HipOps.normalize(vit(frames[sizes[0]].to(dev)).cpu().numpy())
        b = HipOps.normalize(swin(frames[sizes[1]].to(dev)).cpu().numpy())
        assert np.array_equal(subs[0].feature, a) and np.array_equal(subs[1].feature, b)   # chunking changes nothing
        assert got.feature.shape[1] == 16 and got.feature.shape[0] <= len(stamps) and np.isfinite(got.feature).all()
        assert got.video_id == vid
