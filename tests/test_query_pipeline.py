"""Query-side extraction loop (extract_query_feats.py mirror): plumbing on CPU with stand-in encoders, the real
ensemble (one ViT + one Swin-V2 HIP encoder at different input sizes) on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools import synth  # noqa: E402
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


@pytest.mark.parametrize("group_frames", [1, 12, 512])   # one video per group, two groups, everything in one group
def test_run_query_videos_composes_the_reference_steps(group_frames):
    vids = _videos((8, 12), [9, 5, 7])
    enc = [(_FakeEncoder(6, 1), 8), (_FakeEncoder(4, 2, tokens=True), 12)]   # the second returns [S, T, D]: row 0 is taken
    pca = lambda x: x[:, :5] * 2.0
    scores = {"Q000001": 0.0}  # rejected; the others default to accepted
    finals, per_model = run_query_videos(vids, enc, pca, scores, torch.device("cpu"), ops=_NumpyOps, chunk=4,
                                         group_frames=group_frames)
    assert [f.video_id for f in finals] == ["Q000000", "Q000001", "Q000002"] and len(per_model) == 3
    rnd = 0
    for (vid, frames, stamps), got, subs in zip(vids, finals, per_model):
        raw = [enc[0][0](frames[8]).numpy(), enc[1][0](frames[12])[:, 0].numpy()]
        want, want_subs, rnd = process_query_video(vid, raw, stamps, scores.get(vid, 1.0), pca, rnd, ops=_NumpyOps)
        assert np.array_equal(got.feature, want.feature) and np.array_equal(got.timestamps, want.timestamps)
        assert len(subs) == 2 and all(np.array_equal(a.feature, b.feature) for a, b in zip(subs, want_subs))
    assert finals[1].feature.shape == (1, 512) and np.abs(finals[1].feature).max() <= 1e-5   # the rejected video
    assert finals[0].feature.shape[1] == 5


@pytest.mark.parametrize("two_d", [True, False])
def test_rejected_video_round_trips_through_store_features(tmp_path, two_d):
    """The gate's placeholder ([1, 2] timestamps in the reference, extract_query_feats.py:214-217) must concatenate
    with the accepted videos' timestamps in store_features -- the query entry point stores `finals` as its last step."""
    from vsc.storage import load_features, store_features
    vids = _videos((8,), [6, 4, 5])
    if two_d:   # the layout QueryVideos / the reference's FFmpeg reader yields: [start, end] per frame
        vids = [(v, f, np.stack([t, t + 1], axis=1).astype(np.float32)) for v, f, t in vids]
    enc = [(_FakeEncoder(6, 1), 8)]
    finals, _ = run_query_videos(vids, enc, lambda x: x[:, :6], {"Q000001": 0.0}, torch.device("cpu"), ops=_NumpyOps)
    finals = [f._replace(feature=np.pad(f.feature, ((0, 0), (0, 512 - f.feature.shape[1])))) if hasattr(f, "_replace")
              else type(f)(video_id=f.video_id, timestamps=f.timestamps,
                           feature=np.pad(f.feature, ((0, 0), (0, 512 - f.feature.shape[1])))) for f in finals]
    path = str(tmp_path / "q.npz")
    store_features(path, finals)
    back = load_features(path)
    assert [b.video_id for b in back] == [f.video_id for f in finals]
    for a, b in zip(finals, back):
        assert np.array_equal(np.asarray(a.timestamps, np.float32), np.asarray(b.timestamps, np.float32))
        assert np.array_equal(a.feature, b.feature)
    assert back[1].feature.shape == (1, 512) and back[1].timestamps.shape == ((1, 2) if two_d else (1,))


def test_scorer_overrides_and_records_scores():
    vids = _videos((8,), [6, 6])
    for v in vids:
        v[1]["clip"] = v[1][8]
    enc = [(_FakeEncoder(6, 1), 8)]
    calls = []

    def scorer(frames):
        calls.append(frames.shape[0])
        return 0.0 if len(calls) == 1 else 0.9   # first video rejected, second accepted

    scores = {"Q000000": 1.0}   # would accept the first video: the scorer's verdict wins and is written back
    finals, _ = run_query_videos(vids, enc, lambda x: x[:, :3], scores, torch.device("cpu"), ops=_NumpyOps, scorer=scorer)
    assert calls == [6, 6] and scores == {"Q000000": 0.0, "Q000001": 0.9}
    assert finals[0].feature.shape == (1, 512) and finals[1].feature.shape[1] == 3


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
    assert a.split == "test" and a.score_threshold == 0.001 and a.max_batch is None


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


@pytest.mark.gpu
def test_reference_side_extraction_groups_loader_batches():
    """src.extractor.extract_vsc_feat on the HIP encoder: the valid frames of consecutive loader batches go through the encoder in groups
    (pinned staging, aligned chunks, one copy back per group) -- the same rows as encoding every video on its own, padding dropped,
    order kept; a group boundary inside the run (group_frames = 16) and the default single group."""
    from src.dataset import TensorFrames, collate_fn
    from src.extractor import extract_vsc_feat
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    dev = torch.device("cuda", 0)
    cfg = get_config("tiny")
    enc = HipEncoder(cfg, synth.encoder_weights(3, cfg), max_batch=8)
    lens = [5, 11, 1, 7, 9, 3]
    vids = [(torch.from_numpy(synth.frames(30 + i, n, cfg)), f"R{i:06d}") for i, n in enumerate(lens)]
    want = np.concatenate([enc(f.to(dev)).cpu().numpy() for f, _ in vids])
    for group_frames in (16, 2048):
        loader = torch.utils.data.DataLoader(TensorFrames(vids), batch_size=2, collate_fn=collate_fn)
        ids, feats, stamps = extract_vsc_feat(enc, loader, dev, group_frames=group_frames)
        assert ids == [v for (_, v), n in zip(vids, lens) for _ in range(n)]
        assert stamps.tolist() == [t for n in lens for t in range(n)]
        assert np.array_equal(feats, want)


@pytest.mark.gpu
def test_group_batched_postprocessing_equals_the_per_video_steps():
    """run_query_videos on the GPU batches normalisation / similarity / PCA per group of videos and keeps the features on the device
    (process_query_group, pinned staging in encode_group): bit for bit the per-video process_query_video on the same features,
    including a rejected video's placeholder and the rnd_idx sequence, for any grouping."""
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    from src.query_postprocess import HipOps
    dev = torch.device("cuda", 0)
    vcfg, scfg = get_config("tiny"), get_swin_config("tiny_swin")
    vit = HipEncoder(vcfg, synth.encoder_weights(3, vcfg), max_batch=8)
    swin = SwinHipEncoder(scfg, synth.swin_weights(4, scfg), max_batch=8)
    vids = []
    for i, n in enumerate([11, 4, 9, 1, 6]):
        fr = synth.frames(10 + i, n, vcfg)
        fr[n // 2] = fr[0]                       # a duplicated frame: the filter has something to drop
        vids.append((f"Q{i:06d}", {vcfg.image_size: torch.from_numpy(fr), scfg.image_size: torch.from_numpy(synth.swin_frames(20 + i, n, scfg))}, np.arange(n)))

    class Fitted:
        mean_ = synth.normalish(6, (vcfg.out_dim + scfg.out_dim,)) * 0.01
        components_ = synth.normalish(5, (16, vcfg.out_dim + scfg.out_dim)) / 4.0
        whiten = False

    pca = HipPCA(Fitted)
    enc = [(vit, vcfg.image_size), (swin, scfg.image_size)]
    scores = {"Q000001": 0.0, "Q000003": 0.0005}
    for group_frames in (1, 14, 1024):
        finals, per_model = run_query_videos(vids, enc, pca.transform, dict(scores), dev, chunk=8, group_frames=group_frames)
        rnd = 0
        for (vid, frames, stamps), got, subs in zip(vids, finals, per_model):
            raw = [vit(frames[vcfg.image_size].to(dev)).cpu().numpy(), swin(frames[scfg.image_size].to(dev)).cpu().numpy()]
            want, want_subs, rnd = process_query_video(vid, raw, stamps, scores.get(vid, 1.0), pca.transform, rnd, ops=HipOps)
            assert np.array_equal(got.feature, want.feature) and np.array_equal(got.timestamps, want.timestamps), (group_frames, vid)
            assert all(np.array_equal(a.feature, b.feature) for a, b in zip(subs, want_subs))
    assert len(finals[0].feature) < 11      # the duplicate was dropped
    # a video WITHOUT frames (an unreadable file) inside a group, and a group made only of such videos: the other videos' results are
    # untouched and the frameless ones get the placeholder descriptor (the group path indexed an empty list / aborted the group)
    empty = lambda name: (name, {vcfg.image_size: torch.zeros((0,) + tuple(vids[0][1][vcfg.image_size].shape[1:])),
                                 scfg.image_size: torch.zeros((0,) + tuple(vids[0][1][scfg.image_size].shape[1:]))}, np.arange(0))
    mixed = [vids[0], empty("Q000100"), vids[2]]
    f_mixed, _ = run_query_videos(mixed, enc, pca.transform, {}, dev, chunk=8, group_frames=1024)
    f_plain, _ = run_query_videos([vids[0], vids[2]], enc, pca.transform, {}, dev, chunk=8, group_frames=1024)
    assert np.array_equal(f_mixed[0].feature, f_plain[0].feature) and np.array_equal(f_mixed[2].feature, f_plain[1].feature)
    assert f_mixed[1].feature.shape == (1, 512) and np.abs(f_mixed[1].feature).max() <= 1e-5
    f_none, pm_none = run_query_videos([empty("Q000101"), empty("Q000102")], enc, pca.transform, {}, dev, chunk=8, group_frames=1024)
    assert [f.feature.shape for f in f_none] == [(1, 512), (1, 512)] and all(len(sub.feature) == 0 for pm in pm_none for sub in pm)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("bf16", 1e-2), ("fp16", 2e-3)])
def test_video_scorer_on_hip_path(precision, tol):
    """CLIP tower (tiny_clip preset) -> MS head (a vsm preset matched to its width) -> sigmoid, vs the oracles; the tower in both operand
    types (the reference runs it under fp16 autocast: extract_query_feats.py:159), the head is fp32 in both."""
    from oracle import vit_oracle, vsm_oracle
    from src.query_pipeline import VideoScorer
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    from vsc_hip.video_score import VideoScoreHead
    from vsc_hip.vsm_config import get_vsm_config
    dev = torch.device("cuda", 0)
    ccfg = get_config("tiny_clip")
    vcfg = get_vsm_config("tiny_vsm", feat_dim=ccfg.width)
    cw, vw = synth.encoder_weights(8, ccfg), synth.vsm_weights(9, vcfg)
    scorer = VideoScorer(HipEncoder(ccfg, cw, max_batch=8, precision=precision), VideoScoreHead(vcfg, vw), dev, chunk=5)
    frames = torch.from_numpy(synth.frames(30, 14, ccfg))   # more frames than max_frames (12): truncated like the reference
    got = scorer(frames)
    cls = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in cw.items()}, ccfg, frames[: vcfg.max_frames], l2=False)
    want = vsm_oracle.video_score(vw, vcfg, cls)
    assert 0.0 < got < 1.0 and abs(got - want) < tol, (precision, got, want)


def test_query_videos_dataset_reads_zips(tmp_path):
    """zip of jpgs -> uint8 frames per input size, through the DataLoader the entry point uses."""
    import io
    from zipfile import ZipFile
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
    import extract_query_feats as E
    rng = np.random.RandomState(0)
    for vid, n in (("Q100001", 3), ("Q100002", 2)):
        d = tmp_path / vid[-2:]
        d.mkdir(exist_ok=True)
        with ZipFile(d / f"{vid}.zip", "w") as z:
            for i in range(n):
                buf = io.BytesIO()
                Image.fromarray(rng.randint(0, 255, (40, 60, 3), dtype=np.uint8)).save(buf, format="JPEG")
                z.writestr(f"{i:04d}.jpg", buf.getvalue())
    items = list(E.zip_videos(["Q100001", "Q100002", "Q100099"], str(tmp_path), [16, 24], with_clip=True, workers=0))
    assert [v[0] for v in items] == ["Q100001", "Q100002"]           # the missing video is skipped, as ZipFrames does
    vid, frames, stamps = items[0]
    assert frames[16].shape == (3, 16, 16, 3) and frames[24].shape == (3, 24, 24, 3) and frames["clip"].shape == (3, 224, 224, 3)
    assert all(f.dtype == torch.uint8 for f in frames.values()) and stamps.tolist() == [[0, 1], [1, 2], [2, 3]]


def test_model_zoo_reads_torchscript_and_plain_checkpoints(tmp_path):
    """The reference ships traced .torchscript.pt files: their state dict keeps the module's parameter names."""
    from src.model_zoo import _state_dict

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(4, 3)
            self.norm = torch.nn.LayerNorm(3)

        def forward(self, x):
            return self.norm(self.proj(x))

    m = Tiny().eval()
    torch.jit.save(torch.jit.trace(m, torch.zeros(2, 4)), str(tmp_path / "m.torchscript.pt"))
    torch.save({"state_dict": m.state_dict()}, str(tmp_path / "m.pth"))
    torch.save(m.state_dict(), str(tmp_path / "plain.pth"))
    want = {k: v for k, v in m.state_dict().items()}
    for name in ("m.torchscript.pt", "m.pth", "plain.pth"):
        got = _state_dict(str(tmp_path / name))
        assert sorted(got) == sorted(want) and all(torch.equal(got[k], want[k]) for k in want), name
