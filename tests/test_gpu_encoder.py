"""Frame -> descriptor parity of the HIP encoder (through vsc_encoder_* of the C ABI)
against the golden vectors (transformers' ViTModel / CLIPVisionModel outputs committed
under tests/golden) and against the fp32 oracle on fresh inputs.

Tolerance: BASELINE.json's north_star asks for L2-normalised descriptors within 1e-3
(absolute) of the fp32 reference with bf16 MFMA compute; un-normalised features and
hidden states are checked relative to their scale.
"""
import os

import numpy as np
import pytest
import torch

import parity_bounds
from tools import synth
from vsc_hip.config import get_config

pytestmark = pytest.mark.gpu

DESC_L2_ATOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


def _encoder(preset, seed, **kw):
    from vsc_hip.encoder import HipEncoder
    cfg = get_config(preset)
    w = synth.encoder_weights(seed, cfg)
    return cfg, w, HipEncoder(cfg, w, **kw)


@pytest.mark.parametrize("preset", ["tiny", "tiny_clip", "vit_b16_224", "vit_v68"])
def test_encoder_matches_golden(dev, preset, golden_dir):
    g = np.load(os.path.join(golden_dir, f"vit_{preset}.npz"))
    cfg, _, enc = _encoder(preset, int(g["weights_seed"]), max_batch=4)
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    desc, tok = enc(x, return_tokens=True)
    desc, tok = desc.cpu().numpy(), tok.cpu().numpy()
    # hidden states are O(1) after the final LayerNorm; bf16 GEMM inputs through `layers` blocks
    assert np.abs(tok[:, :4] - g["tokens_head"]).max() < 0.08
    assert np.abs(tok[:, -2:] - g["tokens_tail"]).max() < 0.08
    assert np.abs(tok[:, :4] - g["tokens_head"]).mean() < 0.01
    scale = np.abs(g["desc"]).max()
    assert np.abs(desc - g["desc"]).max() < 0.02 * scale
    enc2 = _encoder(preset, int(g["weights_seed"]), max_batch=2, l2_normalize=True)[2]
    d2 = enc2(x).cpu().numpy()
    np.testing.assert_allclose(d2, g["desc_l2"], rtol=0, atol=DESC_L2_ATOL)
    parity_bounds.check(f"vit/{preset}", d2, g["desc_l2"])      # mean |d| and |mean d|: a bias does not average out
    np.testing.assert_allclose(np.linalg.norm(d2, axis=1), 1.0, atol=1e-5)


def test_encoder_matches_golden_on_frames_that_differ(dev, golden_dir):
    """The second ViT-B/16 fixture: frames of different structure (tools/synth.py structured_frames) instead of i.i.d. noise.  Noise
    frames all look alike to a random-weight network -- the first fixture's descriptors have cosine 0.98 to one another, so its 1e-3
    tolerance is 4 % of the frame-to-frame signal; here the cosines are 0.49 .. 0.9 and the same tolerance is well under 1 % of it."""
    g = np.load(os.path.join(golden_dir, "vit_vit_b16_224_structured.npz"))
    cfg, _, enc = _encoder("vit_b16_224", int(g["weights_seed"]), max_batch=4, l2_normalize=True)
    x = torch.from_numpy(synth.structured_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    d = enc(x).cpu().numpy()
    ref = g["desc_l2"]
    cos = (ref @ ref.T)[np.triu_indices(len(ref), 1)]
    assert cos.min() < 0.6 and cos.mean() < 0.8                       # the fixture is what it claims to be
    np.testing.assert_allclose(d, ref, rtol=0, atol=DESC_L2_ATOL)
    parity_bounds.check("vit/vit_b16_224_structured", d, ref)
    assert np.abs((d @ d.T)[np.triu_indices(len(ref), 1)] - cos).max() < 1e-3      # and the frame-to-frame geometry is the reference's


@pytest.mark.parametrize("precision,bound", [("bf16", 1e-3), ("fp16", 2e-4)])
def test_encoder_outlier_fixture(dev, golden_dir, precision, bound):
    """Weights with a residual channel riding at ~100 through the whole network (|x| max 104 in the middle), LayerNorm gains x 20, hidden units
    at 40 (tools/synth.vit_outlier_weights); golden = the reference's own VIT wrapper (check_golden_against_reference.vit_outlier: 0.0).  The
    default path (explicit fp32 LayerNorm, then the operand rounding) holds its bound in both operand types.  The LayerNorm-FOLDED path
    (fuse_ln = 1, opt-in) feeds the GEMMs the rounded RAW residual stream: the offset channel's token-to-token variation drowns in the rounding
    of 100 -- this fixture is why it is not the default; its error is reported, and bounded only loosely."""
    from vsc_hip.encoder import HipEncoder
    g = np.load(f"{golden_dir}/vit_vit_b16_224_outlier.npz")
    cfg = get_config("vit_b16_224")
    w = synth.vit_outlier_weights(int(g["weights_seed"]), cfg)
    x = torch.from_numpy(synth.structured_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    errs = {}
    for fuse in (0, 1):
        enc = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, precision=precision, fuse_ln=fuse)
        d = enc(x).cpu().numpy()
        enc.close()
        assert np.isfinite(d).all()
        errs[fuse] = float(np.abs(d - g["desc_l2"]).max())
    print(f"{precision} operands, ViT outlier fixture: max |d| {errs[0]:.2e} (explicit LayerNorm), {errs[1]:.2e} (fuse_ln)")
    assert errs[0] <= bound, errs
    assert errs[1] <= 5e-2, errs


def test_mean_bound_sees_a_one_percent_scale_error(dev, golden_dir):
    """The net itself under test: ONE weight tensor of the HIP encoder off by 1 % (5 % for a bias) passes the 1e-3 maximum bound
    and must FAIL the mean bound of parity_bounds (measured: 1.34e-4 / 1.9e-4 / 1.09e-4 against 1.07e-4)."""
    g = np.load(os.path.join(golden_dir, "vit_vit_b16_224.npz"))
    cfg = get_config("vit_b16_224")
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    from vsc_hip.encoder import HipEncoder
    for key, f in (("blocks.5.fc1.weight", 1.01), ("blocks.0.qkv.bias", 1.05), ("blocks.11.proj.weight", 1.01)):
        w = synth.encoder_weights(int(g["weights_seed"]), cfg)
        w[key] = (w[key] * f).astype(np.float32)
        out = HipEncoder(cfg, w, max_batch=2, l2_normalize=True)(x).cpu().numpy()
        assert np.abs(out - g["desc_l2"]).max() < DESC_L2_ATOL            # invisible to the maximum bound
        with pytest.raises(AssertionError, match="mean"):
            parity_bounds.check("vit/vit_b16_224", out, g["desc_l2"])


def test_encoder_batching_is_invisible(dev):
    """Ragged batch (n not a multiple of max_batch) and different max_batch give
    bit-identical descriptors: frames are independent."""
    cfg, w, enc3 = _encoder("tiny", 3, max_batch=3, l2_normalize=True, lanes=2)   # 4 chunks over 2 streams
    from vsc_hip.encoder import HipEncoder
    enc7 = HipEncoder(cfg, w, max_batch=7, l2_normalize=True)
    x = torch.from_numpy(synth.frames(5, 11, cfg)).to(dev)
    a, b = enc3(x), enc7(x)
    assert torch.equal(a, b)
    assert torch.equal(enc7(x[4:5]), a[4:5])
    assert enc3(x[:0]).shape == (0, cfg.desc_dim)


def test_encoder_vs_oracle_fresh_inputs(dev):
    from oracle import vit_oracle
    cfg, w, enc = _encoder("vit_b16_224", 21, max_batch=8, l2_normalize=True)
    x = torch.from_numpy(synth.frames(22, 6, cfg))
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x).numpy()
    out = enc(x.to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=DESC_L2_ATOL)
    parity_bounds.check("vit/fresh", out, ref)
    # cosine between HIP and reference descriptors of the same frame
    assert ((out * ref).sum(1) > 0.9999).all()


def test_encoder_wide_model_vs_oracle(dev):
    """Width 1280 (ViT-H-like, two layers): the final LayerNorm + GeM takes its 8-waves-per-frame variant (width > 1024), the
    head reads 1280 inputs per output, K = 1280 / 5120 GEMMs run the one-tile kernel; 11 frames = three head groups, ragged."""
    from dataclasses import replace
    from oracle import vit_oracle
    from vsc_hip.encoder import HipEncoder
    cfg = replace(get_config("tiny"), name="wide", width=1280, heads=20, mlp_dim=5120, out_dim=512)
    w = synth.encoder_weights(31, cfg)
    enc = HipEncoder(cfg, w, max_batch=4, l2_normalize=True)
    x = torch.from_numpy(synth.frames(32, 11, cfg))
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x).numpy()
    out = enc(x.to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=DESC_L2_ATOL)


def test_encoder_rejects_wrong_input(dev):
    from vsc_hip import _lib
    cfg, w, enc = _encoder("tiny", 3, max_batch=2)
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 3, 32, 32, device=dev))
    with pytest.raises(_lib.HipPathUnavailable):
        enc(torch.zeros(1, 3, 64, 64))
    w2 = dict(w)
    del w2["blocks.1.fc2.bias"]
    from vsc_hip.encoder import HipEncoder
    with pytest.raises(KeyError):
        HipEncoder(cfg, w2)


@pytest.mark.parametrize("preset,n", [("vit_b32_384", 3), ("clip_vit_l14_224", 2)])
def test_other_reference_backbones_full_size(dev, preset, n):
    """The reference's other ViT towers at full size against the fp32 oracle: timm ViT-B/32-384
    (vit_v68 backbone: 145 tokens, eps 1e-6, pooled feature) and the CLIP ViT-L/14 video-score tower
    (257 tokens, patch 14 -> K padded 588->640, ln_pre, QuickGELU, CLS readout)."""
    from oracle import vit_oracle
    cfg, w, enc = _encoder(preset, 31, max_batch=2, l2_normalize=True)
    x = torch.from_numpy(synth.frames(32, n, cfg))
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x).numpy()
    out = enc(x.to(dev)).cpu().numpy()
    assert out.shape == (n, cfg.width)
    np.testing.assert_allclose(out, ref, rtol=0, atol=DESC_L2_ATOL)
    assert ((out * ref).sum(1) > 0.9999).all()


@pytest.mark.parametrize("preset,n", [("tiny_sscd", 5), ("vit_v68", 3)])
def test_sscd_head_model(dev, preset, n):
    """vit_v68 = timm ViT-B/32-384 + the SSCD head (Conv1d 768->2048 over tokens, GeM, Linear
    2048->512; train/train_v68/vsc/baseline/model_factory/backbones/sscd.py:25-42,88-94), the model
    infer/infer_ref.sh actually runs.  The conv output is rounded to bf16 before the cube."""
    from oracle import vit_oracle
    cfg, w, enc = _encoder(preset, 41, max_batch=2, l2_normalize=True)
    x = torch.from_numpy(synth.frames(42, n, cfg))
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x).numpy()
    out = enc(x.to(dev)).cpu().numpy()
    assert out.shape == (n, cfg.out_dim)
    np.testing.assert_allclose(out, ref, rtol=0, atol=DESC_L2_ATOL)


@pytest.mark.parametrize("preset", ["tiny", "tiny_clip", "vit_b16_224"])
def test_layernorm_folding_matches_golden_and_the_unfolded_path(dev, preset, golden_dir):
    """fuse_ln=1 (LN2 / next LN1 inside the fc1 / qkv epilogues, statistics from the proj / fc2 write-out) against the
    golden descriptors and against fuse_ln=0 (the separate LayerNorm passes, the default) on the same frames."""
    g = np.load(os.path.join(golden_dir, f"vit_{preset}.npz"))
    cfg = get_config(preset)
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    fused = _encoder(preset, int(g["weights_seed"]), max_batch=4, l2_normalize=True, fuse_ln=1)[2]
    plain = _encoder(preset, int(g["weights_seed"]), max_batch=4, l2_normalize=True, fuse_ln=0)[2]
    df, tf = fused(x, return_tokens=True)
    dp, tp = plain(x, return_tokens=True)
    df, dp, tf, tp = (t.cpu().numpy() for t in (df, dp, tf, tp))
    np.testing.assert_allclose(df, g["desc_l2"], rtol=0, atol=DESC_L2_ATOL)
    assert np.abs(df - dp).max() < 5e-4            # two bf16 pipelines with different rounding points
    assert np.abs(tf - tp).max() < 0.08 and np.abs(tf - tp).mean() < 0.01
    assert not np.array_equal(df, dp)              # the folded path really ran


def test_layernorm_folding_ragged_rows(dev):
    """A fold on ragged chunks (rows not a multiple of the 256-row tile, last chunk shorter) matches the default path."""
    cfg, w, plain = _encoder("tiny", 5, max_batch=37, l2_normalize=True)
    fused = _encoder("tiny", 5, max_batch=37, l2_normalize=True, fuse_ln=1)[2]
    x = torch.from_numpy(synth.frames(6, 50, cfg)).to(dev)
    dp, df = plain(x).cpu().numpy(), fused(x).cpu().numpy()
    assert np.abs(df - dp).max() < 5e-4 and not np.array_equal(df, dp)


def _u8_and_reference_tensor(seed, n, size, mean, std):
    """Random decoded frames uint8 [n,H,W,3] and the fp32 tensor torchvision's ToTensor + Normalize builds from them."""
    u8 = torch.from_numpy((synth.uniform(seed, (n, size, size, 3), 0.0, 256.0)).astype(np.uint8))
    x = u8.permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    return u8, x


@pytest.mark.parametrize("preset,mean,std", [("tiny", (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
                                             ("tiny_clip", (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)),
                                             ("vit_b16_224", (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))])
def test_uint8_frames_give_bit_identical_descriptors(dev, preset, mean, std):
    """Decoded uint8 HWC frames with the normalisation fused into patchify == the fp32 tensor path, bit for bit."""
    cfg, w, _ = _encoder(preset, 3, max_batch=4)
    from vsc_hip.encoder import HipEncoder
    enc = HipEncoder(cfg, w, max_batch=4, l2_normalize=True, u8_mean=mean, u8_std=std)
    u8, x = _u8_and_reference_tensor(17, 6, cfg.image_size, mean, std)
    a, b = enc(u8.to(dev)).cpu().numpy(), enc(x.to(dev)).cpu().numpy()
    assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        enc(u8.permute(0, 3, 1, 2).contiguous().to(dev))   # uint8 must be HWC


def test_layernorm_folding_in_the_benchmarked_configuration(dev):
    """fuse_ln=1 at 332-frame chunks on two lanes: the folding epilogues of the PERSISTENT GEMMs (gemm_bf16_v4_kernel with
    RESADD_STATS / LNF write-outs; small batches run them on the one-tile kernels) -- against the unfolded path on every frame,
    against the fp32 oracle on sampled frames, and run to run."""
    from oracle import vit_oracle
    cfg, w, plain = _encoder("vit_b16_224", 21, max_batch=332, l2_normalize=True, lanes=2)
    fused = _encoder("vit_b16_224", 21, max_batch=332, l2_normalize=True, lanes=2, fuse_ln=1)[2]
    n = 700
    x = torch.from_numpy(synth.frames(23, n, cfg))
    xd = x.to(dev)
    df, dp = fused(xd).cpu().numpy(), plain(xd).cpu().numpy()
    assert np.isfinite(df).all()
    # two bf16 pipelines with different rounding points: the maximum over 700 x 512 values (6e-4 measured; 3.8e-4 over the six
    # frames of the small test); the oracle comparison below holds the descriptor tolerance
    assert np.abs(df - dp).max() < 1e-3 and np.abs(df - dp).mean() < 1e-4 and not np.array_equal(df, dp)
    sample = [0, 1, 331, 332, 500, 663, 664, 699]
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    np.testing.assert_allclose(df[sample], ref, rtol=0, atol=DESC_L2_ATOL)
    for _ in range(3):
        assert np.array_equal(fused(xd).cpu().numpy(), df)


def test_encoder_in_the_benchmarked_configuration_vs_oracle(dev):
    """The configuration bench.py times -- 332-frame chunks (256 row tiles of 256 -> persistent gemm_bf16_v4, skewed
    attention residents), two lanes -- on 700 frames (332 + 332 + a ragged 36-frame chunk): sampled frames against the
    fp32 oracle at the descriptor tolerance, and every frame against the max_batch = 8 path (one-tile kernels, one lane),
    which the golden vectors hold.  infer/src/extractor.py:23 runs the same model whatever the batch."""
    from oracle import vit_oracle
    from vsc_hip.encoder import HipEncoder
    cfg, w, big = _encoder("vit_b16_224", 21, max_batch=332, l2_normalize=True, lanes=2)
    small = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, lanes=1)
    n = 700
    x = torch.from_numpy(synth.frames(23, n, cfg))
    xd = x.to(dev)
    out_big = big(xd).cpu().numpy()
    out_small = small(xd).cpu().numpy()
    assert np.isfinite(out_big).all()
    assert np.abs(out_big - out_small).max() < 2e-4
    sample = [0, 1, 331, 332, 500, 663, 664, 699]        # first / last rows of each chunk, both lanes, the ragged chunk
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    np.testing.assert_allclose(out_big[sample], ref, rtol=0, atol=DESC_L2_ATOL)
    parity_bounds.check("vit/benchmarked", out_big[sample], ref)
    assert ((out_big[sample] * ref).sum(1) > 0.9999).all()
    # a second call on the same encoder (workspaces and lanes reused) returns the same bits
    assert np.array_equal(big(xd).cpu().numpy(), out_big)
