"""Swin-V2 on the HIP path: kernels alone against plain torch fp32 statements of the same op, and the
whole encoder against the golden vectors (transformers.Swinv2Model) and the fp32 oracle.
Tolerance on L2-normalised descriptors: 1e-3 absolute (north_star)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import parity_bounds
from oracle import swin_oracle
from tools import synth
from vsc_hip.swin_config import get_swin_config

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_swin_switches():
    """a test that fails between set_option(..., "1") and its reset must not leave the switch on for the tests behind it"""
    yield
    from vsc_hip import _lib
    for name in ("VSC_SWIN_MLP512", "VSC_SWIN_QKV512", "VSC_SWIN_FUSED_MLP"):
        _lib.set_option(name, None)


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


def _rand(seed, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, shape, std))


@pytest.mark.parametrize("frames,res,window,shift,heads", [(2, 32, 16, 0, 2), (2, 32, 16, 8, 2), (1, 16, 16, 0, 4),
                                                           (3, 16, 8, 4, 2), (2, 8, 8, 0, 4), (1, 64, 16, 8, 4),
                                                           # windows of BASELINE.json configs[4] (Swin-V2-L at 384): 24, clipped to 12
                                                           (2, 48, 24, 12, 2), (1, 48, 24, 0, 3), (1, 24, 24, 0, 4), (2, 12, 12, 0, 3),
                                                           (1, 24, 12, 6, 2)])
def test_window_attention(dev, frames, res, window, shift, heads):
    from vsc_hip import ops
    c, n = heads * 32, window * window
    qkv = _rand(res + shift, (frames * res * res, 3 * c)).to(torch.bfloat16)
    table = 16 * torch.sigmoid(_rand(7, (heads, (2 * window - 1) ** 2)))            # compact table the kernel takes
    bias = table[:, swin_oracle.relative_position_index(window).reshape(-1)].reshape(heads, n, n)
    scale = torch.exp(torch.clamp(math.log(10.0) + _rand(8, (heads,), 0.4), max=math.log(100.0)))
    out = ops.window_attention_bf16(qkv.to(dev), table.to(dev), scale.to(dev), frames, res, window, shift, heads)
    # torch statement of torch2scripts.py:147-187 + :272-296 on the same bf16 qkv
    x = qkv.float().reshape(frames, res, res, 3 * c)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = swin_oracle._windows(x, res, window)
    q, k, v = xw.reshape(-1, n, 3, heads, 32).permute(2, 0, 3, 1, 4)
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1) * scale.reshape(1, heads, 1, 1) + bias[None]
    if shift:
        m = swin_oracle.shift_mask(res, window, shift)
        attn = (attn.reshape(frames, -1, heads, n, n) + m[None, :, None]).reshape(-1, heads, n, n)
    o = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(-1, n, c)
    o = swin_oracle._unwindows(o, res, window, frames)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    ref = o.reshape(frames * res * res, c)
    got = out.float().cpu()
    # q-hat / k-hat / P rounded to bf16, bf16 output
    torch.testing.assert_close(got, ref, rtol=2 ** -6, atol=2e-2)
    assert (got - ref).abs().mean() < 4e-3
    # bounded form: the heads' upper bounds folded into their tables, no row maximum in the kernel -- the same softmax.  One
    # head keeps a scale too large for the bound (it stays on the row maximum inside the same launch).
    big = scale.clone()
    big[0] = 60.0
    for sc_ in (scale, big):
        a = ops.window_attention_bf16(qkv.to(dev), table.to(dev), sc_.to(dev), frames, res, window, shift, heads).float().cpu()
        b = ops.window_attention_bf16(qkv.to(dev), table.to(dev), sc_.to(dev), frames, res, window, shift, heads, bounded=True).float().cpu()
        torch.testing.assert_close(b, a, rtol=2 ** -6, atol=1e-2)
        assert (a - b).abs().mean() < 2e-3
    torch.testing.assert_close(ops.window_attention_bf16(qkv.to(dev), table.to(dev), scale.to(dev), frames, res, window, shift, heads,
                                                         bounded=True).float().cpu(), ref, rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("rows,width", [(7, 64), (1000, 128), (33, 1024)])
def test_ln_residual(dev, rows, width):
    from vsc_hip import ops
    t, x0 = _rand(1, (rows, width), 2.0) + 0.3, _rand(2, (rows, width))
    g, b = 0.3 + _rand(3, (width,), 0.05), _rand(4, (width,), 0.05)
    ref = x0 + F.layer_norm(t, (width,), g, b, 1e-5)
    x, xb = ops.ln_residual(t.to(dev), g.to(dev), b.to(dev), 1e-5, x_in=x0.to(dev))
    torch.testing.assert_close(x.cpu(), ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))
    x2, _ = ops.ln_residual(t.to(dev), g.to(dev), b.to(dev), 1e-5)
    torch.testing.assert_close(x2.cpu(), F.layer_norm(t, (width,), g, b, 1e-5), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,n,k", [(5, 128, 64), (1000, 128, 128), (777, 256, 256), (300, 256, 1024), (129, 512, 512),
                                   (2048, 512, 2048),
                                   # more 256 x 256 tiles than CUs: the persistent kernel's LN_RES write-out -- one tile per row
                                   # (N = 256), and tile pairs exchanging their row statistics through L2 (N = 512; ragged M)
                                   (70000, 256, 256), (66000, 256, 1024), (65536, 512, 512), (67507, 512, 2048), (131072, 512, 128)])
def test_gemm_ln_matches_torch(dev, m, n, k):
    """Row-owning GEMM with the res-post-norm epilogue vs fp32 torch on the same bf16 operands."""
    from vsc_hip import ops, _lib
    if m > 60000:
        _lib.set_option("VSC_GEMM_LN_V4", "1")    # the persistent kernel on every shape it supports (default: long K only)
    a = _rand(11, (m, k)).to(torch.bfloat16)
    w = _rand(12, (n, k), k ** -0.5).to(torch.bfloat16)
    bias, x0 = _rand(13, (n,), 0.2), _rand(14, (m, n))
    g, b = 0.3 + _rand(15, (n,), 0.05), _rand(16, (n,), 0.05)
    t = a.float() @ w.float().T + bias
    ref = x0 + F.layer_norm(t, (n,), g, b, 1e-5)
    x, xb = ops.gemm_ln_bf16(a.to(dev), w.to(dev), bias.to(dev), g.to(dev), b.to(dev), 1e-5, x_in=x0.to(dev))
    torch.testing.assert_close(x.cpu(), ref, rtol=1e-4, atol=1e-4)
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))
    # no bias, no residual (the PatchMerging form), in place on x is also what the encoder does
    x2, _ = ops.gemm_ln_bf16(a.to(dev), w.to(dev), None, g.to(dev), b.to(dev), 1e-5)
    torch.testing.assert_close(x2.cpu(), F.layer_norm(a.float() @ w.float().T, (n,), g, b, 1e-5), rtol=1e-4, atol=1e-4)
    _lib.set_option("VSC_GEMM_LN_V4", None)


@pytest.mark.parametrize("m,c", [(5, 128), (1000, 128), (4096, 128), (70001, 128), (3, 256), (777, 256), (40000, 256),
                                 # c = 512: the one-wave-per-SIMD kernel (csrc/swin_mlp512.hip): one ragged tile, whole tiles, a ragged last tile of
                                 # every wave position, more tiles than CUs (stage 2 of Swin-V2-B at 256 frames is 65 536 rows)
                                 (5, 512), (128, 512), (1000, 512), (33 * 128 + 17, 512), (65536 + 77, 512)])
def test_swin_mlp_matches_torch(dev, m, c):
    """The fused MLP kernel (both Linears, GELU, LayerNorm, residual, shadow) vs fp32 torch on the same bf16 operands with
    the same rounding point for the hidden activations; ragged last row tiles, more row tiles than CUs."""
    from vsc_hip import ops
    x0 = _rand(21, (m, c))
    w1, b1 = _rand(22, (4 * c, c), c ** -0.5), _rand(23, (4 * c,), 0.2)
    w2, b2 = _rand(24, (c, 4 * c), (4 * c) ** -0.5), _rand(25, (c,), 0.2)
    g, b = 0.3 + _rand(26, (c,), 0.05), _rand(27, (c,), 0.05)
    xb = x0.to(torch.bfloat16).float()
    h = F.gelu(xb @ w1.to(torch.bfloat16).float().T + b1).to(torch.bfloat16).float()
    ref = x0 + F.layer_norm(h @ w2.to(torch.bfloat16).float().T + b2, (c,), g, b, 1e-5)
    x, xb_out = ops.swin_mlp_bf16(x0.to(dev), w1, b1, w2, b2, g, b, 1e-5)
    # rounding flips of single hidden activations (polynomial GELU, summation order) move a row's LayerNorm input by ~1e-3
    torch.testing.assert_close(x.cpu(), ref, rtol=0, atol=2e-3)
    assert (x.cpu() - ref).abs().mean() < 2e-4
    assert torch.equal(xb_out.cpu(), x.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("m,c", [(777, 128), (3, 256), (5000, 256), (5, 512), (128, 512), (1000, 512), (33 * 128 + 17, 512), (65536 + 77, 512)])
def test_swin_proj_mlp_matches_torch(dev, m, c):
    """proj + LayerNorm + residual + MLP + LayerNorm + residual in one launch (vsc_swin_proj_mlp_bf16; at c = 512 variant 1 of the
    generated kernel body: the projection on GEMM 1's machinery, x1 once through memory as fp32, its shadow in registers only) vs
    fp32 torch on the same bf16 operands with the same rounding points."""
    from vsc_hip import ops
    x0, att = _rand(31, (m, c)), _rand(32, (m, c)).to(torch.bfloat16)
    wp, bp = _rand(33, (c, c), c ** -0.5), _rand(34, (c,), 0.2)
    g1, be1 = 0.3 + _rand(35, (c,), 0.05), _rand(36, (c,), 0.05)
    w1, b1 = _rand(22, (4 * c, c), c ** -0.5), _rand(23, (4 * c,), 0.2)
    w2, b2 = _rand(24, (c, 4 * c), (4 * c) ** -0.5), _rand(25, (c,), 0.2)
    g2, be2 = 0.3 + _rand(26, (c,), 0.05), _rand(27, (c,), 0.05)
    x1 = x0 + F.layer_norm(att.float() @ wp.to(torch.bfloat16).float().T + bp, (c,), g1, be1, 1e-5)
    h = F.gelu(x1.to(torch.bfloat16).float() @ w1.to(torch.bfloat16).float().T + b1).to(torch.bfloat16).float()
    ref = x1 + F.layer_norm(h @ w2.to(torch.bfloat16).float().T + b2, (c,), g2, be2, 1e-5)
    x, xb = ops.swin_proj_mlp_bf16(x0.to(dev), att, wp, bp, g1, be1, w1, b1, w2, b2, g2, be2, 1e-5)
    torch.testing.assert_close(x.cpu(), ref, rtol=0, atol=3e-3)       # (one more bf16 rounding of x1 feeds the MLP than in the MLP-only test)
    assert (x.cpu() - ref).abs().mean() < 3e-4
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("m", [5, 128, 1000, 33 * 128 + 17, 65536 + 77])
def test_swin_proj_mlp_qkv_equals_proj_mlp_then_gemm(dev, m):
    """The next block's qkv Linear in the fused kernel's epilogue (vsc_swin_proj_mlp_qkv_bf16, variant 9 of the generated body) vs
    the two launches it replaces: x bit for bit the proj+MLP kernel's; qkv = vsc_gemm_bf16 on that kernel's shadow up to bf16
    rounding flips (same bf16 operands, fp32 accumulation from zero, bias added last) -- and against fp32 torch."""
    from vsc_hip import ops
    c = 512
    x0, att = _rand(31, (m, c)), _rand(32, (m, c)).to(torch.bfloat16)
    wp, bp = _rand(33, (c, c), c ** -0.5), _rand(34, (c,), 0.2)
    g1, be1 = 0.3 + _rand(35, (c,), 0.05), _rand(36, (c,), 0.05)
    w1, b1 = _rand(22, (4 * c, c), c ** -0.5), _rand(23, (4 * c,), 0.2)
    w2, b2 = _rand(24, (c, 4 * c), (4 * c) ** -0.5), _rand(25, (c,), 0.2)
    g2, be2 = 0.3 + _rand(26, (c,), 0.05), _rand(27, (c,), 0.05)
    wq, bq = _rand(37, (3 * c, c), c ** -0.5), _rand(38, (3 * c,), 0.2)
    x_ref, xb_ref = ops.swin_proj_mlp_bf16(x0.to(dev), att, wp, bp, g1, be1, w1, b1, w2, b2, g2, be2, 1e-5)
    x, qkv = ops.swin_proj_mlp_qkv_bf16(x0.to(dev), att, wp, bp, g1, be1, w1, b1, w2, b2, g2, be2, wq, bq, 1e-5)
    assert torch.equal(x, x_ref)
    t = xb_ref.float().cpu() @ wq.to(torch.bfloat16).float().T + bq
    torch.testing.assert_close(qkv.float().cpu(), t, rtol=2 ** -7, atol=2e-3)
    # the stand-alone GEMM sums K in another tile order: same values up to rounding flips of the bf16 result
    g = ops.gemm_bf16(xb_ref, wq.to(dev).to(torch.bfloat16), bq.to(dev))
    assert float((qkv != g).float().mean()) < 0.02
    torch.testing.assert_close(qkv.float(), g.float(), rtol=2 ** -7, atol=1e-5)
    assert abs(float((qkv.float() - g.float()).mean())) < 1e-5


def test_swin_mlp_rejects_other_widths(dev):
    from vsc_hip import ops
    from vsc_hip._lib import VscHipError
    c = 64
    with pytest.raises(VscHipError, match="unsupported"):
        ops.swin_mlp_bf16(torch.zeros(4, c, device=dev), torch.zeros(4 * c, c), torch.zeros(4 * c), torch.zeros(c, 4 * c),
                          torch.zeros(c), torch.ones(c), torch.zeros(c), 1e-5)


def test_swin_fused_mlp_equals_two_gemm_path(dev):
    """Swin-V2-B with the MLPs of stages 0-2 fused (default) and as fc1 + fc2 launches (VSC_SWIN_FUSED_MLP=0; VSC_SWIN_MLP512=0 for
    the 512-wide stage alone): same descriptors to bf16 rounding flips, and the switches are really taken (profile classes)."""
    from vsc_hip import _lib
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_weights(9, cfg)
    x = torch.from_numpy(synth.swin_frames(10, 6, cfg)).to(dev)
    enc = SwinHipEncoder(cfg, w, max_batch=4, l2_normalize=True)
    # (4-frame chunks are 8 tiles of the 512-wide stage's fused kernel: by default such calls keep the GEMM launches there -- the kernel
    # pays from ~0.6 of a round of 256 tiles on; VSC_SWIN_MLP512=1 takes it at every size)
    enc.set_profiling(True)
    enc(x)
    assert enc.profile()["s2.fc1"][1] == 2 * cfg.depths[2] and "s1.fc1" not in enc.profile()
    _lib.set_option("VSC_SWIN_MLP512", "1")
    enc.set_profiling(True)
    fused = enc(x).cpu().numpy()
    prof = enc.profile()
    # stages 0-2 (widths 128, 256, 512) run one kernel per MLP, booked under fc2_ln; the 1024-wide last stage keeps its two GEMMs
    assert all(f"s{s}.fc1" not in prof for s in range(3)) and prof["s3.fc1"][1] > 0
    assert prof["s0.fc2_ln"][1] == 2 * cfg.depths[0] and prof["s2.fc2_ln"][1] == 2 * cfg.depths[2] and "s2.proj_ln" not in prof
    _lib.set_option("VSC_SWIN_FUSED_MLP", "0")
    try:
        enc.set_profiling(True)
        plain = enc(x).cpu().numpy()
        prof = enc.profile()
        assert all(prof[f"s{s}.fc1"][1] == 2 * cfg.depths[s] for s in range(3))
    finally:
        _lib.set_option("VSC_SWIN_FUSED_MLP", None)
        enc.set_profiling(False)
    assert np.abs(fused - plain).max() < 4e-4
    # the 512-wide stage alone on the two-GEMM path (VSC_SWIN_MLP512=0): 18 of the 24 blocks change kernels
    _lib.set_option("VSC_SWIN_MLP512", "0")
    try:
        enc.set_profiling(True)
        plain512 = enc(x).cpu().numpy()
        prof = enc.profile()
        assert "s0.fc1" not in prof and prof["s2.fc1"][1] == 2 * cfg.depths[2]
    finally:
        _lib.set_option("VSC_SWIN_MLP512", "1")
        enc.set_profiling(False)
    assert np.abs(fused - plain512).max() < 4e-4 and not np.array_equal(fused, plain512)
    # the 512-wide stage with every block launching its own qkv GEMM (VSC_SWIN_QKV512=0; default: blocks 1..17 get their qkv
    # from the previous block's kernel, only the first one launches the GEMM -- per chunk)
    assert prof["s2.qkv"][1] == 2 * cfg.depths[2]
    enc.set_profiling(True)
    enc(x)
    assert enc.profile()["s2.qkv"][1] == 2 and enc.profile()["s1.qkv"][1] == 2 * cfg.depths[1]
    _lib.set_option("VSC_SWIN_QKV512", "0")
    try:
        enc.set_profiling(True)
        own_qkv = enc(x).cpu().numpy()
        assert enc.profile()["s2.qkv"][1] == 2 * cfg.depths[2]
    finally:
        _lib.set_option("VSC_SWIN_QKV512", None)
        _lib.set_option("VSC_SWIN_MLP512", None)
        enc.set_profiling(False)
    assert np.abs(fused - own_qkv).max() < 4e-4


def test_swin_encoder_is_deterministic_at_full_chunks(dev):
    """Run-to-run bit equality of Swin-V2-B at 256-frame chunks (fused MLP, fused PatchMerging, bounded softmax): the fused MLP once
    published its weight chunks with a bare s_barrier -- no wait for the LDS-DMA that fills them -- and corrupted a few frames in
    every other run (tools/micro/swin_determinism.py)."""
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    enc = SwinHipEncoder(cfg, synth.swin_weights(9, cfg), max_batch=256, l2_normalize=True)
    x = torch.from_numpy(synth.swin_frames(10, 300, cfg)).to(dev)
    first = enc(x).clone()
    for _ in range(5):
        assert torch.equal(enc(x), first)


@pytest.mark.parametrize("preset,frames", [("tiny_swin", 5), ("swinv2_base_256", 3)])
def test_swin_patch_merging_inside_the_gemm_is_bit_identical(dev, preset, frames):
    """PatchMerging's 2 x 2 gather done by the reduction GEMM's operand staging (default) against the gather kernel + GEMM
    (VSC_SWIN_FUSED_MERGE=0): the same values reach the same MFMAs in the same order -> identical descriptors; and the
    buffer swap behind the fused form survives repeated calls."""
    from vsc_hip import _lib
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config(preset)
    enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=2, l2_normalize=True)
    x = torch.from_numpy(synth.swin_frames(6, frames, cfg)).to(dev)
    fused = enc(x).clone()
    assert torch.equal(enc(x), fused)
    _lib.set_option("VSC_SWIN_FUSED_MERGE", "0")
    try:
        plain = enc(x).clone()
    finally:
        _lib.set_option("VSC_SWIN_FUSED_MERGE", None)
    assert torch.equal(fused.view(torch.int32), plain.view(torch.int32))
    assert torch.equal(enc(x), fused)


def test_gemm_ln_rejects_other_widths(dev):
    from vsc_hip import ops
    from vsc_hip._lib import VscHipError
    a, w = torch.zeros(4, 64, dtype=torch.bfloat16, device=dev), torch.zeros(96, 64, dtype=torch.bfloat16, device=dev)
    g = torch.ones(96, device=dev)
    with pytest.raises(VscHipError, match="unsupported"):
        ops.gemm_ln_bf16(a, w, None, g, g, 1e-5)


def test_merge_gather_bit_exact(dev):
    from vsc_hip import ops
    frames, res, c = 3, 8, 64
    xb = _rand(5, (frames * res * res, c)).to(torch.bfloat16)
    g = xb.reshape(frames, res, res, c)
    ref = torch.cat([g[:, 0::2, 0::2], g[:, 1::2, 0::2], g[:, 0::2, 1::2], g[:, 1::2, 1::2]], -1).reshape(-1, 4 * c)
    out = ops.merge_gather_bf16(xb.to(dev), frames, res).cpu()
    assert torch.equal(out.view(torch.int16), ref.contiguous().view(torch.int16))


@pytest.mark.parametrize("preset", ["tiny_swin", "tiny_swin_w8", "swinv2_base_256", "tiny_swin_w24", "swinv2_large_384"])
def test_swin_encoder_matches_golden(dev, preset, golden_dir):
    from vsc_hip.swin_encoder import SwinHipEncoder
    g = np.load(os.path.join(golden_dir, f"swin_{preset}.npz"))
    cfg = get_swin_config(preset)
    w = synth.swin_weights(int(g["weights_seed"]), cfg)
    x = torch.from_numpy(synth.swin_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    enc = SwinHipEncoder(cfg, w, max_batch=2)
    desc, tok = enc(x, return_tokens=True)
    desc, tok = desc.cpu().numpy(), tok.cpu().numpy()
    assert np.abs(tok[:, :4] - g["tokens_head"]).max() < 0.1
    assert np.abs(tok[:, :4] - g["tokens_head"]).mean() < 0.012
    assert np.abs(desc - g["desc"]).max() < 0.02 * np.abs(g["desc"]).max()
    d2 = SwinHipEncoder(cfg, w, max_batch=3, l2_normalize=True)(x).cpu().numpy()
    np.testing.assert_allclose(d2, g["desc_l2"], rtol=0, atol=1e-3)
    parity_bounds.check(f"swin/{preset}", d2, g["desc_l2"])     # mean |d| and |mean d| (tests/parity_bounds.py)
    np.testing.assert_allclose(np.linalg.norm(d2, axis=1), 1.0, atol=1e-5)
    if 512 in [cfg.dim(s) for s in range(cfg.stages)]:
        # chunks this small take the GEMM launches in the 512-wide stage (its fused kernel would leave most CUs idle): the same fixture
        # through the fused kernel as well (VSC_SWIN_MLP512=1: at every size)
        from vsc_hip import _lib
        _lib.set_option("VSC_SWIN_MLP512", "1")
        try:
            d3 = SwinHipEncoder(cfg, w, max_batch=3, l2_normalize=True)(x).cpu().numpy()
        finally:
            _lib.set_option("VSC_SWIN_MLP512", None)
        assert not np.array_equal(d3, d2)
        np.testing.assert_allclose(d3, g["desc_l2"], rtol=0, atol=1e-3)
        parity_bounds.check(f"swin/{preset}", d3, g["desc_l2"])


def test_swin_encoder_matches_golden_on_frames_that_differ(dev, golden_dir):
    """Swin-V2-B's second fixture (= the reference's own class, check_golden_against_reference.py): frames of different structure
    (synth.structured_frames), descriptors with cosine 0.79 .. 0.9 to one another instead of 0.98 -- through the GEMM launches of the
    512-wide stage (6 frames: the default) and through its fused kernel."""
    from vsc_hip import _lib
    from vsc_hip.swin_encoder import SwinHipEncoder
    g = np.load(os.path.join(golden_dir, "swin_swinv2_base_256_structured.npz"))
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_weights(int(g["weights_seed"]), cfg)
    x = torch.from_numpy(synth.structured_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    ref = g["desc_l2"]
    cos = (ref @ ref.T)[np.triu_indices(len(ref), 1)]
    assert cos.min() < 0.85
    for forced in (None, "1"):
        _lib.set_option("VSC_SWIN_MLP512", forced)
        d = SwinHipEncoder(cfg, w, max_batch=4, l2_normalize=True)(x).cpu().numpy()
        np.testing.assert_allclose(d, ref, rtol=0, atol=1e-3)
        parity_bounds.check("swin/swinv2_base_256_structured", d, ref)


def test_swin_encoder_vs_oracle_and_batching(dev):
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("tiny_swin")
    w = synth.swin_weights(9, cfg)
    x = torch.from_numpy(synth.swin_frames(10, 7, cfg))
    with torch.no_grad():
        ref = swin_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x).numpy()
    a = SwinHipEncoder(cfg, w, max_batch=3, l2_normalize=True)(x.to(dev))
    b = SwinHipEncoder(cfg, w, max_batch=7, l2_normalize=True)(x.to(dev))
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=0, atol=1e-3)
    assert torch.equal(a, b)
    w2 = dict(w)
    del w2["layers.1.blocks.0.attn.logit_scale"]
    with pytest.raises(KeyError):
        SwinHipEncoder(cfg, w2)


@pytest.mark.parametrize("precision,bound", [("bf16", 1e-3), ("fp16", 2e-4)])
def test_swin_outlier_fixture(dev, precision, bound, golden_dir):
    """Weights with what trained checkpoints have and random init lacks (tools/synth.swin_outlier_weights: LayerNorm gains x 20 in a few
    channels, a residual channel at ~100 through all 18 blocks of the deepest stage, every other block's heads at the logit-scale clamp
    of 100, hidden units deep in GELU's linear range); golden = the reference's own SwinTransformerV2 (check_golden_against_reference.
    swin_outlier, max |fixture - class| 0.0).  Both operand types, both forms of the 512-wide stage."""
    from vsc_hip import _lib
    from vsc_hip.swin_encoder import SwinHipEncoder
    g = np.load(f"{golden_dir}/swin_swinv2_base_256_outlier.npz")
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_outlier_weights(int(g["weights_seed"]), cfg)
    x = torch.from_numpy(synth.structured_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    for fused in ("1", "0"):
        with _lib.option("VSC_SWIN_MLP512", fused):
            enc = SwinHipEncoder(cfg, w, max_batch=8, l2_normalize=True, precision=precision)
            d = enc(x).cpu().numpy()
            enc.close()
        assert np.isfinite(d).all()
        err = np.abs(d - g["desc_l2"])
        print(f"{precision} operands, outlier fixture (VSC_SWIN_MLP512={fused}): max {err.max():.2e} mean {err.mean():.2e}")
        assert err.max() <= bound, (precision, fused, float(err.max()))


@pytest.mark.parametrize("precision,bound", [("bf16", 3e-4), ("fp16", 6e-5)])
def test_a_frame_alone_and_inside_a_full_chunk(dev, precision, bound):
    """The kernel path of the 512-wide stage is chosen per chunk by its row count (swin_encoder.hip: fills512): the same 8 frames
    encoded alone (GEMM launches) and as rows of a 256-frame call (fused kernels, the next block's qkv inside) agree to rounding-order
    noise, which is what src/extractor.py documents -- grouping loader batches into 2 048-frame calls does not move a descriptor by
    more than this."""
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=256, l2_normalize=True, precision=precision)
    few = torch.from_numpy(synth.structured_frames(31, 8, cfg))
    rest = torch.from_numpy(synth.swin_frames(32, 8, cfg)).repeat(31, 1, 1, 1)
    alone = enc(few.to(dev)).cpu()
    inside = enc(torch.cat([rest[:100], few, rest[100:]]).to(dev)).cpu()[100:108]
    err = float((alone - inside).abs().max())
    print(f"{precision}: the same frames alone / inside a 256-frame call: max |d| {err:.2e}")
    assert 0 < err <= bound or err == 0.0
    enc.close()


def test_swin_uint8_frames_bit_identical(dev):
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("tiny_swin")
    enc = SwinHipEncoder(cfg, synth.swin_weights(4, cfg), max_batch=3, l2_normalize=True)
    u8 = torch.from_numpy((synth.uniform(19, (5, cfg.image_size, cfg.image_size, 3), 0.0, 256.0)).astype(np.uint8))
    x = (u8.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
    assert np.array_equal(enc(u8.to(dev)).cpu().numpy(), enc(x.to(dev)).cpu().numpy())


def test_swin_encoder_in_the_benchmarked_configuration_vs_oracle(dev):
    """Swin-V2-B as bench.py runs it -- 256-frame chunks on two lanes (persistent GEMMs, full gemm_ln grids) -- on 520
    frames (256 + 256 + a ragged 8-frame chunk): sampled frames against the fp32 oracle, every frame against the
    max_batch = 4 path that the golden vectors hold."""
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_weights(9, cfg)
    n = 520
    x = torch.from_numpy(synth.swin_frames(10, n, cfg))
    xd = x.to(dev)
    big = SwinHipEncoder(cfg, w, max_batch=256, l2_normalize=True)
    small = SwinHipEncoder(cfg, w, max_batch=4, l2_normalize=True)
    out_big = big(xd).cpu().numpy()
    out_small = small(xd).cpu().numpy()
    assert np.isfinite(out_big).all()
    # two bf16 pipelines with different fp32 summation orders (persistent GEMMs, the pair-exchange LayerNorm of the long-K
    # gemm_ln launches): rounding flips of the bf16 activations, observed 2.3e-4
    assert np.abs(out_big - out_small).max() < 4e-4
    sample = [0, 255, 256, 400, 511, 512, 519]
    with torch.no_grad():
        ref = swin_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    np.testing.assert_allclose(out_big[sample], ref, rtol=0, atol=1e-3)
    parity_bounds.check("swin/benchmarked", out_big[sample], ref)
    assert np.array_equal(big(xd).cpu().numpy(), out_big)


def test_swin_profiling_classes(dev):
    """vsc_swin_set_profiling / get_profile: every kernel class of every stage reports its launches, results are unchanged,
    and a multi-chunk call while profiling runs its chunks back to back (no lanes)."""
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("tiny_swin")
    enc = SwinHipEncoder(cfg, synth.swin_weights(4, cfg), max_batch=3, l2_normalize=True)
    x = torch.from_numpy(synth.swin_frames(6, 7, cfg)).to(dev)
    ref = enc(x)
    enc.set_profiling(True)
    out = enc(x)
    prof = enc.profile()
    enc.set_profiling(False)
    assert torch.equal(out, ref)
    chunks = 3
    assert prof["patchify"][1] == chunks and prof["pool_head"][1] == chunks
    for s in range(cfg.stages):
        fused_mlp = cfg.dim(s) in (128, 256, 512)   # one kernel for the whole MLP, booked under fc2_ln
        fused_proj = fused_mlp                      # ... with proj + LayerNorm in front of it as well
        for kind in ("qkv", "attention", "proj_ln", "fc1", "fc2_ln"):
            ms, n = prof.get(f"s{s}.{kind}", (0.0, 0))
            if (kind == "fc1" and fused_mlp) or (kind == "proj_ln" and fused_proj):
                assert n == 0
            elif kind == "qkv" and cfg.dim(s) == 512:
                assert n == chunks and ms > 0     # the first block's only: the others' qkv comes out of the previous block's kernel
            else:
                assert n == chunks * cfg.depths[s] and ms > 0
        if s + 1 < cfg.stages:
            assert prof[f"s{s}.merge"][1] == chunks
    assert torch.equal(enc(x), ref)
