"""The fp16-operand build of the library (libvsc_hip_f16.so, csrc/common.h "the encoders' 16-bit operand type"): the same kernels with
IEEE fp16 instead of bf16 as the MFMA operand type.  Same fixtures as the bf16 tests (tests/golden: outputs of the reference's own
classes), tighter bounds: measured maxima 3e-5 .. 9e-5 (profiles/r06_operand_precision_probe.txt) against 2.4e-4 .. 8.8e-4 with bf16.
Kernel-level cases go through the C ABI with torch.float16 tensors (`ops.operands("fp16")`)."""
import math

import numpy as np
import pytest
import torch

from tools import synth

pytestmark = pytest.mark.gpu

DESC_ATOL_FP16 = 2.0e-4     # maximum over a fixture; measured <= 1.34e-4 (tiny_swin_w8), 1.0e-4 (Swin-V2-B on the structured frames)
MEAN_ATOL_FP16 = 4.5e-5     # mean |d|; measured <= 3.4e-5 (tiny_swin_w8: the PV product of the window attention is bf16 in both builds)


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    lib = _lib.require_device("fp16")
    assert lib.vsc_operand_dtype() == b"fp16"
    return torch.device("cuda:0")


def _l2(x):
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _check(name, got, want):
    d = np.abs(got - want)
    print(f"fp16 operands, {name}: max {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() <= DESC_ATOL_FP16, (name, float(d.max()))
    assert d.mean() <= MEAN_ATOL_FP16, (name, float(d.mean()))


@pytest.mark.parametrize("preset", ["tiny", "tiny_clip", "vit_b16_224", "vit_v68"])
def test_vit_matches_golden_with_fp16_operands(dev, preset, golden_dir):
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    g = np.load(f"{golden_dir}/vit_{preset}.npz")
    cfg = get_config(preset)
    w = synth.encoder_weights(int(g["weights_seed"]), cfg)
    enc = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, precision="fp16")
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    _check(f"vit/{preset}", enc(x).cpu().numpy(), g["desc_l2"])
    if preset == "vit_b16_224":
        gs = np.load(f"{golden_dir}/vit_vit_b16_224_structured.npz")
        xs = torch.from_numpy(synth.structured_frames(int(gs["frames_seed"]), int(gs["n_frames"]), cfg)).to(dev)
        _check("vit/vit_b16_224 structured", enc(xs).cpu().numpy(), gs["desc_l2"])
        # uint8 frames: the patchify kernel's ToTensor + Normalize, then the same network
        u8 = ((xs.permute(0, 2, 3, 1) * 0.5 + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()
        f32 = ((u8.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2).contiguous()
        assert torch.equal(enc(u8), enc(f32))
    enc.close()


def test_vit_layernorm_folding_with_fp16_operands(dev, golden_dir):
    """fuse_ln = 1 (LayerNorm folded into the neighbouring GEMM epilogues, DESIGN 4.1: W' = lp(gamma * W), column sums of the ROUNDED weights
    on the host -- `bf16_round` there is the build's operand type) against the same golden fixture."""
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    g = np.load(f"{golden_dir}/vit_vit_b16_224.npz")
    cfg = get_config("vit_b16_224")
    w = synth.encoder_weights(int(g["weights_seed"]), cfg)
    enc = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, precision="fp16", fuse_ln=1)
    x = torch.from_numpy(synth.frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    _check("vit/vit_b16_224 fuse_ln", enc(x).cpu().numpy(), g["desc_l2"])
    enc.close()


@pytest.mark.parametrize("preset", ["tiny_swin", "tiny_swin_w8", "tiny_swin_w24", "swinv2_base_256", "swinv2_large_384"])
def test_swin_matches_golden_with_fp16_operands(dev, preset, golden_dir):
    """Every window-attention kernel (16 x 16 plain / streamed / shifted-streamed, 8 x 8, 24 x 24 and 12 x 12 wide forms), the fused
    MLP kernels of all three widths and the LayerNorm write-outs, on the fixtures that equal the reference's SwinTransformerV2."""
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    g = np.load(f"{golden_dir}/swin_{preset}.npz")
    cfg = get_swin_config(preset)
    w = synth.swin_weights(int(g["weights_seed"]), cfg)
    enc = SwinHipEncoder(cfg, w, max_batch=8, l2_normalize=True, precision="fp16")
    x = torch.from_numpy(synth.swin_frames(int(g["frames_seed"]), int(g["n_frames"]), cfg)).to(dev)
    _check(f"swin/{preset}", enc(x).cpu().numpy(), _l2(g["desc"]))
    if preset == "swinv2_base_256":
        gs = np.load(f"{golden_dir}/swin_swinv2_base_256_structured.npz")
        xs = torch.from_numpy(synth.structured_frames(int(gs["frames_seed"]), int(gs["n_frames"]), cfg)).to(dev)
        _check("swin/swinv2_base_256 structured", enc(xs).cpu().numpy(), gs["desc_l2"])
    enc.close()


def test_vit_fp16_operands_in_the_benchmarked_configuration(dev):
    """332-frame chunks on two lanes through the persistent GEMMs: equal to the small-batch path (golden-held above) within rounding-order
    noise, a sample against the fp32 oracle, and the same bits on a second pass."""
    from oracle import vit_oracle
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    cfg = get_config("vit_b16_224")
    w = synth.encoder_weights(7, cfg)
    n = 700
    base = torch.from_numpy(synth.structured_frames(21, 28, cfg))
    x = base.repeat(25, 1, 1, 1)[:n] * torch.linspace(0.6, 1.0, n).view(-1, 1, 1, 1)
    big = HipEncoder(cfg, w, max_batch=332, l2_normalize=True, lanes=2, precision="fp16")
    out1 = big(x.to(dev)).cpu()
    out2 = big(x.to(dev)).cpu()
    assert torch.equal(out1, out2), "fp16-operand ViT step is not deterministic at full chunks"
    small = HipEncoder(cfg, w, max_batch=8, l2_normalize=True, lanes=1, precision="fp16")
    sample = [0, 1, 331, 332, 333, 663, 664, 699]
    got_small = small(x[sample].to(dev)).cpu()
    assert float((out1[sample] - got_small).abs().max()) <= 5e-5
    with torch.no_grad():
        ref = vit_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    _check("vit/benchmarked", out1[sample].numpy(), ref)
    assert torch.isfinite(out1).all()
    big.close()
    small.close()


def test_swin_fp16_operands_in_the_benchmarked_configuration(dev):
    """256-frame chunks: the fused stage kernels (generated-asm body with v_mfma_f32_16x16x32_f16 / v_cvt_pk_f16_f32) at full size."""
    from oracle import swin_oracle
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    w = synth.swin_weights(5, cfg)
    n = 520
    base = torch.from_numpy(synth.structured_frames(22, 26, cfg))
    x = base.repeat(20, 1, 1, 1)[:n] * torch.linspace(0.6, 1.0, n).view(-1, 1, 1, 1)
    big = SwinHipEncoder(cfg, w, max_batch=256, l2_normalize=True, precision="fp16")
    out1 = big(x.to(dev)).cpu()
    out2 = big(x.to(dev)).cpu()
    assert torch.equal(out1, out2), "fp16-operand Swin step is not deterministic at full chunks"
    sample = [0, 255, 256, 511, 512, 519]
    with torch.no_grad():
        ref = swin_oracle.descriptors({k: torch.from_numpy(v) for k, v in w.items()}, cfg, x[sample]).numpy()
    _check("swin/benchmarked", out1[sample].numpy(), _l2(ref))
    big.close()


def _rand(seed, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, shape, std))


@pytest.mark.parametrize("m,n,k", [(200, 132, 128), (5043, 2304, 768), (65404, 768, 768), (37, 512, 3072)])
def test_gemm_with_fp16_operands(dev, m, n, k):
    from vsc_hip import _lib, ops
    a, w, b = _rand(1, (m, k)).half(), _rand(2, (n, k), 0.05).half(), _rand(3, (n,))
    rows = torch.tensor(sorted({0, 1, m // 2, m - 1}))
    ref = a[rows].float() @ w.float().t() + b
    with ops.operands("fp16"):
        out = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_BF16)
        assert out.dtype == torch.float16
        gelu = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_GELU_BF16)
    # fp32 accumulation, one fp16 rounding of the result: 2^-11 relative
    torch.testing.assert_close(out[rows.to(dev)].float().cpu(), ref, rtol=2 ** -10, atol=3e-4)
    torch.testing.assert_close(gelu[rows.to(dev)].float().cpu(), torch.nn.functional.gelu(ref), rtol=2 ** -10, atol=3e-4)


def test_vit_attention_with_fp16_operands(dev):
    from vsc_hip import ops
    frames, tokens, heads = 3, 197, 12
    qkv = _rand(5, (frames * tokens, 3 * heads * 64)).half()
    with ops.operands("fp16"):
        out = ops.attention_bf16(qkv.to(dev), frames, tokens, heads).float().cpu()
    q, k, v = qkv.float().view(frames, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).permute(0, 2, 1, 3).reshape(frames * tokens, heads * 64)
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-3)    # probabilities and context rounded to fp16 (bf16 build: 1.5e-2)


@pytest.mark.parametrize("res,window,shift,heads", [(32, 16, 0, 4), (32, 16, 8, 4), (16, 8, 4, 8), (24, 24, 0, 6), (24, 12, 6, 6)])
def test_window_attention_with_fp16_operands(dev, res, window, shift, heads):
    """Against the fp32 statement of WindowAttention.forward (torch2scripts.py:147-187, :272-296) that tests/test_gpu_swin.py uses;
    logit scales up to 60 (span 120 + the bias range: far past anything a bound-shifted softmax could hold in fp16's exponent range)."""
    import torch.nn.functional as F
    from oracle import swin_oracle
    from vsc_hip import ops
    frames, c, n = 2, heads * 32, window * window
    qkv = _rand(9, (frames * res * res, 3 * c)).half()
    table = 16 * torch.sigmoid(_rand(10, (heads, (2 * window - 1) ** 2)))
    bias = table[:, swin_oracle.relative_position_index(window).reshape(-1)].reshape(heads, n, n)
    scale = torch.linspace(8.0, 60.0, heads)
    with ops.operands("fp16"):
        out = ops.window_attention_bf16(qkv.to(dev), table.to(dev), scale.to(dev), frames, res, window, shift, heads).float().cpu()
        # the bounded softmax (probabilities down to e^-69) in the fp16 build: P and V go through the PV MFMA as bf16 in both builds
        small = torch.linspace(6.0, 14.0, heads)
        a = ops.window_attention_bf16(qkv.to(dev), table.to(dev), small.to(dev), frames, res, window, shift, heads).float().cpu()
        b = ops.window_attention_bf16(qkv.to(dev), table.to(dev), small.to(dev), frames, res, window, shift, heads, bounded=True).float().cpu()
        assert torch.isfinite(b).all()
        torch.testing.assert_close(b, a, rtol=2 ** -6, atol=1e-2)
        assert (a - b).abs().mean() < 2e-3
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
    assert torch.isfinite(out).all()
    # q-hat / k-hat rounded to fp16 (at scale 60 a 2^-11 error of a cosine is 0.03 in the logit), P and V to bf16, fp16 output.  The bf16
    # build's test holds 2e-2 / mean 4e-3 at scales around 10.
    torch.testing.assert_close(out, ref, rtol=2 ** -6, atol=2e-2)
    assert (out - ref).abs().mean() < 2e-3


def test_fp16_overflow_saturates_instead_of_inf(dev):
    """A value past fp16's range at a rounding point becomes +-65504, not inf (MODE.FP16_OVFL, csrc/common.h lp_kernel_entry): GEMM
    write-outs (plain and GELU), the LayerNorm output, and a whole encoder whose fc1 bias drives hidden units to 1e6 -- finite
    descriptors where inf would turn the next LayerNorm into NaN."""
    from vsc_hip import _lib, ops
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder
    a = torch.full((300, 256), 30.0).half()
    w = torch.full((128, 256), 10.0).half()
    w[1] = -10.0
    with ops.operands("fp16"):
        out = ops.gemm_bf16(a.to(dev), w.to(dev), None, epilogue=_lib.EPI_BF16).float().cpu()       # 256 * 300 = 76 800 > 65 504
        gel = ops.gemm_bf16(a.to(dev), w.to(dev), None, epilogue=_lib.EPI_GELU_BF16).float().cpu()
        big = torch.zeros(4, 768)
        big[:, 0] = 1.0
        ln = ops.layernorm(big.to(dev), torch.full((768,), 5000.0).to(dev), torch.zeros(768).to(dev), 1e-6).float().cpu()   # 27.7 * 5000
    assert torch.isfinite(out).all() and float(out[0, 0]) == 65504.0 and float(out[0, 1]) == -65504.0
    assert torch.isfinite(gel).all() and float(gel[0, 0]) == 65504.0 and abs(float(gel[0, 1])) < 0.5     # (the GELU polynomial is off by 3.5e-6 |x| at x = -76 800)
    assert torch.isfinite(ln).all() and float(ln.max()) == 65504.0
    cfg = get_config("tiny")
    wts = {k: v.copy() for k, v in synth.encoder_weights(7, cfg).items()}
    wts["blocks.0.fc1.bias"][:8] = 1.0e6
    x = torch.from_numpy(synth.frames(11, 3, cfg)).to(dev)
    d = HipEncoder(cfg, wts, max_batch=4, l2_normalize=True, precision="fp16")(x)
    assert torch.isfinite(d).all()
