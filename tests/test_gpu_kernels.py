"""Per-kernel parity on a real MI355X, through the C ABI (vsc_hip.ops -> libvsc_hip.so).

Floating-point kernels are compared with a plain PyTorch fp32 statement of the same
op on the same (bf16-rounded) inputs; tolerances are written next to each check.
"""
import math

import numpy as np
import pytest
import torch

from tools import synth
from vsc_hip import _lib as _vsc_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()  # fails loudly if the HIP path cannot run
    return torch.device("cuda:0")


def _rand(seed, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, shape, std))


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (200, 132, 128), (1000, 768, 768),
                                   (5043, 2304, 768), (37, 512, 3072), (1, 4, 64),
                                   # the short-K tile choices of launch_v2_pick: D (N % 256 = 128), C, B (N <= 128), ragged M
                                   (1500, 384, 128), (2049, 512, 256), (1300, 128, 512), (1025, 1032, 192)])
def test_gemm_bf16_store(dev, m, n, k):
    from vsc_hip import ops, _lib
    a = _rand(1, (m, k)).to(torch.bfloat16)
    w = _rand(2, (n, k), 0.05).to(torch.bfloat16)
    b = _rand(3, (n,))
    ref = a.float() @ w.float().t() + b
    out = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_BF16).float().cpu()
    # fp32 accumulation, one bf16 rounding of the result: |err| <= 2^-8 |ref| + accumulation noise
    torch.testing.assert_close(out, ref, rtol=2 ** -7, atol=2e-3)


def test_gemm_is_not_transposed(dev):
    """A = identity-like selector, asymmetric W: catches row/col swaps in the epilogue."""
    from vsc_hip import ops, _lib
    m = n = 192
    k = 256
    a = torch.zeros(m, k)
    a[torch.arange(m), torch.arange(m)] = 1.0
    w = (torch.arange(n).float()[:, None] * 0.5 + torch.arange(k).float()[None, :] * 0.001953125)
    out = ops.gemm_bf16(a.to(dev), w.to(dev), None, epilogue=_lib.EPI_BF16).float().cpu()
    ref = a.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t()
    torch.testing.assert_close(out, ref.to(torch.bfloat16).float(), rtol=0, atol=0)


@pytest.mark.parametrize("epi", ["gelu", "qgelu"])
@pytest.mark.parametrize("m,n,k", [(300, 512, 128), (1333, 512, 128), (1100, 3072, 768)])
def test_gemm_activation_epilogues(dev, epi, m, n, k):
    from vsc_hip import ops, _lib
    a = _rand(4, (m, k)).to(torch.bfloat16)
    w = _rand(5, (n, k), 0.1).to(torch.bfloat16)
    b = _rand(6, (n,), 0.1)
    z = a.float() @ w.float().t() + b
    if epi == "gelu":
        ref, code = torch.nn.functional.gelu(z), _lib.EPI_GELU_BF16
    else:
        ref, code = z * torch.sigmoid(1.702 * z), _lib.EPI_QGELU_BF16
    out = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=code).float().cpu()
    torch.testing.assert_close(out, ref, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("m,n,k", [(517, 768, 3072), (1300, 768, 768), (1100, 512, 256)])
def test_gemm_residual_epilogue_in_place(dev, m, n, k):
    from vsc_hip import ops, _lib
    a = _rand(7, (m, k)).to(torch.bfloat16)
    w = _rand(8, (n, k), 0.02).to(torch.bfloat16)
    b = _rand(9, (n,), 0.1)
    res = _rand(10, (m, n))
    ref = res + a.float() @ w.float().t() + b
    x = res.clone().to(dev)
    out = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_RESADD_F32, aux=x, out=x)
    assert out.data_ptr() == x.data_ptr()
    # fp32 out; only the accumulation order differs
    torch.testing.assert_close(x.cpu(), ref, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("epi", ["bf16", "gelu", "qgelu", "resadd", "f32"])
@pytest.mark.parametrize("m,n,k", [(23040 + 77, 768, 256),      # 273 tiles on 256 CUs: some workgroups take two, ragged M
                                   (30001, 1000, 384),           # ragged N tile (n % 256 = 232), 6 K-tiles
                                   (18000, 2304, 1152),          # 639 tiles, several N-groups, 18 K-tiles
                                   (65404, 768, 768),            # the encoder's proj shape at B = 332: 3 whole rounds
                                   (70000, 512, 128)])           # K = 128: two K-tiles per output tile, every one of them stages the next tile (Swin stage-1 shapes)
def test_gemm_persistent_kernel_equals_one_tile_per_workgroup(dev, epi, m, n, k):
    """v4 (persistent: the operand ring streams across output tiles, write-out through 4 KiB per wave) runs the same
    MFMA chain and the same epilogue arithmetic as v3 -> identical bits; v3 itself is checked against fp32 above."""
    import os
    from vsc_hip import ops, _lib
    kind = {"bf16": _lib.EPI_BF16, "gelu": _lib.EPI_GELU_BF16, "qgelu": _lib.EPI_QGELU_BF16, "resadd": _lib.EPI_RESADD_F32,
            "f32": _lib.EPI_F32}[epi]
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    x = torch.randn(m, n, generator=g).to(dev) if epi == "resadd" else None
    outs = {}
    for v4 in ("0", "1", "1"):      # twice through v4: the second launch finds the ring / bias rows of the first in LDS
        _vsc_lib.set_option("VSC_GEMM_V4", v4)
        try:
            xx = None if x is None else x.clone()
            outs[v4] = ops.gemm_bf16(a, w, b, epilogue=kind, aux=xx, out=xx).clone()
        finally:
            _vsc_lib.set_option("VSC_GEMM_V4", None)
        torch.cuda.synchronize()
        if v4 == "1":
            bits = torch.int32 if outs["0"].dtype == torch.float32 else torch.int16
            assert torch.equal(outs["0"].view(bits), outs["1"].view(bits))
    # and against fp32 on a slice (rows that straddle tile boundaries, the last row)
    rows = torch.tensor([0, 255, 256, 257, m // 2, m - 1], device=dev)
    ref = a[rows].float() @ w.float().t() + b
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif epi == "qgelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    elif epi == "resadd":
        ref = ref + x[rows]
    torch.testing.assert_close(outs["1"][rows].float(), ref, rtol=2 ** -7, atol=3e-3)


def test_gemm_persistent_kernel_without_bias(dev):
    """No bias vector: the persistent kernel's per-tile bias row comes from a zero-extent buffer descriptor (LDS-DMA zero fill)."""
    from vsc_hip import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(3)
    m, n, k = 25000, 1024, 384
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    out = ops.gemm_bf16(a, w, None, epilogue=_lib.EPI_BF16)
    rows = torch.tensor([0, 1, 255, 256, 12345, m - 1], device=dev)
    torch.testing.assert_close(out[rows].float(), a[rows].float() @ w.float().t(), rtol=2 ** -7, atol=3e-3)
    x = torch.randn(m, n, generator=g).to(dev)
    y = ops.gemm_bf16(a, w, None, epilogue=_lib.EPI_RESADD_F32, aux=x.clone(), out=None)
    torch.testing.assert_close(y[rows], x[rows] + a[rows].float() @ w.float().t(), rtol=1e-4, atol=2e-3)


def test_gemm_patch_epilogue(dev):
    from vsc_hip import ops, _lib
    frames, tokens, n, k = 3, 17, 128, 768
    a = _rand(11, (frames * (tokens - 1), k)).to(torch.bfloat16)
    w = _rand(12, (n, k), 0.03).to(torch.bfloat16)
    b = _rand(13, (n,), 0.1)
    pos = _rand(14, (tokens, n), 0.3)
    z = (a.float() @ w.float().t() + b).reshape(frames, tokens - 1, n) + pos[1:]
    out = ops.gemm_bf16(a.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_PATCH_F32, aux=pos.to(dev),
                        tokens=tokens).cpu().reshape(frames, tokens, n)
    torch.testing.assert_close(out[:, 1:], z, rtol=1e-5, atol=2e-4)
    assert torch.count_nonzero(out[:, 0]) == 0  # CLS rows are not this kernel's to write


def test_gemm_rejects_bad_k(dev):
    from vsc_hip import ops, _lib
    with pytest.raises(_lib.VscHipError, match="multiple of 64"):
        ops.gemm_bf16(torch.zeros(4, 100, device=dev), torch.zeros(8, 100, device=dev))


@pytest.mark.parametrize("frames,tokens,heads", [(2, 17, 2), (3, 197, 12), (1, 145, 12), (2, 257, 16),
                                                 (1, 1, 1), (2, 32, 3)])
def test_attention(dev, frames, tokens, heads):
    from vsc_hip import ops
    d = heads * 64
    qkv = _rand(20 + tokens, (frames * tokens, 3 * d))
    qkv[:, : 2 * d] *= 2.0  # peaky-ish softmax
    qkv = qkv.to(torch.bfloat16)
    q, k, v = qkv.float().reshape(frames, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / math.sqrt(64)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(frames * tokens, d)
    out = ops.attention_bf16(qkv.to(dev), frames, tokens, heads).float().cpu()
    # P is rounded to bf16 before PV and the output is bf16: 2^-8 relative on O(1) values
    torch.testing.assert_close(out, ref, rtol=2 ** -6, atol=1e-2)
    assert (out - ref).abs().mean() < 2e-3


def test_attention_spiked_key(dev):
    """One key dominates one query (forces a large max); padded keys must stay masked."""
    from vsc_hip import ops
    frames, tokens, heads = 1, 197, 1
    qkv = _rand(31, (tokens, 192), 0.3)
    qkv[5, :64] = 6.0
    qkv[100, 64:128] = 6.0  # q5 . k100 = 64*36/8 = 288 -> softmax one-hot on key 100
    qkv = qkv.to(torch.bfloat16)
    out = ops.attention_bf16(qkv.to(dev), frames, tokens, heads).float().cpu()
    torch.testing.assert_close(out[5], qkv[100, 128:].float(), rtol=2 ** -7, atol=1e-3)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("rows,width", [(5, 128), (1000, 768), (3, 1024), (7, 2048)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_layernorm(dev, rows, width, out_f32):
    from vsc_hip import ops
    x = _rand(40, (rows, width), 2.0) + 0.5
    g = 1.0 + _rand(41, (width,), 0.1)
    b = _rand(42, (width,), 0.1)
    for eps in (1e-12, 1e-6):
        ref = torch.nn.functional.layer_norm(x, (width,), g, b, eps)
        out = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), eps, out_f32).float().cpu()
        if out_f32:
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        else:
            torch.testing.assert_close(out, ref.to(torch.bfloat16).float(), rtol=2 ** -7, atol=1e-5)


@pytest.mark.parametrize("preset", ["tiny", "vit_b16_224", "clip_vit_l14_224"])
def test_patchify_bit_exact(dev, preset):
    from oracle import vit_oracle
    from vsc_hip import ops
    from vsc_hip.config import get_config
    cfg = get_config(preset)
    x = torch.from_numpy(synth.frames(50, 2, cfg))
    kpad = (cfg.patch_dim + 63) // 64 * 64
    out = ops.patchify_bf16(x.to(dev), cfg.patch_size, kpad).cpu()
    ref = vit_oracle.patchify(x, cfg.patch_size).reshape(-1, cfg.patch_dim).to(torch.bfloat16)
    assert torch.equal(out[:, : cfg.patch_dim].view(torch.int16), ref.view(torch.int16))
    assert torch.count_nonzero(out[:, cfg.patch_dim:].float()) == 0


def test_l2_normalize(dev):
    from oracle import knn_oracle
    from vsc_hip import ops
    x = synth.normalish(60, (1001, 512))
    x[17] = 0.0
    out = ops.l2_normalize_(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = knn_oracle.l2_normalize(x)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)  # only the reduction order differs
    assert not out[17].any()


@pytest.mark.parametrize("frames,tokens,heads", [(3, 197, 12), (2, 145, 4), (5, 50, 2), (2, 256, 3)])
def test_attention_lds_dma_kernel_is_bit_identical(dev, frames, tokens, heads):
    """VSC_ATTN_DMA=1: the persistent attention kernel (K / V by LDS-DMA into a double buffer, V row-major read with
    ds_read_b64_tr_b16) against the default one: the same MFMAs on the same operands in the same order -> identical bits."""
    from vsc_hip import _lib, ops
    g = torch.Generator().manual_seed(frames * 1000 + tokens)
    qkv = (torch.randn(frames * tokens, 3 * heads * 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    ref = ops.attention_bf16(qkv, frames, tokens, heads).clone()
    with _lib.option("VSC_ATTN_DMA", "1"):
        for _ in range(3):
            assert torch.equal(ops.attention_bf16(qkv, frames, tokens, heads), ref)
