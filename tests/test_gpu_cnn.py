"""Matching-track networks on the HIP path (vsc_hip/cnn.py over vsc_conv2d_f32 & co) against the torch fp32 oracle
(oracle/cnn_oracle.py) on the same timm-named state dicts and similarity maps.  Tolerances: the convolutions run on
exact-fp32 MFMA chains, so layers agree with F.conv2d to summation-order rounding; 1e-3 on probabilities is the tier's
bound, the observed error is ~1e-5."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cnn_synth  # noqa: E402
from vsc_hip import _lib as _vsc_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,act,res", [
    (2, 9, 7, 3, 16, 3, 2, "hard_swish", False),     # the classifier's stem: K = 27 (padded to 32), ragged everything
    (1, 16, 16, 18, 18, 3, 1, "relu", True),         # an HRNet BasicBlock convolution with its residual
    (3, 5, 6, 64, 256, 1, 1, "relu", True),          # Bottleneck expansion, 2 channel tiles
    (2, 12, 10, 36, 72, 3, 2, None, False),          # a fuse-layer downsampling step
    (4, 1, 1, 576, 1024, 1, 1, "hard_swish", False),  # conv_head on pooled features (4 rows)
    (1, 20, 20, 334, 64, 1, 1, "relu", False),       # fuse.0
    (2, 7, 7, 24, 8, 5, 1, "hard_sigmoid", False),   # 5x5, K = 600
])
def test_conv2d_matches_torch(dev, n, h, w, cin, cout, k, stride, act, res):
    from vsc_hip import cnn
    rng = np.random.RandomState(n * 100 + cin)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, cin, h, w).astype(np.float32))
    want = F.conv2d(x, sd["c.weight"], sd["c.bias"], stride=stride, padding=k // 2)
    r = torch.from_numpy(rng.randn(*want.shape).astype(np.float32)) if res else None
    if res:
        want = want + r
    want = {"relu": F.relu, "hard_swish": F.hardswish, "hard_sigmoid": F.hardsigmoid, None: lambda v: v}[act](want)
    conv = cnn.Conv(sd, "c", None, stride, dev)
    got = conv(x.permute(0, 2, 3, 1).contiguous().to(dev), act=act,
               residual=None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev))
    assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    # channel window of a wider buffer (the concat path)
    wide = torch.full((n, want.shape[2], want.shape[3], cout + 5), 7.0, device=dev)
    conv(x.permute(0, 2, 3, 1).contiguous().to(dev), act=act, residual=None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
         out=wide, coff=3)
    assert torch.equal(wide[..., 3:3 + cout], got) and (wide[..., :3] == 7).all() and (wide[..., 3 + cout:] == 7).all()


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride", [
    (2, 16, 16, 20, 18, 3, 1),      # HRNet's high-resolution branch (18 channels carried as 20), 3 x 3, borders on all sides
    (3, 13, 11, 36, 72, 3, 2),      # a downsampling step: odd sizes, stride 2
    (1, 7, 9, 24, 8, 5, 1),         # 5 x 5, K = 600 (19 K-steps, the last one past K)
    (2, 33, 17, 64, 64, 3, 1),      # several pixel tiles, the last one ragged
    (5, 6, 6, 40, 80, 1, 2),        # 1 x 1 with a stride: not the in-place case
    (3, 40, 56, 20, 18, 3, 1),      # 6720 pixels: 27 pixel tiles walked persistently, images change inside tiles
    (1, 3, 2, 8, 16, 3, 1),         # an image smaller than the window on one axis: most taps outside on every pixel
    (2, 9, 9, 12, 24, 5, 2),        # 5 x 5 (25 taps), stride 2
])
def test_conv2d_implicit_gather_equals_materialised_patches(dev, n, h, w, cin, cout, k, stride):
    """The narrow convolution kernel gathers its patch operand from the feature map while staging it (no im2col matrix);
    it must put the same values into the same LDS image as the packed path -> identical bits.  The packed path is checked
    against torch above."""
    import os
    from vsc_hip import cnn
    rng = np.random.RandomState(cin * 7 + k)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    conv = cnn.Conv(sd, "c", None, stride, dev)
    outs = []
    for flag in ("0", "1"):
        _vsc_lib.set_option("VSC_CONV_IMPLICIT", flag)
        try:
            outs.append(conv(x, act="relu").clone())
        finally:
            _vsc_lib.set_option("VSC_CONV_IMPLICIT", None)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    want = F.relu(F.conv2d(x.cpu().permute(0, 3, 1, 2), sd["c.weight"], sd["c.bias"], stride=stride, padding=k // 2))
    assert torch.allclose(outs[1].cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("n,h,w,cin,cout,res", [
    (2, 16, 32, 20, 18, True),      # whole tiles: two tile rows of one 32-pixel tile column
    (3, 21, 45, 20, 20, True),      # ragged in both directions, images change between tiles
    (1, 5, 3, 20, 18, False),       # an image smaller than one tile (and than the halo)
    (16, 56, 56, 20, 18, True),     # 1568 tiles: every workgroup walks several (the halo double buffer)
    (2, 19, 40, 36, 18, False),     # 36 input channels (9 chunks per pixel, an odd chunk count: the zero weight chunk)
    (1, 8, 33, 20, 32, False),      # all 32 channel rows of the accumulator
    (2, 19, 40, 36, 36, True),      # HRNet's second branch: two accumulators per wave, 36 of their 64 rows exist
    (1, 9, 33, 20, 36, False),      # a transition: 20 -> 36 channels
    (16, 28, 28, 36, 36, True),     # several tiles per workgroup with the large halos (one workgroup per CU)
])
def test_conv2d_direct_3x3_matches_torch_and_the_gemm_path(dev, n, h, w, cin, cout, res):
    """conv3x3_direct_kernel (halo tile in LDS, nine taps read from it) against torch and against the implicit-GEMM kernel it
    replaces for the thin 3 x 3 layers (VSC_CONV_DIRECT=0): the same products in a different summation order."""
    from vsc_hip import cnn
    rng = np.random.RandomState(cin + h)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    got = conv(x, act="relu", residual=r).clone()   # 20 -> <= 20 channels: the split-bf16 kernel (conv3x3_direct_x3_kernel); else the fp32 one
    _vsc_lib.set_option("VSC_CONV_X3", "0")
    try:
        f32 = conv(x, act="relu", residual=r).clone()   # conv3x3_direct_kernel on the fp32 matrix pipe
    finally:
        _vsc_lib.set_option("VSC_CONV_X3", None)
    _vsc_lib.set_option("VSC_CONV_DIRECT", "0")
    try:
        gemm = conv(x, act="relu", residual=r).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_DIRECT", None)
    want = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), sd["c.weight"].double(), sd["c.bias"].double(), padding=1)
    if res:
        want = want + r.double().cpu().permute(0, 3, 1, 2)
    want = F.relu(want)
    assert torch.allclose(got.double().cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    assert torch.allclose(got, gemm, atol=5e-6, rtol=1e-5) and torch.allclose(f32, gemm, atol=5e-6, rtol=1e-5)
    # the split-bf16 products (x1 w1 + x1 w2 + x2 w1 + x1 w3 + x3 w1 + x2 w2, fp32 accumulation) are as close to float64 as the fp32 pipe's
    e_x3, e_f32 = float((got.double().cpu().permute(0, 3, 1, 2) - want).abs().max()), float((f32.double().cpu().permute(0, 3, 1, 2) - want).abs().max())
    assert e_x3 < 2.0 * e_f32 + 1e-6, (e_x3, e_f32)
    assert not torch.equal(got, torch.zeros_like(got))
    for _ in range(3):   # run-to-run bit equality (the halo tiles arrive by LDS-DMA: a missing wait shows up as flicker)
        assert torch.equal(conv(x, act="relu", residual=r), got)
    # a channel window of a wider buffer at an offset that is not a multiple of 4 (scalar stores), other activations
    for act in ("hard_swish", None):
        wide = torch.full((n, h, w, cout + 7), 3.0, device=dev)
        conv(x, act=act, residual=r, out=wide, coff=5)
        _vsc_lib.set_option("VSC_CONV_DIRECT", "0")
        try:
            ref = conv(x, act=act, residual=r).clone()
        finally:
            _vsc_lib.set_option("VSC_CONV_DIRECT", None)
        assert torch.allclose(wide[..., 5:5 + cout], ref, atol=5e-6, rtol=1e-5)
        assert (wide[..., :5] == 3).all() and (wide[..., 5 + cout:] == 3).all()


@pytest.mark.parametrize("n,h,w,cout,res,act", [
    (2, 128, 128, 256, True, "relu"),      # whole tiles, the Bottleneck's last convolution with its residual
    (3, 113, 97, 256, True, "relu"),       # 32 883 pixels: a ragged last tile (rows past the end are dropped by the descriptor)
    (2, 130, 127, 256, False, None),       # the downsample convolution: no residual, no activation
    (1, 200, 170, 128, True, "hard_swish"),   # four of the eight waves own channels
])
def test_conv2d_1x1_expansion_stream_kernel(dev, n, h, w, cout, res, act):
    """conv1x1_expand64_kernel (pixels as the MFMA row operand, weights in registers, input tiles by LDS-DMA) against torch
    and against the tile kernel it replaces for 64 -> 128..256 1 x 1 layers (VSC_CONV_EXPAND=0): same ascending-k chains per
    output, bias and residual added in the same order -> identical bits."""
    from vsc_hip import cnn
    rng = np.random.RandomState(cout + h)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, 64, 1, 1) / 8).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, 64).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    got = conv(x, act=act, residual=r).clone()
    _vsc_lib.set_option("VSC_CONV_EXPAND", "0")
    try:
        tile = conv(x, act=act, residual=r).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_EXPAND", None)
    want = F.conv2d(x.cpu().permute(0, 3, 1, 2), sd["c.weight"], sd["c.bias"])
    if res:
        want = want + r.cpu().permute(0, 3, 1, 2)
    want = {"relu": F.relu, "hard_swish": F.hardswish, None: lambda v: v}[act](want)
    assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    assert torch.equal(got.view(torch.int32), tile.view(torch.int32))
    for _ in range(3):
        assert torch.equal(conv(x, act=act, residual=r), got)
    wide = torch.full((n, h, w, cout + 8), 3.0, device=dev)   # a channel window of a wider buffer
    conv(x, act=act, residual=r, out=wide, coff=4)
    assert torch.equal(wide[..., 4:4 + cout], got) and (wide[..., :4] == 3).all() and (wide[..., 4 + cout:] == 3).all()


@pytest.mark.parametrize("n,h,w,cin,cout,res,act", [
    (48, 40, 40, 16, 72, False, "relu"),          # the classifier's first expansion: 3 channel blocks, the last one 8 wide
    (150, 20, 20, 24, 88, False, "relu"),         # cin % 8 == 4: a pixel's pad chunk reads the next pixel (zero weights)
    (400, 10, 10, 40, 240, False, "hard_swish"),
    (401, 10, 10, 48, 288, False, "hard_swish"),  # two channel groups of 256 (grid.y), ragged last tile
    (1400, 5, 5, 96, 576, False, "hard_swish"),   # three groups, 128-pixel tiles
    (350, 10, 10, 88, 176, True, None),           # with a residual
])
def test_conv2d_pointwise_stream_kernel(dev, n, h, w, cin, cout, res, act):
    """conv1x1_stream_kernel (any cin % 4 == 0 up to 96, all output channels of a pixel tile in one workgroup) against torch and,
    bit for bit, against the tile kernels (VSC_CONV_EXPAND=0)."""
    from vsc_hip import cnn
    rng = np.random.RandomState(cin + cout)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)),
          "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    got = conv(x, act=act, residual=r).clone()
    _vsc_lib.set_option("VSC_CONV_EXPAND", "0")
    try:
        tile = conv(x, act=act, residual=r).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_EXPAND", None)
    want = F.conv2d(x.cpu().permute(0, 3, 1, 2), sd["c.weight"], sd["c.bias"])
    if res:
        want = want + r.cpu().permute(0, 3, 1, 2)
    want = {"relu": F.relu, "hard_swish": F.hardswish, None: lambda v: v}[act](want)
    assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    assert torch.equal(got.view(torch.int32), tile.view(torch.int32))
    for _ in range(3):
        assert torch.equal(conv(x, act=act, residual=r), got)


@pytest.mark.parametrize("n,h,w,cout,stride,act", [
    (3, 160, 160, 16, 2, "hard_swish"),     # the classifier's stem at its resolution
    (2, 37, 53, 16, 2, "relu"),             # odd sizes: the last band of output rows is ragged
    (2, 18, 22, 32, 1, None),               # stride 1, eight channel quads
    (1, 5, 4, 8, 2, "relu"),                # an image smaller than one band
])
def test_conv2d_stem_direct_kernel(dev, n, h, w, cout, stride, act):
    """conv_stem3_kernel (3 input channels, 3 x 3; input rows in LDS, weights in registers) against torch and, bit for bit, against
    the patch-matrix GEMM path (VSC_CONV_STEM=0): the same fmaf chain in ascending k."""
    from vsc_hip import cnn
    rng = np.random.RandomState(h + cout)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, 3, 3, 3) / 5).astype(np.float32)), "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, 3).astype(np.float32)).to(dev)
    conv = cnn.Conv(sd, "c", None, stride, dev)
    got = conv(x, act=act).clone()
    _vsc_lib.set_option("VSC_CONV_STEM", "0")
    try:
        gemm = conv(x, act=act).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_STEM", None)
    want = F.conv2d(x.cpu().permute(0, 3, 1, 2), sd["c.weight"], sd["c.bias"], stride=stride, padding=1)
    want = {"relu": F.relu, "hard_swish": F.hardswish, None: lambda v: v}[act](want)
    assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    assert torch.equal(got.view(torch.int32), gemm.view(torch.int32))
    wide = torch.full(got.shape[:3] + (cout + 8,), 3.0, device=dev)
    conv(x, act=act, out=wide, coff=4)
    assert torch.equal(wide[..., 4:4 + cout], got) and (wide[..., :4] == 3).all() and (wide[..., 4 + cout:] == 3).all()


@pytest.mark.parametrize("n,h,w,cin,cout,res,act", [
    (4, 128, 128, 64, 64, False, "relu"),       # whole 4 x 32 tiles
    (3, 150, 171, 64, 48, True, "relu"),        # ragged tiles in both directions, 48 of the 64 weight rows exist, residual
    (5, 118, 113, 64, 64, True, None),
    (3, 150, 171, 256, 20, False, "relu"),      # HRNet's transition from the 256-wide stem: four 64-channel passes per tile
    (2, 224, 224, 256, 20, True, None),
])
def test_conv2d_3x3_tap_streaming_split_bf16_kernel(dev, n, h, w, cin, cout, res, act):
    """conv3x3_tap_x3_kernel (64 input channels: halo tile resident as three bf16 planes, weights streamed and split per tap) against
    float64 and against the fp32-pipe path (VSC_CONV_X3=0): as close to float64 as the fp32 pipe, run to run identical."""
    from vsc_hip import cnn
    rng = np.random.RandomState(h + cout)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)), "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    got = conv(x, act=act, residual=r).clone()
    _vsc_lib.set_option("VSC_CONV_X3", "0")
    try:
        f32 = conv(x, act=act, residual=r).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_X3", None)
    want = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), sd["c.weight"].double(), sd["c.bias"].double(), padding=1)
    if res:
        want = want + r.double().cpu().permute(0, 3, 1, 2)
    if act:
        want = F.relu(want)
    e_x3 = float((got.double().cpu().permute(0, 3, 1, 2) - want).abs().max())
    e_f32 = float((f32.double().cpu().permute(0, 3, 1, 2) - want).abs().max())
    assert e_x3 < 2e-5 and e_x3 < 2.0 * e_f32 + 1e-6, (e_x3, e_f32)
    assert not torch.equal(got, f32)                       # the split-bf16 kernel really ran
    for _ in range(3):
        assert torch.equal(conv(x, act=act, residual=r), got)
    wide = torch.full((n, h, w, cout + 8), 3.0, device=dev)
    conv(x, act=act, residual=r, out=wide, coff=4)
    assert torch.equal(wide[..., 4:4 + cout], got) and (wide[..., :4] == 3).all() and (wide[..., 4 + cout:] == 3).all()


@pytest.mark.parametrize("n,h,w,cin,cout,res,act", [
    (16, 56, 56, 72, 72, True, "relu"),       # HRNet's third branch
    (7, 61, 45, 72, 72, False, "relu"),       # ragged last pixel tile, images change inside tiles
    (16, 28, 28, 144, 144, True, "relu"),     # fourth branch: two channel groups of 80 rows
    (5, 47, 39, 144, 72, False, None),
])
def test_conv2d_3x3_split_bf16_implicit_gemm(dev, n, h, w, cin, cout, res, act):
    """conv_x3_gemm_kernel (both operands split into three bf16 planes once per call, implicit GEMM with LDS-DMA gathers, six MFMAs
    per block) against float64 and the fp32-pipe path (VSC_CONV_X3=0)."""
    from vsc_hip import cnn
    rng = np.random.RandomState(h + cout)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)), "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    got = conv(x, act=act, residual=r).clone()
    _vsc_lib.set_option("VSC_CONV_X3", "0")
    try:
        f32 = conv(x, act=act, residual=r).clone()
    finally:
        _vsc_lib.set_option("VSC_CONV_X3", None)
    want = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), sd["c.weight"].double(), sd["c.bias"].double(), padding=1)
    if res:
        want = want + r.double().cpu().permute(0, 3, 1, 2)
    if act:
        want = F.relu(want)
    e_x3 = float((got.double().cpu().permute(0, 3, 1, 2) - want).abs().max())
    e_f32 = float((f32.double().cpu().permute(0, 3, 1, 2) - want).abs().max())
    assert e_x3 < 2e-5 and e_x3 < 2.0 * e_f32 + 1e-6, (e_x3, e_f32)
    assert not torch.equal(got, f32)
    for _ in range(3):
        assert torch.equal(conv(x, act=act, residual=r), got)
    wide = torch.full((n, h, w, cout + 7), 3.0, device=dev)   # a window at an offset that is not a multiple of 4: scalar stores
    conv(x, act=act, residual=r, out=wide, coff=5)
    assert torch.equal(wide[..., 5:5 + cout], got) and (wide[..., :5] == 3).all() and (wide[..., 5 + cout:] == 3).all()


def test_depthwise_pool_scale_upsample(dev):
    from vsc_hip import cnn
    rng = np.random.RandomState(0)
    for c, k, stride in ((16, 3, 2), (96, 5, 1), (240, 5, 2), (18, 3, 1), (7, 5, 2)):   # the last two: c % 4 != 0 -> the scalar kernel
        sd = {"d.weight": torch.from_numpy(rng.randn(c, 1, k, k).astype(np.float32) * 0.2)}
        x = torch.from_numpy(rng.randn(2, c, 11, 9).astype(np.float32))
        want = F.hardswish(F.conv2d(x, sd["d.weight"], None, stride=stride, padding=k // 2, groups=c))
        got = cnn.DwConv(sd, "d", None, stride, dev)(x.permute(0, 2, 3, 1).contiguous().to(dev), act="hard_swish")
        assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=1e-5)
    # small maps (the classifier's 5 x 5 / 10 x 10 stages): dwconv_small_kernel (channel slabs, the image in LDS) against torch and,
    # bit for bit, against the row kernel (VSC_DWCONV_SMALL=0); 9 images per workgroup column, a partial last slab (240 = 3 x 64 + 48)
    for c, k, stride, hh, ww, nimg in ((576, 5, 1, 5, 5, 40), (240, 5, 1, 10, 10, 19), (288, 5, 2, 10, 10, 7), (16, 3, 1, 6, 7, 3), (120, 5, 1, 10, 10, 2100)):
        sd = {"d.weight": torch.from_numpy(rng.randn(c, 1, k, k).astype(np.float32) * 0.2), "d.bias": torch.from_numpy(rng.randn(c).astype(np.float32))}
        x = torch.from_numpy(rng.randn(nimg, hh, ww, c).astype(np.float32)).to(dev)
        dw = cnn.DwConv(sd, "d", None, stride, dev)
        got = dw(x, act="hard_swish").clone()
        _vsc_lib.set_option("VSC_DWCONV_SMALL", "0")
        try:
            rows = dw(x, act="hard_swish").clone()
        finally:
            _vsc_lib.set_option("VSC_DWCONV_SMALL", None)
        assert torch.equal(got.view(torch.int32), rows.view(torch.int32)), (c, k, stride)
        if nimg < 100:
            want = F.hardswish(F.conv2d(x.cpu().permute(0, 3, 1, 2), sd["d.weight"], sd["d.bias"], stride=stride, padding=k // 2, groups=c))
            assert torch.allclose(got.cpu().permute(0, 3, 1, 2), want, atol=1e-5)
    x = torch.from_numpy(rng.randn(3, 6, 5, 70).astype(np.float32)).to(dev)
    assert torch.allclose(cnn.avgpool(x).reshape(3, 70), x.mean((1, 2)), atol=1e-6)
    src = torch.from_numpy(rng.randn(2, 3, 4, 18).astype(np.float32)).to(dev)
    out = torch.from_numpy(rng.randn(2, 12, 16, 30).astype(np.float32)).to(dev)
    want = out.clone()
    want[..., 5:23] = F.relu(want[..., 5:23] + F.interpolate(src.permute(0, 3, 1, 2), scale_factor=4, mode="nearest").permute(0, 2, 3, 1))
    cnn.upsample_into(src, out, 4, 5, True, "relu")
    assert torch.equal(out, want)


def test_upsample_sum_is_the_chain_of_upsample_adds(dev):
    """vsc_upsample_sum_f32 (one pass per HRNet fuse node) adds its terms in the order of the separate passes: identical bits."""
    from vsc_hip import cnn
    rng = np.random.RandomState(5)
    n, h, w, c = 2, 16, 24, 20
    base = torch.from_numpy(rng.randn(n, h, w, c).astype(np.float32)).to(dev)
    terms = [(torch.from_numpy(rng.randn(n, h // f, w // f, c).astype(np.float32)).to(dev), f) for f in (1, 2, 8)]
    for nterms in (0, 1, 2, 3):
        for with_base in (True, False):
            if not with_base and nterms == 0:
                continue
            for act in ("relu", None):
                want = base.clone() if with_base else torch.zeros_like(base)
                for t, f in terms[:nterms]:
                    want = want + F.interpolate(t.permute(0, 3, 1, 2), scale_factor=f, mode="nearest").permute(0, 2, 3, 1)
                if act:
                    want = F.relu(want)
                got = cnn.upsample_sum(base if with_base else None, terms[:nterms], torch.empty_like(base), act)
                assert torch.equal(got, want), (nterms, with_base, act)
    inplace = base.clone()   # base aliasing the output
    cnn.upsample_sum(inplace, terms, inplace, "relu")
    assert torch.equal(inplace, cnn.upsample_sum(base, terms, torch.empty_like(base), "relu"))
    # the concat path (a channel window of a wider buffer) now runs on the same kernel
    wide = torch.full((n, h, w, 48), 2.0, device=dev)
    cnn.upsample_into(terms[1][0], wide, 2, 8, True, None)
    assert torch.equal(wide[..., 8:28], 2.0 + F.interpolate(terms[1][0].permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1))
    assert (wide[..., :8] == 2).all() and (wide[..., 28:] == 2).all()


def test_squeeze_excite_one_launch_equals_the_four_launch_form(dev):
    """vsc_se_block_f32 (image in LDS: mean, two Linears, scale, one read and one write of x) against avgpool + two convolutions +
    channel_scale on the same weights: same arithmetic in another summation order."""
    from vsc_hip import cnn
    rng = np.random.RandomState(9)
    for n, h, w, c, cr in ((37, 10, 10, 120, 32), (5, 10, 10, 144, 40), (3, 40, 40, 16, 8), (2, 7, 9, 96, 24)):
        sd = {"s.conv_reduce.weight": torch.from_numpy((rng.randn(cr, c, 1, 1) / np.sqrt(c)).astype(np.float32)),
              "s.conv_reduce.bias": torch.from_numpy(rng.randn(cr).astype(np.float32) * 0.1),
              "s.conv_expand.weight": torch.from_numpy((rng.randn(c, cr, 1, 1) / np.sqrt(cr)).astype(np.float32)),
              "s.conv_expand.bias": torch.from_numpy(rng.randn(c).astype(np.float32) * 0.1)}
        se = cnn.SqueezeExcite(sd, "s", dev)
        x = torch.from_numpy(rng.randn(n, h, w, c).astype(np.float32)).to(dev)
        got = se(x.clone())
        cnn.SqueezeExcite.FUSED = False
        try:
            ref = se(x.clone())
        finally:
            cnn.SqueezeExcite.FUSED = True
        xt = x.cpu().permute(0, 3, 1, 2)
        gate = F.hardsigmoid(F.conv2d(F.relu(F.conv2d(xt.mean((2, 3), keepdim=True), sd["s.conv_reduce.weight"], sd["s.conv_reduce.bias"])),
                                      sd["s.conv_expand.weight"], sd["s.conv_expand.bias"]))
        assert torch.allclose(got.cpu().permute(0, 3, 1, 2), xt * gate, atol=1e-5, rtol=1e-5), (c, cr)
        assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5) and not torch.equal(got, x)
        assert torch.equal(se(x.clone()), got)


def test_mobilenetv3_classifier_matches_oracle(dev):
    from oracle import cnn_oracle
    from vsc_hip import cnn
    sds = [cnn_synth.mobilenetv3_small_state(11), cnn_synth.mobilenetv3_small_state(12)]
    x = cnn_synth.similarity_maps(13, 6, 160, 160)          # MatchClassifyDataset resolution (infer_matching.py:159)
    with torch.no_grad():
        strip = [{k[len("model."):]: v for k, v in sd.items()} for sd in sds]
        want_logits = cnn_oracle.mobilenetv3_small(strip[0], x)
        want = cnn_oracle.match_classify_probability(strip, x)
    models = [cnn.MobileNetV3SmallHip(sd, dev) for sd in sds]
    got_logits = models[0](x.to(dev)).cpu()
    got = cnn.match_classify_probability(models, x.to(dev)).cpu()
    assert got_logits.shape == (6, 2)
    assert float((got_logits - want_logits).abs().max()) < 1e-3 * max(1.0, float(want_logits.abs().max()))
    assert float((got - want).abs().max()) < 1e-4            # probabilities (tier bound 1e-3)


@pytest.mark.parametrize("n,h,w", [(2, 32, 48), (1, 224, 224)])   # 224 x 224 = MatchRefineDataset resolution (infer_matching.py:178)
def test_hrnet_refine_matches_oracle(dev, n, h, w):
    from oracle import cnn_oracle
    from vsc_hip import cnn
    sd = cnn_synth.hrnet_refine_state(21)
    x = cnn_synth.similarity_maps(22, n, h, w)
    with torch.no_grad():
        want_logits = cnn_oracle.hrnet_refine(sd, x)
        want = cnn_oracle.match_refine_probability([sd], x) if h <= 64 else want_logits.softmax(dim=1)
    model = cnn.HRNetRefineHip(sd, dev)
    got_logits = model(x.to(dev)).cpu()
    assert got_logits.shape == (n, 2, h, w)
    assert float((got_logits - want_logits).abs().max()) < 1e-3 * max(1.0, float(want_logits.abs().max()))
    got = cnn.match_refine_probability([model], x.to(dev)).cpu() if h <= 64 else got_logits.softmax(dim=1)
    assert float((got - want).abs().max()) < 1e-4            # probability map (tier bound 1e-3)


def test_hrnet_fuse0_per_source_equals_the_concatenated_form(dev):
    """fuse.0 applied per source at the source's resolution and summed (a 1 x 1 convolution commutes with nearest upsampling)
    against the convolution over the materialised 336-channel concatenation: the same products, partial sums in source order."""
    from vsc_hip import cnn
    net = cnn.HRNetRefineHip(cnn_synth.hrnet_refine_state(7), dev)
    x = cnn_synth.similarity_maps(8, 2, 64, 96).to(dev)
    a = net(x)
    cnn.HRNetRefineHip.SPLIT_FUSE0 = False
    try:
        b = net(x)
    finally:
        cnn.HRNetRefineHip.SPLIT_FUSE0 = True
    assert a.shape == b.shape and float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max())) and not torch.equal(a, b)


def test_match_classify_and_refine_entry_points(dev):
    """src.matching.match_classify / match_refine (infer_matching.py:158-204) end to end on synthetic candidates: datasets,
    batching, model averaging, transposed pass, crop to the valid h x w -- against the oracle's restatement of the same steps,
    and into generate_matching_result (a planted diagonal segment must come out as a match)."""
    from oracle import cnn_oracle
    from src import matching
    rng = np.random.RandomState(3)
    cls_sds = [cnn_synth.mobilenetv3_small_state(31), cnn_synth.mobilenetv3_small_state(32)]
    ref_sds = [cnn_synth.hrnet_refine_state(33)]
    cls_models, refine_models = matching.load_match_models(cls_sds, ref_sds, dev)
    feats = [(rng.randn(h, w) * 0.1).astype(np.float32) for h, w in ((30, 50), (200, 170), (12, 9))]   # one larger than 160: cropped
    infos = [("Q1", "R1"), ("Q2", "R7"), ("Q3", "R2")]
    got = matching.match_classify(cls_models, feats, infos, batch_size=2, device=dev)
    assert [(q, r) for q, r, _ in got] == infos
    data = matching.MatchClassifyDataset(feats, infos, matching.MATCH_CLS_RESOLUTION)
    x = torch.from_numpy(np.stack([data[i][0] for i in range(3)]))
    with torch.no_grad():
        want = cnn_oracle.match_classify_probability([{k[len("model."):]: v for k, v in sd.items()} for sd in cls_sds], x)
    assert np.allclose([p for _, _, p in got], want.numpy(), atol=1e-4)

    d = 64
    meta = []
    for qn, rn in ((20, 36), (7, 11)):
        q = rng.randn(qn, d).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        r = rng.randn(rn, d).astype(np.float32)
        r /= np.linalg.norm(r, axis=1, keepdims=True)
        meta.append((f"Q{qn}", f"R{rn}", q, r))
    res = matching.match_refine(refine_models, meta, batch_size=1, device=dev)
    assert [(a, b) for a, b, _, _ in res] == [("Q20", "R36"), ("Q7", "R11")]
    for (qid, rid, prob, sim), (_, _, q, r) in zip(res, meta):
        assert prob.shape == sim.shape == (len(q), len(r))
        canvas = np.zeros((224, 224), np.float32)
        canvas[:len(q), :len(r)] = sim
        with torch.no_grad():
            want = cnn_oracle.match_refine_probability(ref_sds, torch.from_numpy(np.stack([canvas] * 3)[None]))[0, 1, :len(q), :len(r)]
        assert np.abs(prob - want.numpy()).max() < 1e-4
        assert np.abs(sim - q @ r.T).max() < 1e-5
    out = matching.generate_matching_result([["Qx", "Ry", np.eye(40, dtype=np.float32) * 0.9, None]])
    assert len(out) == 1 and out[0][:2] == ["Qx", "Ry"]


def test_infer_matching_entry_point_end_to_end(dev, tmp_path):
    """infer_matching.py (Main.run() of the reference from the query descriptors on): score normalisation -> candidate
    retrieval -> classifier -> refinement -> localisation, on synthetic descriptor files with a planted copy and
    random-weight networks.  Checks the plumbing the reference's run() has: candidate csv = brute-force statement of
    :229-262, every stage consumes the previous one's output, output columns / ordering."""
    import csv
    import importlib
    import subprocess
    from src import matching
    from vsc.baseline.score_normalization import normalize, query_score_normalize, ref_score_normalize
    from vsc.index import VideoFeature
    from vsc.storage import store_features
    rng = np.random.RandomState(7)
    d = 512
    mk = lambda pre, i, n: VideoFeature(video_id=f"{pre}{i:06d}", timestamps=np.arange(n, dtype=np.float32),
                                        feature=rng.randn(n, d).astype(np.float32))
    refs = [mk("R", 200000 + i, n) for i, n in enumerate((30, 44, 25))]
    norm = [mk("R", 100000 + i, 20) for i in range(4)]
    queries = [mk("Q", 300000 + i, n) for i, n in enumerate((18, 27))]
    queries[0].feature[3:15] = refs[1].feature[10:22] + 0.05 * rng.randn(12, d).astype(np.float32)   # planted copy
    sn_refs = ref_score_normalize(refs, norm, beta=1.5, nk=10)
    paths = {}
    for name, feats in (("q", queries), ("norm", norm), ("refs", refs), ("sn", sn_refs)):
        paths[name] = str(tmp_path / f"{name}.npz")
        store_features(paths[name], feats)
    cls_paths, ref_paths = [], []
    for i in range(2):
        p = str(tmp_path / f"cls{i}.pt")
        torch.save(cnn_synth.mobilenetv3_small_state(40 + i), p)
        cls_paths.append(p)
    p = str(tmp_path / "refine0.pt")
    torch.save(cnn_synth.hrnet_refine_state(50), p)
    ref_paths.append(p)
    out_csv, cand_csv = str(tmp_path / "out" / "matches.csv"), str(tmp_path / "cands.csv")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "vsc22-submission_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, os.path.join(root, "vsc22-submission_amd", "infer_matching.py"), "--query_features", paths["q"],
                          "--norm_refs", paths["norm"], "--refs", paths["refs"], "--sn_refs", paths["sn"], "--cls_models", *cls_paths,
                          "--refine_models", *ref_paths, "--candidates_csv", cand_csv, "--output", out_csv],
                         capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    # candidate retrieval == the reference's dict of best frame scores above SEARCH_THRESHOLD (float64 brute force)
    import collections
    low = matching.calclualte_low_var_dim(norm)
    snq = query_score_normalize(queries, norm, collections.defaultdict(lambda: 1.0), low_var_dim=low, beta=1.5, nk=10)
    want = {}
    for q in snq:
        for r in sn_refs:
            s = (q.feature.astype(np.float64) @ r.feature.astype(np.float64).T).max()
            if s > matching.SEARCH_THRESHOLD:
                want[q.video_id, r.video_id] = s
    with open(cand_csv) as f:
        got = [(r[0], r[1], float(r[2])) for r in list(csv.reader(f))[1:]]
    assert {(q, r) for q, r, _ in got} == set(want) and all(abs(s - want[q, r]) < 1e-4 for q, r, s in got)
    assert all(a[2] >= b[2] for a, b in zip(got, got[1:])) and got[0][:2] == ("Q300000", "R200001")   # the planted pair ranks first
    with open(out_csv) as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["query_id", "ref_id", "query_start", "query_end", "ref_start", "ref_end", "score"]
    for r in rows[1:]:
        assert (r[0], r[1]) in want and float(r[2]) <= float(r[3]) and float(r[4]) <= float(r[5])
