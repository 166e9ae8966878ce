"""Run-to-run bit equality of every kernel family that publishes LDS-DMA data through a barrier, at the sizes the bench runs.
Parity tests compare values within a tolerance on a handful of frames; round 3's LDS-DMA publication bug (a bare s_barrier
in front of freshly DMA'd weight chunks) passed all of them and corrupted a few frames in every other run at full chunk size.
A race shows as a difference between two runs of the same call -- so every such kernel gets five repeats here."""
import numpy as np
import pytest
import torch

from tools import synth
from vsc_hip.config import get_config

pytestmark = pytest.mark.gpu

REPEATS = 5


@pytest.fixture(scope="module")
def dev():
    from vsc_hip import _lib
    _lib.require_device()
    return torch.device("cuda:0")


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_vit_encoder_two_lanes_full_chunks(dev, precision):
    """ViT-B/16 at the benchmarked configuration: 332-frame chunks on two lanes, persistent v4 GEMMs (LDS-DMA ring streaming
    across tiles), attention, LayerNorm -- 700 frames = two full chunks + a ragged one.  Both builds of the library."""
    from vsc_hip.encoder import HipEncoder
    cfg = get_config("vit_b16_224")
    enc = HipEncoder(cfg, synth.encoder_weights(7, cfg), max_batch=332, l2_normalize=True, lanes=2, precision=precision)
    base = torch.from_numpy(synth.frames(21, 20, cfg)).to(dev)
    x = (base.repeat(35, 1, 1, 1) + 0.001 * torch.arange(700, device=dev).view(-1, 1, 1, 1)).contiguous()
    first = enc(x).clone()
    assert torch.isfinite(first).all()
    for _ in range(REPEATS):
        assert torch.equal(enc(x), first)
    enc.close()


def test_knn_prefilter_sweep_65536_x_1m(dev):
    """vsc_knn_ip_f32 at 65 536 x 1M x 512, k = 100 (bf16 pre-filter sweep: the shared 256 x 256 x 64 LDS-DMA loop, candidate
    lists, compactions, exact re-scoring).  The survivors' ORDER in the lists may differ between runs; the result may not."""
    from vsc_hip import ops
    g = torch.Generator(device=dev).manual_seed(4)
    r = torch.randn(1000000, 512, generator=g, device=dev)
    q = torch.randn(65536, 512, generator=g, device=dev)
    ops.l2_normalize_(r)
    ops.l2_normalize_(q)
    D0, I0 = ops.knn_ip(q, r, 100)
    D0, I0 = D0.clone(), I0.clone()
    assert bool((D0[:, :-1] >= D0[:, 1:]).all()) and int(I0.min()) >= 0
    for _ in range(REPEATS):
        D, I = ops.knn_ip(q, r, 100)
        assert torch.equal(D, D0) and torch.equal(I, I0)
    del r, q
    torch.cuda.empty_cache()


def test_range_and_pair_max_sweeps_repeat(dev):
    """The two other sweeps on the shared loop (fixed threshold): CSR of the range search and the video-pair table."""
    from vsc_hip import ops
    g = torch.Generator(device=dev).manual_seed(5)
    r = torch.randn(400000, 512, generator=g, device=dev)
    q = torch.randn(8192, 512, generator=g, device=dev)
    ops.l2_normalize_(r)
    ops.l2_normalize_(q)
    qv = (torch.arange(8192, device=dev) // 32).int()
    rv = (torch.arange(400000, device=dev) // 32).int()
    first = [t.clone() for t in ops.range_search_ip(q, r, 0.15, capacity=1 << 22)]
    firstp = [t.clone() for t in ops.video_pair_max(q, qv, 256, r, rv, 12500, 0.15, capacity=1 << 22)]
    assert int(first[0][-1]) > 1000
    for _ in range(REPEATS):
        for a, b in zip(ops.range_search_ip(q, r, 0.15, capacity=1 << 22), first):
            assert torch.equal(a, b)
        for a, b in zip(ops.video_pair_max(q, qv, 256, r, rv, 12500, 0.15, capacity=1 << 22), firstp):
            assert torch.equal(a, b)


@pytest.mark.parametrize("name,m,n,k,epi", [("swin s2 fc1", 65536, 2048, 512, "GELU"), ("vit fc1", 65404, 3072, 768, "GELU"),
                                            ("clip fc1", 65535, 4096, 1024, "QGELU"), ("swin s0 qkv (ragged N)", 1048576, 384, 128, "BF16"),
                                            ("vit proj", 65404, 768, 768, "RESADD")])
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_persistent_gemm_repeats_and_equals_the_one_tile_kernel(dev, name, m, n, k, epi, precision):
    """The persistent kernel's write-outs at full size: bit-equal to the one-tile-per-workgroup kernel (same K loop, same rounding
    points) and to themselves over 20 launches.  Round 4's first buffer-descriptor write-out stored garbage in ~1 % of the GELU
    tiles, different ones every run (a 16-byte buffer store with a scalar offset whose data register the next pass's first VALU
    instruction overwrote: common.h buffer_store_b128_soff) -- every parity test at small sizes passed."""
    from vsc_hip import _lib, ops
    code = {"GELU": _lib.EPI_GELU_BF16, "QGELU": _lib.EPI_QGELU_BF16, "BF16": _lib.EPI_BF16, "RESADD": _lib.EPI_RESADD_F32}[epi]
    lp = torch.float16 if precision == "fp16" else torch.bfloat16     # (the fp16 build: v_cvt_pk_f16_f32 in the same write-outs)
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(m, k, generator=g, device=dev).to(lp)
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).to(lp)
    b = torch.randn(n, generator=g, device=dev)
    aux0 = torch.randn(m, n, generator=g, device=dev) if epi == "RESADD" else None

    def run():
        aux = aux0.clone() if aux0 is not None else None
        return ops.gemm_bf16(a, w, b, epilogue=code, aux=aux, out=aux).clone()

    with ops.operands(precision):
        with _lib.option("VSC_GEMM_V4", "0"):
            ref = run()
        first = run()
        assert torch.isfinite(first.float()).all()
        assert torch.equal(first, ref), name
        for _ in range(20):
            assert torch.equal(run(), first), name


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_swin_mlp512_repeats_at_stage2_size(dev, precision):
    """swin_mlp512_kernel (one asm statement with hand-assigned registers and hand-counted waits: csrc/gen_mlp512_loop.py) at Swin-V2-B's
    stage-2 size, 256 frames = 65 536 rows + a ragged tile: five launches from the same input, bit for bit.  A missing wait state or
    a too-early barrier shows as a difference between runs long before it shows in a tolerance."""
    from vsc_hip import ops
    g = torch.Generator().manual_seed(3)
    m, c = 65536 + 50, 512
    x0 = torch.randn(m, c, generator=g)
    w1, b1 = torch.randn(4 * c, c, generator=g) * c ** -0.5, torch.randn(4 * c, generator=g) * 0.2
    w2, b2 = torch.randn(c, 4 * c, generator=g) * (4 * c) ** -0.5, torch.randn(c, generator=g) * 0.2
    gam, bet = 0.3 + 0.05 * torch.randn(c, generator=g), 0.05 * torch.randn(c, generator=g)
    xd = x0.to(dev)
    with ops.operands(precision):
        first, first_b = ops.swin_mlp_bf16(xd, w1, b1, w2, b2, gam, bet, 1e-5)
        assert torch.isfinite(first).all()
        for _ in range(REPEATS):
            y, yb = ops.swin_mlp_bf16(xd, w1, b1, w2, b2, gam, bet, 1e-5)
            assert torch.equal(y, first) and torch.equal(yb, first_b)


def test_matching_networks_repeat_at_the_benchmarked_batches(dev):
    """conv.hip at the sizes bench.py times -- HRNet-W18 refinement net on 16 x 3 x 224 x 224 (3 136 tiles of the split-bf16 direct
    kernel, the tap-streamed 64-channel kernel, the plane implicit GEMM), MobileNetV3 classifier on 2048 x 3 x 160 x 160 -- five
    passes each, bit for bit.  These kernels use both mechanisms that corrupted data silently in earlier rounds (LDS-DMA
    publication through a barrier, 16-byte buffer stores with a scalar offset); their parity tests stop at 16 x 56 x 56."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cnn_synth
    from vsc_hip import cnn
    net = cnn.HRNetRefineHip(cnn_synth.hrnet_refine_state(7), dev)
    x = cnn_synth.similarity_maps(3, 16, 224, 224).to(dev)
    first = net(x).clone()
    assert torch.isfinite(first).all()
    for _ in range(REPEATS):
        assert torch.equal(net(x), first)
    del net, x, first
    cls = cnn.MobileNetV3SmallHip(cnn_synth.mobilenetv3_small_state(1), dev)
    xc = cnn_synth.similarity_maps(2, 8, 160, 160).to(dev).repeat(256, 1, 1, 1)
    first = cls(xc).clone()
    assert torch.isfinite(first).all() and first.shape[0] == 2048
    for _ in range(REPEATS):
        assert torch.equal(cls(xc), first)
