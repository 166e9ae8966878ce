"""The QKV phase of swin_mlp512_kernel<9> (the next block's qkv Linear behind proj + MLP) at Swin-V2-B's stage-2 shape (65 536 rows):
its time = launch with the phase - launch without (vsc_swin_proj_mlp_bf16), and what is left of it when one ingredient is taken out
(ablation variants 10 .. 13: wrong results, diagnostic build only):
    cd vsc22-submission_amd/csrc && VSC_GEN_QKV_ABL=1 python3 gen_mlp512_loop.py > swin_mlp512_loop.inc &&
        make -j8 EXTRA=-DVSC_MLP_ABLATION OBJDIR=../lib/obj_abl LIB=../lib/libvsc_hip_abl.so && git checkout swin_mlp512_loop.inc
    VSC_HIP_LIB=vsc22-submission_amd/lib/libvsc_hip_abl.so python tools/micro/qkv512_phase.py        (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib
from vsc_hip._lib import check, ptr, current_stream
lib = _lib.require_device()
dev = torch.device("cuda:0")
m, c = 256 * 256, 512
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(m, c, device=dev, generator=g)
xb = x.to(torch.bfloat16)
qkv = torch.empty(m, 3 * c, device=dev, dtype=torch.bfloat16)
att = torch.randn(m, c, device=dev, generator=g).to(torch.bfloat16)
mat = lambda r, k: (torch.randn(r, k, device=dev, generator=g) * k ** -0.5).to(torch.bfloat16)
wp, w1, w2, wq = mat(c, c), mat(4 * c, c), mat(c, 4 * c), mat(3 * c, c)
vec = lambda n, s=0.1: torch.randn(n, device=dev, generator=g) * s
bp, b1, b2, bq = vec(c), vec(4 * c), vec(c), vec(3 * c)
g1, be1, g2, be2 = 0.3 + vec(c, 0.05), vec(c, 0.05), 0.3 + vec(c, 0.05), vec(c, 0.05)


x0 = x.clone()


def once(fn):
    """one launch on the same input every time (the kernel works in place on x, and the chip's clock depends on the operand values)"""
    x.copy_(x0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def med(v):
    return sorted(v)[len(v) // 2]


names = {1: "proj + MLP without the phase", 0: "the stand-alone qkv GEMM", 9: "the phase as shipped", 10: "no LDS-DMA in the chunk loop", 11: "no barrier",
         12: "no bias + rounding + store", 13: "no fragment reads", 14: "stamped", 15: "bias + rounding, no stores",
         16: "fragment ring of 10", 17: "ring refills every other group", 18: "ring of 10 + refills every other group"}
RIGHT = (9, 16, 17, 18)      # variants whose results must equal variant 9's
variants = [1, 0] + [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "9,10,11,12,13,14,15,16,17,18".split(","))]


def launch(v):
    if v == 1:
        return lambda: check(lib.vsc_swin_proj_mlp_bf16(ptr(att), ptr(wp), ptr(bp), ptr(g1), ptr(be1), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(g2), ptr(be2),
                                                        ptr(x), ptr(xb), m, c, 1e-5, current_stream()))
    if v == 0:
        return lambda: check(lib.vsc_gemm_bf16(ptr(xb), ptr(wq), ptr(bq), None, ptr(qkv), m, 3 * c, c, _lib.EPI_BF16, 0, current_stream()))

    def f():
        _lib.set_option("VSC_SWIN_MLP_ABL", None if v == 9 else str(v))
        check(lib.vsc_swin_proj_mlp_qkv_bf16(ptr(att), ptr(wp), ptr(bp), ptr(g1), ptr(be1), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(g2), ptr(be2),
                                             ptr(wq), ptr(bq), ptr(x), ptr(qkv), m, c, 1e-5, current_stream()))
    return f


# the variants interleaved, round after round: box drift hits all of them alike
ts, extra, ref = {v: [] for v in variants}, {}, None
for rep in range(12):
    for v in variants:
        if rep == 0 and v >= 9:
            qkv.zero_()
        ts[v].append(once(launch(v)))
        if rep == 0 and v in RIGHT:
            if ref is None:
                ref = qkv.clone()
            extra[v] = "  results equal variant 9's: " + str(bool(torch.equal(qkv, ref)))
        if rep == 11 and v == 14:
            cyc = qkv.view(torch.int32)[:, :4].reshape(-1, 32, 4)[:, :16].float().mean(dim=(0, 1))      # (the first 16 rows of every wave's 32 carry the stamps)
            extra[v] = (f"\n      stamped, cycles per chunk (mean over the waves): whole phase {cyc[0] / 48:.0f} = wait for the LDS-DMA + barrier {cyc[1] / 48:.0f}"
                        f" + fragments, MFMAs, A's results {cyc[2] / 48:.0f} + B's results {cyc[3] / 48:.0f} (each stamp costs an SMEM round trip)")
base = med(ts[1][1:])
for v in variants:
    t = med(ts[v][1:])
    line = f"variant {v:2d} ({names.get(v, '?'):36s}): {t:7.1f} us (min {min(ts[v]):.1f}, max {max(ts[v][1:]):.1f})"
    if v >= 9:
        line += f"  -> phase {t - base:6.1f} us = {(t - base) / 2 * 2.4e3 / 48:5.0f} cycles per chunk at 2.4 GHz (MFMA floor 1024)"
    print(line + extra.get(v, ""))
_lib.set_option("VSC_SWIN_MLP_ABL", None)
