"""vsc_window_attention_bf16 at Swin-V2-B's stage-3 shape (256 frames x 16x16 tokens, one window, 16 heads) with the
ablations of the -DVSC_ATTN_ABLATION build (VSC_WATTN_ABL).  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
from vsc_hip import _lib as _vsc_lib
dev = torch.device("cuda:0")
frames, res, window, heads = 256, 16, 16, 16
qkv = torch.randn(frames * res * res, 3 * heads * 32, device=dev).to(torch.bfloat16)
bias = torch.randn(heads, (2 * window - 1) ** 2, device=dev)
scale = torch.full((heads,), 10.0, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def timeit(shift, it=10, flush=False):
    ts = []
    for _ in range(it):
        if flush: big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.window_attention_bf16(qkv, bias, scale, frames, res, window, shift, heads); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
ops.window_attention_bf16(qkv, bias, scale, frames, res, window, 0, heads)
print(f"stage 3 (res 16 = one window): warm {timeit(0):.1f} us  cold {timeit(0, flush=True):.1f} us", flush=True)
for abl, what in ((1, "no exp2"), (2, "no MFMA"), (3, "no MFMA, no exp2"), (8, "no K/V loads"), (16, "no stores"), (24, "no K/V loads, no stores"), (27, "only Q loads + LDS + rest of VALU")):
    _vsc_lib.set_option("VSC_WATTN_ABL", str(abl))
    print(f"abl {abl:2d} ({what}): warm {timeit(0):.1f} us  cold {timeit(0, flush=True):.1f} us", flush=True)
_vsc_lib.set_option("VSC_WATTN_ABL", None)
# stage 1 / 2 shapes (shifted windows)
for fr, rs, hd in ((64, 64, 4), (128, 32, 8)):
    q2 = torch.randn(fr * rs * rs, 3 * hd * 32, device=dev).to(torch.bfloat16)
    b2 = torch.randn(hd, (2 * window - 1) ** 2, device=dev); s2 = torch.full((hd,), 10.0, device=dev)
    ops.window_attention_bf16(q2, b2, s2, fr, rs, window, 8, hd)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.window_attention_bf16(q2, b2, s2, fr, rs, window, 8, hd); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{fr} frames res {rs} heads {hd} shift 8: {sorted(ts)[2]:.1f} us", flush=True)
