"""vsc_attention_bf16 at the ViT-B/16 shape (332 frames x 197 tokens x 12 heads), with and without the qkv tensor in the
Infinity Cache.  (run on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch

from vsc_hip import ops
from vsc_hip import _lib as _vsc_lib

dev = torch.device("cuda:0")
frames, tokens, heads = 332, 197, 12
qkv = torch.randn(frames * tokens, 3 * heads * 64, device=dev).to(torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # flushes the Infinity Cache between launches


def timeit(it=10, flush=True):
    ts = []
    for _ in range(it):
        if flush:
            big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_bf16(qkv, frames, tokens, heads)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


ops.attention_bf16(qkv, frames, tokens, heads)
print(f"cold (Infinity Cache flushed) {timeit():.1f} us   warm {timeit(flush=False):.1f} us", flush=True)

for skew in (0, 4000, 8000, 12000, 16000, 24000, 32000):
    _vsc_lib.set_option("VSC_ATTN_SKEW", str(skew))
    print(f"skew {skew}: cold {timeit():.1f} us   warm {timeit(flush=False):.1f} us", flush=True)
_vsc_lib.set_option("VSC_ATTN_SKEW", None)

# ablations (library built with -DVSC_ATTN_ABLATION): what each part of the kernel costs
for abl, what in ((1, "no exp2"), (2, "no PV MFMA"), (4, "no QK MFMA"), (6, "no MFMA"), (7, "no MFMA, no exp2"), (8, "no K/V loads"), (16, "no stores"),
                  (24, "no K/V loads, no stores"), (31, "nothing but Q loads + LDS + VALU")):
    _vsc_lib.set_option("VSC_ATTN_ABL", str(abl))
    print(f"abl {abl:2d} ({what}): cold {timeit():.1f} us   warm {timeit(flush=False):.1f} us", flush=True)
_vsc_lib.set_option("VSC_ATTN_ABL", None)
