"""Is the ViT attention launch bound by HBM access locality?  The same 3984 (frame, head) items of 197 tokens, once in the encoder's
layout (qkv row-major [token][3 x 12 x 64]: an item's rows are 128-byte pieces 4608 bytes apart) and once as 3984 one-head
'frames' (each item a contiguous 75.6-KB block).  Same MFMAs, same bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
dev = torch.device("cuda:0")
tokens = 197
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def timeit(frames, heads, flush):
    qkv = torch.randn(frames * tokens, 3 * heads * 64, device=dev).to(torch.bfloat16)
    for _ in range(3): ops.attention_bf16(qkv, frames, tokens, heads)
    ts = []
    for _ in range(12):
        if flush: big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attention_bf16(qkv, frames, tokens, heads); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for mode in ("0", "1"):
    _lib.set_option("VSC_ATTN_DMA", mode)
    for flush in (True, False):
        a = timeit(332, 12, flush); b = timeit(3984, 1, flush)
        print(f"kernel {mode} {'cold' if flush else 'warm'}: interleaved heads {a:.1f} us   contiguous items {b:.1f} us", flush=True)
_lib.set_option("VSC_ATTN_DMA", None)
