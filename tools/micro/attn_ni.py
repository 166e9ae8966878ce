"""vsc_attention_bf16 at the ViT-B/16 shape: one vs two (frame, head) items per workgroup (VSC_ATTN_NI), skew sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
from vsc_hip import _lib as _vsc_lib
dev = torch.device("cuda:0")
frames, tokens, heads = 332, 197, 12
qkv = torch.randn(frames * tokens, 3 * heads * 64, device=dev).to(torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def timeit(it=10, flush=True):
    ts = []
    for _ in range(it):
        if flush: big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attention_bf16(qkv, frames, tokens, heads); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
ref = None
for ni in ("1", "2"):
    for skew in ("0", "8000", "16000", "24000"):
        _vsc_lib.set_option("VSC_ATTN_NI", ni); _vsc_lib.set_option("VSC_ATTN_SKEW", skew)
        o = ops.attention_bf16(qkv, frames, tokens, heads)
        if ref is None: ref = o.clone()
        same = torch.equal(o, ref)
        print(f"NI {ni} skew {skew}: cold {timeit():.1f} us  warm {timeit(flush=False):.1f} us  identical {same}", flush=True)
