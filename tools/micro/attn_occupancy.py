import os, sys
sys.path.insert(0, "vsc22-submission_amd")
import torch
from vsc_hip import ops
dev = torch.device("cuda:0")
frames, heads = 332, 12
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for tokens in (128, 160, 176, 192, 197, 208, 224, 256):
    qkv = torch.randn(frames * tokens, 3 * heads * 64, device=dev).to(torch.bfloat16)
    ts = []
    for _ in range(8):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attention_bf16(qkv, frames, tokens, heads); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[len(ts)//2]
    mb = frames * tokens * (3 + 1) * heads * 64 * 2 / 1e6
    print(f"tokens {tokens}: {us:.1f} us  bytes {mb:.0f} MB -> {mb/us:.2f} TB/s  us per token^2 x1e3 {us/tokens**2*1e3:.3f}", flush=True)
