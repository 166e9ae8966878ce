"""A few launches of vsc_window_attention_bf16 at one Swin-V2-B stage shape (for rocprofv3 --pmc passes).
python tools/micro/wattn_one.py <stage 0..3> [shift] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
dev = torch.device("cuda:0")
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shift = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
frames, res, heads = 256, 64 >> stage, 4 << stage
window = min(16, res)
qkv = torch.randn(frames * res * res, 3 * heads * 32, device=dev).to(torch.bfloat16)
bias = torch.randn(heads, (2 * window - 1) ** 2, device=dev)
scale = torch.full((heads,), 10.0, device=dev)
ts = []
for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.window_attention_bf16(qkv, bias, scale, frames, res, window, shift, heads); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print(f"stage {stage} shift {shift}: {frames} frames res {res} heads {heads} window {window}: median {sorted(ts)[len(ts)//2]:.1f} us", flush=True)
