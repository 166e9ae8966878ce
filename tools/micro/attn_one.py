"""A few launches of vsc_attention_bf16 at the ViT-B/16 shape (for PMC passes): python tools/micro/attn_one.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import ops
dev = torch.device("cuda:0")
frames, tokens, heads = 332, 197, 12
qkv = torch.randn(frames * tokens, 3 * heads * 64, device=dev).to(torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.attention_bf16(qkv, frames, tokens, heads)
torch.cuda.synchronize()
