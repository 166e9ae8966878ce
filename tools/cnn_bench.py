"""Throughput of the matching track's two networks on the HIP path (run on the GPU box):
MobileNetV3-small classifier at the reference's batch (2048 x 3 x 160 x 160, infer_matching.py:159-160) and the HRNet-W18
refinement net at 16 x 3 x 224 x 224 (:178-179), synthetic timm-named weights."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import cnn_synth
from vsc_hip import cnn

dev = torch.device("cuda:0")


def timeit(fn, it):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


only = sys.argv[1] if len(sys.argv) > 1 else ""   # "hrnet": the refinement net alone
if only != "hrnet":
    cls = cnn.MobileNetV3SmallHip(cnn_synth.mobilenetv3_small_state(1), dev)
    x = cnn_synth.similarity_maps(2, 8, 160, 160).to(dev).repeat(256, 1, 1, 1)
    dt = timeit(lambda: cls(x), 3)
    print(f"mobilenetv3_small_100 classifier, batch {x.shape[0]} x 3 x 160 x 160: {dt * 1e3:.1f} ms, {x.shape[0] / dt:.0f} maps/s")
ref = cnn.HRNetRefineHip(cnn_synth.hrnet_refine_state(3), dev)
y = cnn_synth.similarity_maps(4, 16, 224, 224).to(dev)
dt = timeit(lambda: ref(y), 3)
print(f"hrnet_w18 refinement net, batch 16 x 3 x 224 x 224: {dt * 1e3:.1f} ms per pass, {16 / dt:.1f} maps/s (the reference runs 2 passes x 2 models per batch)")
