"""Per-layer time of one HRNet refinement pass (16 x 3 x 224 x 224) on the HIP path: every convolution / upsample-add call is
bracketed by events; calls are grouped by (op, cin, cout, kernel, stride, input h x w)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import cnn_synth
from vsc_hip import cnn

dev = torch.device("cuda:0")
log = []
conv_call, up_call = cnn.Conv.__call__, cnn.upsample_into


def conv_timed(self, x, act=None, residual=None, out=None, coff=0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = conv_call(self, x, act, residual, out, coff)
    e1.record()
    flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * self.cout * self.cin * self.kh * self.kw
    log.append((("conv", self.cin, self.cout, self.kh, self.stride, x.shape[1], x.shape[2], residual is not None), e0, e1, flops,
                4.0 * (x.numel() + y.numel() * (2 if residual is not None else 1))))
    return y


def up_timed(src, out, factor=1, coff=0, accumulate=False, act=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = up_call(src, out, factor, coff, accumulate, act)
    e1.record()
    n, h, w, _ = out.shape
    log.append((("upsample_add", src.shape[3], out.shape[3], factor, int(accumulate), h, w, False), e0, e1, 0.0,
                4.0 * (src.numel() + n * h * w * src.shape[3] * (2 if accumulate else 1))))
    return y


cnn.Conv.__call__ = conv_timed
cnn.upsample_into = up_timed
ref = cnn.HRNetRefineHip(cnn_synth.hrnet_refine_state(3), dev)
y = cnn_synth.similarity_maps(4, 16, 224, 224).to(dev)
for it in range(3):
    log.clear()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    ref(y)
    t1.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl, by in log:
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
    a[2] += fl
    a[3] += by
tot = sum(a[1] for a in agg.values())
print(f"pass {t0.elapsed_time(t1):.2f} ms; sum of bracketed calls {tot / 1e3:.2f} ms; {len(log)} calls")
print(f"{'op':13s} {'cin':>4s} {'cout':>4s} k s {'h x w':>9s} res {'n':>3s} {'total us':>9s} {'avg us':>8s} {'TF/s':>6s} {'GB/s':>6s} {'%':>5s}")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    op, ci, co, k, s, h, w, res = key
    print(f"{op:13s} {ci:4d} {co:4d} {k} {s} {h:4d}x{w:<4d} {int(res):3d} {a[0]:3d} {a[1]:9.1f} {a[1] / a[0]:8.1f} {a[2] / a[1] / 1e6:6.1f} "
          f"{a[3] / a[1] / 1e3:6.0f} {100 * a[1] / tot:5.1f}")
