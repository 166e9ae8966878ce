"""Per-call time of one MobileNetV3-small classifier pass (2048 x 3 x 160 x 160) on the HIP path, grouped by op and shape."""
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


def timed(name, fn, keyfn, bytesfn):
    def wrap(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = fn(*a, **k)
        e1.record()
        log.append((keyfn(y, *a, **k), e0, e1, bytesfn(y, *a, **k)))
        return y
    return wrap


cnn.Conv.__call__ = timed("conv", cnn.Conv.__call__,
                          lambda y, self, x, act=None, residual=None, out=None, coff=0: ("conv", self.cin, self.cout, self.kh, self.stride, x.shape[1], residual is not None),
                          lambda y, self, x, act=None, residual=None, out=None, coff=0: 4.0 * (x.numel() + y.numel() * (2 if residual is not None else 1)))
cnn.DwConv.__call__ = timed("dw", cnn.DwConv.__call__, lambda y, self, x, act=None: ("dwconv", self.c, self.c, self.kh, self.stride, x.shape[1], False),
                            lambda y, self, x, act=None: 4.0 * (x.numel() + y.numel()))
cnn.avgpool = timed("pool", cnn.avgpool, lambda y, x: ("avgpool", x.shape[3], x.shape[3], 0, 0, x.shape[1], False), lambda y, x: 4.0 * x.numel())
se_call = cnn.SqueezeExcite.__call__
cnn.SqueezeExcite.__call__ = timed("se", se_call, lambda y, self, x: ("se_total", x.shape[3], x.shape[3], 0, 0, x.shape[1], False), lambda y, self, x: 4.0 * 3 * x.numel())
cls = cnn.MobileNetV3SmallHip(cnn_synth.mobilenetv3_small_state(1), dev)
x = cnn_synth.similarity_maps(2, 8, 160, 160).to(dev).repeat(256, 1, 1, 1)
for it in range(3):
    log.clear()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    cls(x)
    t1.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, by in log:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
    a[2] += by
print(f"pass {t0.elapsed_time(t1):.2f} ms; {len(log)} calls (se_total contains its avgpool and two convolutions)")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:9s} {key[1]:4d} {key[2]:4d} k{key[3]} s{key[4]} in {key[5]:3d} res {int(key[6])} n {a[0]:2d} total {a[1]:8.1f} us avg {a[1] / a[0]:7.1f} GB/s {a[2] / a[1] / 1e3:6.0f}")
