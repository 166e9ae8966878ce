import os, sys
sys.path.insert(0, "/root/repo/vsc22-submission_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
from vsc_hip import cnn, _lib
dev = torch.device("cuda:0")
for (n, h, w, cin, cout, res) in [(16, 224, 224, 256, 20, False), (3, 150, 171, 256, 20, False), (16, 56, 56, 72, 72, True), (7, 61, 45, 72, 72, False), (16, 28, 28, 144, 144, True), (4, 128, 128, 64, 64, False), (3, 150, 171, 64, 48, True), (16, 224, 224, 64, 64, False), (2, 19, 40, 36, 36, True), (16, 28, 28, 36, 36, True), (16, 112, 112, 36, 36, True), (2, 16, 32, 20, 18, True), (3, 21, 45, 20, 20, True), (1, 5, 3, 20, 18, False), (16, 56, 56, 20, 18, True), (1, 8, 33, 20, 32, False), (16, 224, 224, 20, 20, True)]:
    rng = np.random.RandomState(cin + h)
    sd = {"c.weight": torch.from_numpy((rng.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)), "c.bias": torch.from_numpy(rng.randn(cout).astype(np.float32) * 0.1)}
    x = torch.from_numpy(rng.randn(n, h, w, cin).astype(np.float32)).to(dev)
    r = torch.from_numpy(rng.randn(n, h, w, cout).astype(np.float32)).to(dev) if res else None
    conv = cnn.Conv(sd, "c", None, 1, dev)
    _lib.set_option("VSC_CONV_X3", "0")
    ref = conv(x, act="relu", residual=r).clone()
    _lib.set_option("VSC_CONV_X3", "1")
    got = conv(x, act="relu", residual=r).clone()
    for _ in range(2): assert torch.equal(conv(x, act="relu", residual=r), got)
    ts = []
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(x, act="relu", residual=r); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    _lib.set_option("VSC_CONV_X3", "0")
    t0 = []
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(x, act="relu", residual=r); e1.record(); torch.cuda.synchronize(); t0.append(e0.elapsed_time(e1) * 1e3)
    _lib.set_option("VSC_CONV_X3", None)
    want = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), sd["c.weight"].double(), sd["c.bias"].double(), padding=1)
    if res: want = want + r.double().cpu().permute(0, 3, 1, 2)
    want = F.relu(want).permute(0, 2, 3, 1)
    print(f"{(n,h,w,cin,cout,res)}: x3 vs fp64 max {float((got.double().cpu()-want).abs().max()):.2e}  fp32-mfma vs fp64 max {float((ref.double().cpu()-want).abs().max()):.2e}   x3 {min(ts):.1f} us  fp32 {min(t0):.1f} us")
