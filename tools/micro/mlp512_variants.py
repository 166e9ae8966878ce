"""swin_mlp512_kernel<V>: the variants of the generated body (csrc/gen_mlp512_loop.py VARIANTS) on one box -- time at Swin-V2-B's stage-2
shape (256 frames: 65 536 rows) and bit equality with variant 0 (they differ in schedule only).  Variants >= 2 exist in a diagnostic
build only:   make -C vsc22-submission_amd/csrc clean && make -C vsc22-submission_amd/csrc -j8 EXTRA=-DVSC_MLP_ABLATION   (run on the GPU box)
    python tools/micro/mlp512_variants.py [variants, default 0,1,2,3,4,5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))
import torch
from vsc_hip import _lib, ops
from vsc_hip._lib import check, ptr, current_stream
lib = _lib.require_device()
dev = torch.device("cuda:0")
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5".split(","))]
m, c = 256 * 256, 512
g = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn(m, c, device=dev, generator=g)
w1 = torch.randn(4 * c, c, device=dev, generator=g) * c ** -0.5
w2 = torch.randn(c, 4 * c, device=dev, generator=g) * (4 * c) ** -0.5
b1, b2 = torch.randn(4 * c, device=dev, generator=g) * 0.2, torch.randn(c, device=dev, generator=g) * 0.2
gam, bet = 0.3 + 0.05 * torch.randn(c, device=dev, generator=g), 0.05 * torch.randn(c, device=dev, generator=g)
ref = None
for v in variants:
    _lib.set_option("VSC_SWIN_MLP_ABL", str(v) if v else None)
    y, yb = ops.swin_mlp_bf16(x0, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu(), gam.cpu(), bet.cpu(), 1e-5)
    if ref is None:
        ref = y.clone()
    same = bool(torch.equal(y, ref))
    # timing on resident buffers (in place on x: values drift, the schedule does not care)
    import numpy as np
    w2p = np.empty((c, 4 * c), dtype=np.float32)
    check(lib.vsc_swin_mlp_permute_hidden_f32(np.ascontiguousarray(w2.cpu().numpy()).ctypes.data, w2p.ctypes.data, c))
    w1d, w2d = w1.to(torch.bfloat16), torch.from_numpy(w2p).to(dev).to(torch.bfloat16)
    x, xb = x0.clone(), x0.to(torch.bfloat16)
    ts = []
    for _ in range(11):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.vsc_swin_mlp_bf16(ptr(w1d), ptr(b1), ptr(w2d), ptr(b2), ptr(gam), ptr(bet), ptr(x), ptr(xb), m, c, 1e-5, current_stream()))
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"variant {v}: {t:.1f} us (min {min(ts):.1f})  {16.0 * m * c * c / t / 1e6:.0f} TF/s   bit-identical to variant {variants[0]}: {same}", flush=True)
# per-wave cycle counters of the timing variant
if "--timing" in sys.argv:
    nwg = (m + 127) // 128
    buf = torch.zeros(nwg * 4 * 8, dtype=torch.int32, device=dev)
    check(lib.vsc_debug_mlp512_timing(ptr(buf)))
    _lib.set_option("VSC_SWIN_MLP_ABL", "5")
    x, xb = x0.clone(), x0.to(torch.bfloat16)
    for _ in range(3):
        check(lib.vsc_swin_mlp_bf16(ptr(w1d), ptr(b1), ptr(w2d), ptr(b2), ptr(gam), ptr(bet), ptr(x), ptr(xb), m, c, 1e-5, current_stream()))
    torch.cuda.synchronize()
    check(lib.vsc_debug_mlp512_timing(None))
    t = buf.view(nwg, 4, 8).cpu().numpy().astype("float64")
    names = ["wait for own DMA (64 iters)", "barrier (64 iters)", "work (63 iters)", "start -> epilogue", "epilogue"]
    for k, nme in enumerate(names):
        print(f"{nme:32s} mean {t[:, :, k].mean():10.0f}  min {t[:, :, k].min():10.0f}  max {t[:, :, k].max():10.0f} cycles;  first round of workgroups {t[:256, :, k].mean():10.0f}, second {t[256:, :, k].mean():10.0f}")
_lib.set_option("VSC_SWIN_MLP_ABL", None)
