"""The matching track's two networks on the HIP path: the MobileNetV3 pair classifier and the HRNet refinement net
(reference: VSC22-Matching-Track-1st/train/models.py:6-47, used by infer/infer_matching.py:158-204 as TorchScript modules).

A model is built once from the reference's state dict (timm parameter names): every Conv+BatchNorm pair is folded into
one weight / bias, permuted to (cout, kh, kw, cin) and packed for the fp32 MFMA tiles (vsc_conv_pack_weight_f32).  The
forward pass is a sequence of C-ABI calls on NHWC float32 buffers -- vsc_conv2d_f32 (residual and activation fused),
vsc_dwconv2d_f32, vsc_global_avgpool_f32, vsc_channel_scale_f32, vsc_upsample_add_f32 / vsc_upsample_sum_f32.  torch provides
device memory only.  (Which kernel a layer runs on -- fp32 matrix tiles, the streaming pointwise kernels, the direct 3 x 3 kernels on
fp32 or on split-bf16 operands -- is vsc_conv2d_f32's choice by shape: csrc/conv.hip)
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr

FLOPS = None   # measurement hook (bench.py): a list here accumulates [0] 2 * MACs of every convolution call, [1] (if present) the share of
               # them that vsc_conv2d_f32 ran on the bf16 matrix pipe with split operands (vsc_conv_last_pipe)

ACT = {None: 0, "none": 0, "relu": 1, "hard_swish": 2, "hard_sigmoid": 3, "gelu": 4}
BN_EPS = 1e-5

# timm `mobilenetv3_small_100`: per stage, per block (kind, stride, activation); channel counts, kernel sizes and
# the presence of squeeze-excite are read off the state dict
MBV3_SMALL = (
    (("ds", 2, "relu"),),
    (("ir", 2, "relu"), ("ir", 1, "relu")),
    (("ir", 2, "hard_swish"), ("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")),
    (("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")),
    (("ir", 2, "hard_swish"), ("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")),
    (("cn", 1, "hard_swish"),),
)
HRNET_STAGES = ((2, 1, 2), (3, 4, 3), (4, 3, 4))   # (stage, modules, branches) of hrnet_w18; 4 BasicBlocks per branch
HRNET_BLOCKS = 4


def _np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _fold(sd: dict, conv: str, bn: str | None):
    """(weight [co, ci/groups, kh, kw], bias [co]) with the BatchNorm that follows folded in (float64 arithmetic)."""
    w = _np(sd[conv + ".weight"]).astype(np.float64)
    b = _np(sd[conv + ".bias"]).astype(np.float64) if conv + ".bias" in sd else np.zeros(w.shape[0])
    if bn is not None:
        g, beta = _np(sd[bn + ".weight"]).astype(np.float64), _np(sd[bn + ".bias"]).astype(np.float64)
        mu, var = _np(sd[bn + ".running_mean"]).astype(np.float64), _np(sd[bn + ".running_var"]).astype(np.float64)
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[:, None, None, None]
        b = (b - mu) * s + beta
    return w.astype(np.float32), b.astype(np.float32)


class Conv:
    """One dense convolution: packed weight + bias on the device."""

    def __init__(self, sd, conv, bn, stride=1, device="cuda", pad4=False, cin_map=None):
        """pad4: zero-pad output AND input channels to multiples of 4 (the layer then consumes / produces 16-byte aligned
        pixels: the patch gather runs on float4 loads; padded output channels are exactly 0 after ReLU / hardswish / none).
        cin_map: positions of the real input channels inside a wider, zero-padded input (concatenated padded features)."""
        lib = _lib.require_device()
        w, b = _fold(sd, conv, bn)
        if cin_map is not None:
            wide = np.zeros((w.shape[0], cin_map[1]) + w.shape[2:], np.float32)
            wide[:, cin_map[0]] = w
            w = wide
        if pad4:
            co, ci = -w.shape[0] % 4, -w.shape[1] % 4
            w = np.pad(w, ((0, co), (0, ci), (0, 0), (0, 0)))
            b = np.pad(b, (0, co))
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.stride, self.pad = stride, self.kh // 2
        k = self.cin * self.kh * self.kw
        flat = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(self.cout, k)).to(device)
        self.w = torch.empty((self.cout, lib.vsc_conv_packed_k(self.cin, self.kh, self.kw)), dtype=torch.float32, device=device)
        check(lib.vsc_conv_pack_weight_f32(ptr(flat), ptr(self.w), self.cout, k, current_stream()))
        torch.cuda.current_stream().synchronize()   # `flat` is released on return
        self.b = torch.from_numpy(b).to(device)

    def __call__(self, x, act=None, residual=None, out=None, coff=0):
        """x [n, h, w, cin] -> [n, ho, wo, cout]; with `out` [n, ho, wo, C] the result goes to channels coff : coff + cout."""
        lib = _lib.require_device()
        n, h, w, c = x.shape
        assert c == self.cin and x.is_contiguous() and x.dtype == torch.float32
        ho = (h + 2 * self.pad - self.kh) // self.stride + 1
        wo = (w + 2 * self.pad - self.kw) // self.stride + 1
        if out is None:
            out = torch.empty((n, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        assert out.shape[:3] == (n, ho, wo) and out.is_contiguous()
        ldo = out.shape[3]
        if residual is not None:
            assert residual.shape == (n, ho, wo, self.cout) and residual.is_contiguous()
        view = out.view(-1)[coff:] if coff else out
        check(lib.vsc_conv2d_f32(ptr(x), n, h, w, c, c, ptr(self.w), ptr(self.b), self.cout, self.kh, self.kw, self.stride,
                                 self.pad, ptr(residual), self.cout, ACT[act], ptr(view), ldo, current_stream()))
        if FLOPS is not None:
            f = 2.0 * n * ho * wo * self.cout * self.cin * self.kh * self.kw
            FLOPS[0] += f
            if len(FLOPS) > 1 and lib.vsc_conv_last_pipe() == 1:     # the share that ran on the bf16 pipe with split operands
                FLOPS[1] += f
        return out


class DwConv:
    def __init__(self, sd, conv, bn, stride, device="cuda"):
        w, b = _fold(sd, conv, bn)
        self.c, _, self.kh, self.kw = w.shape
        self.stride, self.pad = stride, self.kh // 2
        self.w = torch.from_numpy(np.ascontiguousarray(w.reshape(self.c, self.kh * self.kw))).to(device)
        self.b = torch.from_numpy(b).to(device)

    def __call__(self, x, act=None):
        lib = _lib.require_device()
        n, h, w, c = x.shape
        assert c == self.c and x.is_contiguous()
        ho = (h + 2 * self.pad - self.kh) // self.stride + 1
        wo = (w + 2 * self.pad - self.kw) // self.stride + 1
        out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        if FLOPS is not None:
            FLOPS[0] += 2.0 * n * ho * wo * c * self.kh * self.kw
        check(lib.vsc_dwconv2d_f32(ptr(x), n, h, w, c, ptr(self.w), ptr(self.b), self.kh, self.kw, self.stride, self.pad,
                                   ACT[act], ptr(out), current_stream()))
        return out


def avgpool(x):
    lib = _lib.require_device()
    n, h, w, c = x.shape
    out = torch.empty((n, 1, 1, c), dtype=torch.float32, device=x.device)
    check(lib.vsc_global_avgpool_f32(ptr(x), n, h * w, c, ptr(out), current_stream()))
    return out


def upsample_into(src, out, factor=1, coff=0, accumulate=False, act=None):
    """out[..., coff : coff + c] (+)= nearest-upsampled src, then act."""
    lib = _lib.require_device()
    n, h, w, ldo = out.shape
    c = src.shape[3]
    assert src.shape == (n, h // factor, w // factor, c) and src.is_contiguous() and out.is_contiguous()
    check(lib.vsc_upsample_add_f32(ptr(src), n, h, w, c, factor, ptr(out), ldo, coff, int(accumulate), ACT[act], current_stream()))
    return out


def upsample_sum(base, terms, out, act=None):
    """out = act(((base + up(t0)) + up(t1)) + up(t2)): one HRNet fuse node; terms = [(tensor [n, h / f, w / f, c], f), ...] (<= 3),
    base [n, h, w, c] or None, may be `out` itself."""
    lib = _lib.require_device()
    n, h, w, c = out.shape
    assert len(terms) <= 3 and out.is_contiguous() and (base is None or (base.shape == out.shape and base.is_contiguous()))
    args = []
    for k in range(3):
        if k < len(terms):
            t, f = terms[k]
            assert t.shape == (n, h // f, w // f, c) and t.is_contiguous()
            args += [ptr(t), f]
        else:
            args += [None, 0]
    check(lib.vsc_upsample_sum_f32(ptr(base), c, *args, n, h, w, c, ACT[act], ptr(out), c, current_stream()))
    return out


class SqueezeExcite:
    def __init__(self, sd, p, device):
        self.reduce = Conv(sd, p + ".conv_reduce", None, device=device)
        self.expand = Conv(sd, p + ".conv_expand", None, device=device)

    FUSED = True   # one launch per block where the image fits LDS (vsc_se_block_f32); False: avgpool + two convolutions + scale

    def __call__(self, x):
        lib = _lib.require_device()
        n, h, w, c = x.shape
        cr = self.reduce.cout
        # (the gate's Linears are re-read per image there: beyond ~8 k weights per matrix -- 240 x 64, 576 x 144 -- the batched GEMMs of the
        #  four-launch form win: 599 vs 160 us at 576 channels, 54 vs 121 us at 96)
        if self.FUSED and c % 4 == 0 and c * cr <= 8192 and (h * w * c + 2 * c + cr) * 4 <= 150 * 1024:
            check(lib.vsc_se_block_f32(ptr(x), n, h * w, c, ptr(self.reduce.w), ptr(self.reduce.b), cr, ptr(self.expand.w), ptr(self.expand.b),
                                       ACT["relu"], ACT["hard_sigmoid"], current_stream()))
            return x
        gate = self.expand(self.reduce(avgpool(x), act="relu"), act="hard_sigmoid")
        check(lib.vsc_channel_scale_f32(ptr(x), ptr(gate), n, h * w, c, current_stream()))
        return x


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    assert x.is_cuda and x.dim() == 4
    return x.float().permute(0, 2, 3, 1).contiguous()


class MobileNetV3SmallHip:
    """ClassifyModel (train/models.py:6-17): x [n, 3, h, w] on the GPU -> logits [n, num_classes]."""

    def __init__(self, state_dict: dict, device="cuda"):
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in state_dict.items()}
        self.stem = Conv(sd, "conv_stem", "bn1", 2, device)
        self.blocks = []
        for s, stage in enumerate(MBV3_SMALL):
            for b, (kind, stride, act) in enumerate(stage):
                p = f"blocks.{s}.{b}"
                se = SqueezeExcite(sd, p + ".se", device) if p + ".se.conv_reduce.weight" in sd else None
                if kind == "ds":
                    layers = (DwConv(sd, p + ".conv_dw", p + ".bn1", stride, device), se, Conv(sd, p + ".conv_pw", p + ".bn2", 1, device))
                elif kind == "ir":
                    layers = (Conv(sd, p + ".conv_pw", p + ".bn1", 1, device), DwConv(sd, p + ".conv_dw", p + ".bn2", stride, device), se,
                              Conv(sd, p + ".conv_pwl", p + ".bn3", 1, device))
                else:
                    layers = (Conv(sd, p + ".conv", p + ".bn1", 1, device),)
                self.blocks.append((kind, stride, act, layers))
        self.head = Conv(sd, "conv_head", None, 1, device)
        cw = {"c.weight": _np(sd["classifier.weight"])[:, :, None, None], "c.bias": sd["classifier.bias"]}
        self.classifier = Conv(cw, "c", None, 1, device)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = self.stem(_nhwc(x), act="hard_swish")
        for kind, stride, act, layers in self.blocks:
            if kind == "ds":
                dw, se, pw = layers
                y = dw(x, act=act)
                if se is not None:
                    y = se(y)
                skip = stride == 1 and pw.cout == x.shape[3]
                x = pw(y, residual=x if skip else None)
            elif kind == "ir":
                pw, dw, se, pwl = layers
                y = dw(pw(x, act=act), act=act)
                if se is not None:
                    y = se(y)
                skip = stride == 1 and pwl.cout == x.shape[3]
                x = pwl(y, residual=x if skip else None)
            else:
                x = layers[0](x, act=act)
        x = self.head(avgpool(x), act="hard_swish")
        return self.classifier(x).reshape(x.shape[0], -1)


class _Basic:
    def __init__(self, sd, p, device):
        self.c1 = Conv(sd, p + ".conv1", p + ".bn1", 1, device, pad4=True)
        self.c2 = Conv(sd, p + ".conv2", p + ".bn2", 1, device, pad4=True)

    def __call__(self, x):
        return self.c2(self.c1(x, act="relu"), act="relu", residual=x)


class _Bottleneck:
    def __init__(self, sd, p, device):
        self.c1 = Conv(sd, p + ".conv1", p + ".bn1", 1, device)
        self.c2 = Conv(sd, p + ".conv2", p + ".bn2", 1, device)
        self.c3 = Conv(sd, p + ".conv3", p + ".bn3", 1, device)
        self.down = Conv(sd, p + ".downsample.0", p + ".downsample.1", 1, device) if p + ".downsample.0.weight" in sd else None

    def __call__(self, x):
        sc = self.down(x) if self.down is not None else x
        return self.c3(self.c2(self.c1(x, act="relu"), act="relu"), act="relu", residual=sc)


class _HrModule:
    """HighResolutionModule: BasicBlocks per branch, then every output branch sums every input branch brought to its
    resolution / width (1x1 conv + nearest upsample downwards in index, chains of stride-2 3x3 convs upwards)."""

    def __init__(self, sd, p, nbr, device):
        self.nbr = nbr
        self.branches = [[_Basic(sd, f"{p}.branches.{i}.{k}", device) for k in range(HRNET_BLOCKS)] for i in range(nbr)]
        self.fuse = {}
        for i in range(nbr):
            for j in range(nbr):
                if j > i:
                    self.fuse[i, j] = Conv(sd, f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", 1, device, pad4=True)
                elif j < i:
                    self.fuse[i, j] = [Conv(sd, f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1", 2, device, pad4=True)
                                       for k in range(i - j)]

    def __call__(self, xs):
        ys = []
        for i in range(self.nbr):
            y = xs[i]
            for blk in self.branches[i]:
                y = blk(y)
            ys.append(y)
        outs = []
        for i in range(self.nbr):
            # summation order of the reference: j = 0 .. nbr-1, ReLU after the last term.  The terms j < i end in a convolution
            # (the running sum is its residual); y_i and the upsampled terms j > i are added in ONE pass (vsc_upsample_sum_f32)
            acc = None
            for j in range(i):
                chain = self.fuse[i, j]
                t = ys[j]
                for conv in chain[:-1]:
                    t = conv(t, act="relu")
                last = j == self.nbr - 1
                acc = chain[-1](t) if acc is None else chain[-1](t, residual=acc, act="relu" if last else None)
            terms = [(ys[i], 1)] + [(self.fuse[i, j](ys[j]), 2 ** (j - i)) for j in range(i + 1, self.nbr)]
            if acc is None:
                base, terms = terms[0][0], terms[1:]
            else:
                base = acc
            out = torch.empty_like(ys[i])
            while True:   # three upsampled terms per pass (hrnet_w18 has at most three)
                upsample_sum(base, terms[:3], out, "relu" if len(terms) <= 3 else None)
                terms = terms[3:]
                if not terms:
                    break
                base = out
            outs.append(out)
        return outs


class HRNetRefineHip:
    """HRnet (train/models.py:20-47): x [n, 3, h, w] on the GPU (h, w multiples of 8) -> logits [n, 2, h, w]."""

    SPLIT_FUSE0 = True   # fuse.0 per source at the source's resolution (False: over the materialised 336-channel concatenation)

    def __init__(self, state_dict: dict, device="cuda"):
        sd = {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")}
        self.conv1 = Conv(sd, "conv1", "bn1", 1, device)   # stride 2 in timm, set to 1 by the reference (models.py:25-26)
        self.conv2 = Conv(sd, "conv2", "bn2", 1, device)
        self.layer1 = [_Bottleneck(sd, f"layer1.{k}", device) for k in range(4)]
        # branch widths 18 / 36 / 72 / 144: the 18-wide branch is carried as 20 channels (two zero channels), so that every
        # pixel is 16-byte aligned and the 3 x 3 patch gathers of its 64 full-resolution convolutions are float4 loads
        self.t1 = [Conv(sd, "transition1.0.0", "transition1.0.1", 1, device, pad4=True),
                   Conv(sd, "transition1.1.0.0", "transition1.1.0.1", 2, device, pad4=True)]
        self.stages = []
        for stage, nmod, nbr in HRNET_STAGES:
            grow = Conv(sd, f"transition{stage - 1}.{nbr - 1}.0.0", f"transition{stage - 1}.{nbr - 1}.0.1", 2, device, pad4=True) if stage > 2 else None
            self.stages.append((grow, [_HrModule(sd, f"stage{stage}.{m}", nbr, device) for m in range(nmod)]))
        # torch.cat([stem 64, 18, 36, 72, 144]) -> fuse.0: here the concatenated pixel is [64 | 18 + 2 zeros | 36 | 72 | 144]
        widths = [64, 18, 36, 72, 144]
        real, off = [], 0
        for wd in widths:
            real += list(range(off, off + wd))
            off += wd + (-wd % 4)
        # the 336-wide form of fuse.0 (over the concatenation) is only built when SPLIT_FUSE0 is off (diagnostic / test switch)
        self._fuse0_args = (state_dict, (real, off), device)
        self._fuse0 = None
        # fuse.0 is a 1 x 1 convolution over torch.cat of nearest-upsampled features, and a 1 x 1 convolution commutes with nearest
        # upsampling: conv(cat_s up(x_s)) = sum_s up(conv_s(x_s)) with conv_s = the weight columns of source s.  Per source at its OWN
        # resolution (the 72- / 144-wide branches at 1/16 and 1/64 of the pixels), no 336-channel tensor (1.08 GB written and read):
        # 1.08 -> 0.45 ms per pass.  The bias rides on the first term; the partial sums are added in source order.
        w0, b0 = _np(state_dict["fuse.0.weight"]), _np(state_dict["fuse.0.bias"])
        self.fuse0_parts, lo = [], 0
        for i, wd in enumerate(widths):
            part = {"p.weight": w0[:, lo:lo + wd], "p.bias": b0 if i == 0 else np.zeros_like(b0)}
            self.fuse0_parts.append(Conv(part, "p", None, 1, device, pad4=True))
            lo += wd
        self.fuse2 = Conv(state_dict, "fuse.2", None, 1, device, pad4=True)
        self.classes = int(_np(state_dict["fuse.2.weight"]).shape[0])

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = _nhwc(x)
        n, h, w, _ = x.shape
        assert h % 8 == 0 and w % 8 == 0, "HRNet input sides must be multiples of 8"
        stem = self.conv1(x, act="relu")
        x = self.conv2(stem, act="relu")
        for blk in self.layer1:
            x = blk(x)
        xs = [self.t1[0](x, act="relu"), self.t1[1](x, act="relu")]
        for grow, modules in self.stages:
            if grow is not None:
                xs = xs + [grow(xs[-1], act="relu")]
            for m in modules:
                xs = m(xs)
        if self.SPLIT_FUSE0:
            srcs = [stem] + xs
            base = self.fuse0_parts[1](srcs[1], residual=self.fuse0_parts[0](srcs[0]))          # both at full resolution
            terms = [(self.fuse0_parts[i](srcs[i]), 2 ** (i - 1)) for i in range(2, len(srcs))]   # 1/2, 1/4, 1/8
            fused = upsample_sum(base, terms, torch.empty_like(base), "relu")
        else:
            widths = [stem.shape[3]] + [y.shape[3] for y in xs]
            cat = torch.empty((n, h, w, sum(widths)), dtype=torch.float32, device=x.device)   # torch.cat of the upsampled features
            off = 0
            for i, y in enumerate([stem] + xs):
                upsample_into(y, cat, 1 if i < 2 else 2 ** (i - 1), off, False, None)
                off += widths[i]
            if self._fuse0 is None:
                sd0, cmap, dev0 = self._fuse0_args
                self._fuse0 = Conv(sd0, "fuse.0", None, 1, dev0, pad4=True, cin_map=cmap)
            fused = self._fuse0(cat, act="relu")
        y = self.fuse2(fused)[..., : self.classes]
        return y.permute(0, 3, 1, 2).contiguous()


def match_classify_probability(models, feature: torch.Tensor) -> torch.Tensor:
    """infer_matching.py:165-168: mean over the classifier models of softmax(model(x))[:, 1].  feature [n, 3, h, w] (GPU)."""
    return sum(m(feature).softmax(dim=1)[:, 1] for m in models) / len(models)


def match_refine_probability(models, feature: torch.Tensor) -> torch.Tensor:
    """infer_matching.py:183-193: per model the class softmax of model(x) and of model(x^T)^T averaged, then the mean over
    models.  feature [n, 3, h, w] (GPU) -> [n, 2, h, w]."""
    preds = []
    for m in models:
        p = m(feature).softmax(dim=1)
        pt = m(feature.transpose(3, 2).contiguous()).softmax(dim=1).transpose(3, 2)
        preds.append((p + pt) / 2)
    return sum(preds) / len(preds)
