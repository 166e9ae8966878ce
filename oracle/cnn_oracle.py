"""TEST INFRASTRUCTURE -- CPU oracle (torch fp32) of the matching track's two networks.  Only tests/, smoke() and bench's
cpu_baseline may import this module; nothing under vsc22-submission_amd/ does.

Reference: VSC22-Matching-Track-1st/train/models.py
  :6-17   ClassifyModel  = timm.create_model('mobilenetv3_small_100', num_classes=2)         (pair classifier, 3x160x160)
  :20-47  HRnet          = timm.create_model('hrnet_w18', features_only=True, feature_location='', out_indices=(0..4)),
                            conv1 / conv2 strides set to 1, nearest upsampling of the four branch outputs to the input
                            resolution, concat with the stem feature (64 + 18 + 36 + 72 + 144 = 334 channels),
                            fuse = Conv1x1(334, 64) - ReLU - Conv1x1(64, 2)                    (refinement net, 3x224x224)
and their use in infer/infer_matching.py:158-204 (softmax over the 2 classes; the refinement net is also run on the
transposed map and the two probability maps are averaged).

PARITY UNPINNED.  timm is a third-party dependency of the reference that is neither vendored under /root/reference nor
installed in this image (torchvision is absent too), and the reference ships no checkpoints or golden outputs for these
models (checkpoints/ is empty).  The two architectures are therefore RESTATED here from timm's published definitions
(timm 0.6.x: mobilenetv3.py `mobilenetv3_small_100` arch_def, _efficientnet_blocks.py DepthwiseSeparableConv /
InvertedResidual / SqueezeExcite; hrnet.py `hrnet_w18` cfg, Bottleneck / BasicBlock / HighResolutionModule /
HighResolutionNetFeatures) with timm's parameter names, so that a timm state dict loads unchanged; what the tests prove is
that the HIP path computes the same function as this restatement, not that the restatement equals timm bit for bit.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5

# timm arch_def of mobilenetv3_small_100: (stage, block) -> (kind, stride, activation, squeeze-excite)
#   ds_r1_k3_s2_e1_c16_se0.25_nre | ir_r1_k3_s2_e4.5_c24_nre, ir_r1_k3_s1_e3.67_c24_nre |
#   ir_r1_k5_s2_e4_c40_se0.25, ir_r2_k5_s1_e6_c40_se0.25 | ir_r2_k5_s1_e3_c48_se0.25 | ir_r3_k5_s2_e6_c96_se0.25 | cn_r1_k1_s1_c576
MBV3_SMALL = [
    [("ds", 2, "relu")],
    [("ir", 2, "relu"), ("ir", 1, "relu")],
    [("ir", 2, "hard_swish"), ("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")],
    [("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")],
    [("ir", 2, "hard_swish"), ("ir", 1, "hard_swish"), ("ir", 1, "hard_swish")],
    [("cn", 1, "hard_swish")],
]
# hrnet_w18: stage2 1 module x 2 branches, stage3 4 x 3, stage4 3 x 4; BasicBlock x 4 per branch; widths 18/36/72/144
HRNET_W18 = {"modules": (1, 4, 3), "branches": (2, 3, 4), "blocks": 4}


def _act(x, name):
    if name == "relu":
        return F.relu(x)
    if name == "hard_swish":
        return F.hardswish(x)
    if name == "hard_sigmoid":
        return F.hardsigmoid(x)
    assert name is None or name == "none", name
    return x


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv(x, sd, p, stride=1, groups=1):
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=w.shape[-1] // 2, groups=groups)


def _se(x, sd, p):
    s = x.mean((2, 3), keepdim=True)
    s = F.relu(_conv(s, sd, p + ".conv_reduce"))
    return x * F.hardsigmoid(_conv(s, sd, p + ".conv_expand"))


def mobilenetv3_small(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """sd: timm mobilenetv3_small_100 state dict (float32 tensors); x [n, 3, h, w] -> logits [n, num_classes]."""
    x = _act(_bn(_conv(x, sd, "conv_stem", 2), sd, "bn1"), "hard_swish")
    for s, stage in enumerate(MBV3_SMALL):
        for b, (kind, stride, act) in enumerate(stage):
            p = f"blocks.{s}.{b}"
            if kind == "ds":
                c = x.shape[1]
                y = _act(_bn(_conv(x, sd, p + ".conv_dw", stride, groups=c), sd, p + ".bn1"), act)
                if p + ".se.conv_reduce.weight" in sd:
                    y = _se(y, sd, p + ".se")
                y = _bn(_conv(y, sd, p + ".conv_pw"), sd, p + ".bn2")
                x = x + y if stride == 1 and y.shape == x.shape else y
            elif kind == "ir":
                y = _act(_bn(_conv(x, sd, p + ".conv_pw"), sd, p + ".bn1"), act)
                y = _act(_bn(_conv(y, sd, p + ".conv_dw", stride, groups=y.shape[1]), sd, p + ".bn2"), act)
                if p + ".se.conv_reduce.weight" in sd:
                    y = _se(y, sd, p + ".se")
                y = _bn(_conv(y, sd, p + ".conv_pwl"), sd, p + ".bn3")
                x = x + y if stride == 1 and y.shape == x.shape else y
            else:
                x = _act(_bn(_conv(x, sd, p + ".conv"), sd, p + ".bn1"), act)
    x = x.mean((2, 3), keepdim=True)
    x = _act(_conv(x, sd, "conv_head"), "hard_swish")
    return F.linear(x.flatten(1), sd["classifier.weight"], sd["classifier.bias"])


def _basic_block(x, sd, p):
    y = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    y = _bn(_conv(y, sd, p + ".conv2"), sd, p + ".bn2")
    return F.relu(y + x)


def _bottleneck(x, sd, p):
    y = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    y = F.relu(_bn(_conv(y, sd, p + ".conv2"), sd, p + ".bn2"))
    y = _bn(_conv(y, sd, p + ".conv3"), sd, p + ".bn3")
    sc = x
    if p + ".downsample.0.weight" in sd:
        sc = _bn(_conv(x, sd, p + ".downsample.0"), sd, p + ".downsample.1")
    return F.relu(y + sc)


def _hr_module(xs, sd, p, nblocks):
    nb = len(xs)
    ys = []
    for i in range(nb):
        y = xs[i]
        for k in range(nblocks):
            y = _basic_block(y, sd, f"{p}.branches.{i}.{k}")
        ys.append(y)
    out = []
    for i in range(nb):
        acc = None
        for j in range(nb):
            if j == i:
                t = ys[j]
            elif j > i:
                q = f"{p}.fuse_layers.{i}.{j}"
                t = _bn(_conv(ys[j], sd, q + ".0"), sd, q + ".1")
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = ys[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _bn(_conv(t, sd, q + ".0", 2), sd, q + ".1")
                    if k < i - j - 1:
                        t = F.relu(t)
            acc = t if acc is None else acc + t
        out.append(F.relu(acc))
    return out


def hrnet_w18_features(sd: dict, x: torch.Tensor, stem_stride: int = 1):
    """timm hrnet_w18 features_only, feature_location='', out_indices (0, 1, 2, 3, 4), keys as in the timm state dict.
    stem_stride = 1 is the reference's modification (train/models.py:25-26)."""
    x = F.relu(_bn(_conv(x, sd, "conv1", stem_stride), sd, "bn1"))
    feats = [x]
    x = F.relu(_bn(_conv(x, sd, "conv2", stem_stride), sd, "bn2"))
    for k in range(4):
        x = _bottleneck(x, sd, f"layer1.{k}")
    xs = [F.relu(_bn(_conv(x, sd, "transition1.0.0"), sd, "transition1.0.1")),
          F.relu(_bn(_conv(x, sd, "transition1.1.0.0", 2), sd, "transition1.1.0.1"))]
    for stage, (nmod, nbr) in enumerate(zip(HRNET_W18["modules"], HRNET_W18["branches"]), start=2):
        if stage > 2:   # a new, half-resolution branch grows out of the last one
            q = f"transition{stage - 1}.{nbr - 1}.0"
            xs = xs + [F.relu(_bn(_conv(xs[-1], sd, q + ".0", 2), sd, q + ".1"))]
        for m in range(nmod):
            xs = _hr_module(xs, sd, f"stage{stage}.{m}", HRNET_W18["blocks"])
    return feats + xs


def hrnet_refine(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """The reference's HRnet module (train/models.py:20-47): sd holds `model.*` (timm features) and `fuse.{0,2}.*`.
    x [n, 3, h, w] (h, w multiples of 8) -> logits [n, 2, h, w]."""
    inner = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    ys = hrnet_w18_features(inner, x, 1)
    ups = [ys[0], ys[1]] + [F.interpolate(y, scale_factor=2 ** i, mode="nearest") for i, y in enumerate(ys[2:], start=1)]
    y = torch.cat(ups, dim=1)
    y = F.relu(F.conv2d(y, sd["fuse.0.weight"], sd["fuse.0.bias"]))
    return F.conv2d(y, sd["fuse.2.weight"], sd["fuse.2.bias"])


def match_refine_probability(models: list, feature: torch.Tensor) -> torch.Tensor:
    """infer_matching.py:183-193: per model softmax over the class axis of model(x) and of model(x^T)^T, averaged; then the
    mean over models.  -> [n, 2, h, w]."""
    preds = []
    for sd in models:
        p = hrnet_refine(sd, feature).softmax(dim=1)
        pt = hrnet_refine(sd, feature.transpose(3, 2)).softmax(dim=1).transpose(3, 2)
        preds.append((p + pt) / 2)
    return sum(preds) / len(preds)


def match_classify_probability(models: list, feature: torch.Tensor) -> torch.Tensor:
    """infer_matching.py:165-168: mean over models of softmax(model(x))[:, 1]."""
    return sum(mobilenetv3_small(sd, feature).softmax(dim=1)[:, 1] for sd in models) / len(models)
