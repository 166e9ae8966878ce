"""Deterministic random state dicts with timm's parameter names and shapes for the matching track's two networks
(mobilenetv3_small_100 with 2 classes; the reference's HRnet wrapper around hrnet_w18 features + fuse head).  Test data only:
the reference ships no checkpoints for them (VSC22-Matching-Track-1st/checkpoints is empty)."""
import numpy as np
import torch


class _Gen:
    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.sd = {}

    def conv(self, name, cout, cin, k, bias=False, groups=1):
        fan = cin // groups * k * k
        self.sd[name + ".weight"] = torch.from_numpy((self.rng.randn(cout, cin // groups, k, k) * np.sqrt(2.0 / fan)).astype(np.float32))
        if bias:
            self.sd[name + ".bias"] = torch.from_numpy((self.rng.randn(cout) * 0.1).astype(np.float32))

    def bn(self, name, c):
        self.sd[name + ".weight"] = torch.from_numpy(self.rng.uniform(0.6, 1.4, c).astype(np.float32))
        self.sd[name + ".bias"] = torch.from_numpy((self.rng.randn(c) * 0.1).astype(np.float32))
        self.sd[name + ".running_mean"] = torch.from_numpy((self.rng.randn(c) * 0.1).astype(np.float32))
        self.sd[name + ".running_var"] = torch.from_numpy(self.rng.uniform(0.5, 1.5, c).astype(np.float32))
        self.sd[name + ".num_batches_tracked"] = torch.tensor(100)


def _div8(v):
    new = max(8, int(v + 4) // 8 * 8)
    return new + 8 if new < 0.9 * v else new


def mobilenetv3_small_state(seed=0, num_classes=2, prefix="model."):
    g = _Gen(seed)
    g.conv("conv_stem", 16, 3, 3)
    g.bn("bn1", 16)
    # (kind, kernel, exp, out, se) per block, grouped by stage: timm arch_def of mobilenetv3_small_100
    plan = [[("ds", 3, 16, 16, True)],
            [("ir", 3, 72, 24, False), ("ir", 3, 88, 24, False)],
            [("ir", 5, 96, 40, True), ("ir", 5, 240, 40, True), ("ir", 5, 240, 40, True)],
            [("ir", 5, 120, 48, True), ("ir", 5, 144, 48, True)],
            [("ir", 5, 288, 96, True), ("ir", 5, 576, 96, True), ("ir", 5, 576, 96, True)],
            [("cn", 1, 0, 576, False)]]
    cin = 16
    for s, stage in enumerate(plan):
        for b, (kind, k, exp, cout, se) in enumerate(stage):
            p = f"blocks.{s}.{b}"
            if kind == "ds":
                g.conv(p + ".conv_dw", cin, cin, k, groups=cin)
                g.bn(p + ".bn1", cin)
                if se:
                    rd = _div8(cin * 0.25)
                    g.conv(p + ".se.conv_reduce", rd, cin, 1, bias=True)
                    g.conv(p + ".se.conv_expand", cin, rd, 1, bias=True)
                g.conv(p + ".conv_pw", cout, cin, 1)
                g.bn(p + ".bn2", cout)
            elif kind == "ir":
                g.conv(p + ".conv_pw", exp, cin, 1)
                g.bn(p + ".bn1", exp)
                g.conv(p + ".conv_dw", exp, exp, k, groups=exp)
                g.bn(p + ".bn2", exp)
                if se:
                    rd = _div8(exp * 0.25)
                    g.conv(p + ".se.conv_reduce", rd, exp, 1, bias=True)
                    g.conv(p + ".se.conv_expand", exp, rd, 1, bias=True)
                g.conv(p + ".conv_pwl", cout, exp, 1)
                g.bn(p + ".bn3", cout)
            else:
                g.conv(p + ".conv", cout, cin, 1)
                g.bn(p + ".bn1", cout)
            cin = cout
    g.conv("conv_head", 1024, 576, 1, bias=True)
    g.sd["classifier.weight"] = torch.from_numpy((g.rng.randn(num_classes, 1024) * 0.05).astype(np.float32))
    g.sd["classifier.bias"] = torch.from_numpy((g.rng.randn(num_classes) * 0.1).astype(np.float32))
    return {prefix + k: v for k, v in g.sd.items()}


def hrnet_refine_state(seed=0):
    g = _Gen(seed)
    g.conv("model.conv1", 64, 3, 3)
    g.bn("model.bn1", 64)
    g.conv("model.conv2", 64, 64, 3)
    g.bn("model.bn2", 64)
    cin = 64
    for k in range(4):
        p = f"model.layer1.{k}"
        g.conv(p + ".conv1", 64, cin, 1)
        g.bn(p + ".bn1", 64)
        g.conv(p + ".conv2", 64, 64, 3)
        g.bn(p + ".bn2", 64)
        g.conv(p + ".conv3", 256, 64, 1)
        g.bn(p + ".bn3", 256)
        if k == 0:
            g.conv(p + ".downsample.0", 256, cin, 1)
            g.bn(p + ".downsample.1", 256)
        cin = 256
    widths = (18, 36, 72, 144)
    g.conv("model.transition1.0.0", 18, 256, 3)
    g.bn("model.transition1.0.1", 18)
    g.conv("model.transition1.1.0.0", 36, 256, 3)
    g.bn("model.transition1.1.0.1", 36)
    for stage, nmod, nbr in ((2, 1, 2), (3, 4, 3), (4, 3, 4)):
        if stage > 2:
            q = f"model.transition{stage - 1}.{nbr - 1}.0"
            g.conv(q + ".0", widths[nbr - 1], widths[nbr - 2], 3)
            g.bn(q + ".1", widths[nbr - 1])
        for m in range(nmod):
            p = f"model.stage{stage}.{m}"
            for i in range(nbr):
                for k in range(4):
                    q = f"{p}.branches.{i}.{k}"
                    g.conv(q + ".conv1", widths[i], widths[i], 3)
                    g.bn(q + ".bn1", widths[i])
                    g.conv(q + ".conv2", widths[i], widths[i], 3)
                    g.bn(q + ".bn2", widths[i])
            for i in range(nbr):
                for j in range(nbr):
                    if j > i:
                        g.conv(f"{p}.fuse_layers.{i}.{j}.0", widths[i], widths[j], 1)
                        g.bn(f"{p}.fuse_layers.{i}.{j}.1", widths[i])
                    elif j < i:
                        for k in range(i - j):
                            cout = widths[i] if k == i - j - 1 else widths[j]
                            g.conv(f"{p}.fuse_layers.{i}.{j}.{k}.0", cout, widths[j], 3)
                            g.bn(f"{p}.fuse_layers.{i}.{j}.{k}.1", cout)
    g.conv("fuse.0", 64, 64 + sum(widths), 1, bias=True)
    g.conv("fuse.2", 2, 64, 1, bias=True)
    return g.sd


def similarity_maps(seed, n, h, w):
    """[n, 3, h, w]: one random 'frame x frame similarity' map per sample with a diagonal copy segment, stacked three times
    as MatchClassifyDataset / MatchRefineDataset do (infer/src/dataset.py:106-144)."""
    rng = np.random.RandomState(seed)
    m = (rng.randn(n, h, w) * 0.08).astype(np.float32)
    for i in range(n):
        y0, x0, ln = rng.randint(0, h // 2), rng.randint(0, w // 2), rng.randint(4, min(h, w) // 2)
        for t in range(ln):
            m[i, y0 + t, x0 + t] += 0.7
    return torch.from_numpy(np.repeat(m[:, None], 3, axis=1))
