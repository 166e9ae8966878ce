"""oracle/cnn_oracle.py on CPU: shapes, the classic identities its structure must satisfy, and state-dict key coverage
(every parameter of the synthetic timm-named checkpoints is consumed -- a silently skipped layer would show up here)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cnn_synth  # noqa: E402
from oracle import cnn_oracle  # noqa: E402


class _Tracking(dict):
    def __init__(self, d):
        super().__init__(d)
        self.used = set()

    def __getitem__(self, k):
        self.used.add(k)
        return super().__getitem__(k)

    def get(self, k, default=None):
        if k in self:
            self.used.add(k)
        return super().get(k, default)


def test_mobilenetv3_small_oracle_shapes_and_key_coverage():
    sd = {k[len("model."):]: v for k, v in cnn_synth.mobilenetv3_small_state(1).items()}
    tracked = _Tracking(sd)
    x = cnn_synth.similarity_maps(2, 3, 160, 160)
    with torch.no_grad():
        y = cnn_oracle.mobilenetv3_small(tracked, x)
    assert y.shape == (3, 2) and torch.isfinite(y).all()
    unused = {k for k in sd if k not in tracked.used and not k.endswith("num_batches_tracked")}
    assert not unused, sorted(unused)[:5]
    n_params = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert abs(n_params - 1.52e6) < 0.05e6      # timm: mobilenetv3_small_100 with a 2-class head has 1.52 M parameters


def test_hrnet_refine_oracle_shapes_and_key_coverage():
    sd = cnn_synth.hrnet_refine_state(3)
    inner = _Tracking({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")})
    x = cnn_synth.similarity_maps(4, 1, 32, 48)
    with torch.no_grad():
        feats = cnn_oracle.hrnet_w18_features(inner, x, 1)
        y = cnn_oracle.hrnet_refine(sd, x)
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 48), (18, 32, 48), (36, 16, 24), (72, 8, 12), (144, 4, 6)]
    assert y.shape == (1, 2, 32, 48) and torch.isfinite(y).all()
    unused = {k for k in inner if k not in inner.used and not k.endswith("num_batches_tracked")}
    assert not unused, sorted(unused)[:5]
    n_backbone = sum(v.numel() for k, v in inner.items() if "running" not in k and "num_batches" not in k)
    assert abs(n_backbone - 9.3e6) < 0.4e6       # hrnet_w18 without incre / classifier modules


def test_refine_probability_is_symmetric_in_transposition():
    """infer_matching.py:186-191 averages model(x) with model(x^T)^T: the result for x^T is the transpose of the result for x."""
    sd = cnn_synth.hrnet_refine_state(5)
    x = cnn_synth.similarity_maps(6, 1, 16, 16)
    with torch.no_grad():
        p = cnn_oracle.match_refine_probability([sd], x)
        pt = cnn_oracle.match_refine_probability([sd], x.transpose(3, 2))
    assert torch.allclose(p.sum(1), torch.ones_like(p.sum(1)), atol=1e-6)
    assert torch.allclose(pt.transpose(3, 2), p, atol=1e-5)
