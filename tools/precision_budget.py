"""Which bf16 rounding points of the ViT path cost the end-to-end uAP parity (build-container dev tool, CPU only).

The HIP encoder rounds to bf16 at a fixed set of points (MFMA inputs bf16, everything else fp32):
    W   weights of every Linear                          (rounded once, at load)
    px  the normalised pixels entering the patch GEMM
    y   LayerNorm outputs (qkv / fc1 GEMM inputs)
    qkv the qkv GEMM's output (attention inputs)
    p   softmax probabilities (PV MFMA input)
    ctx the attention output (proj GEMM input)
    h   GELU(fc1) (fc2 GEMM input)
This tool restates the forward in torch fp32 with any subset of those roundings switched on, runs the 816 frames of the uAP fixture
(tools/synth_videos.py) through it, swaps the result in for the fixture's fp32 ViT descriptors, and pushes the fixture's chain
(tests/golden/gen_uap_e2e_golden.chain: normalise / PCA / duplicate filter / score norm / search / candidates) to the uAP:

    python tools/precision_budget.py all none W y,h qkv,p,ctx ...

prints, per subset: descriptor max / mean |d| against the fixture, max |score d| over the fp32 top-200 candidates, rank inversions
among them, |d uAP|.  It needs no reference code (the fixture holds the fp32 descriptors); the fp32 run ("none") must reproduce the
fixture to ~1e-6, which checks the restatement."""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vsc22-submission_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from tools import synth, synth_videos  # noqa: E402

POINTS = ("W", "px", "y", "qkv", "p", "ctx", "h", "Wpatch", "Wqkv", "Wproj", "Wfc1", "Wfc2")     # W = all five weight classes


LOW = torch.float16 if os.environ.get("PB_DTYPE", "bf16") == "fp16" else torch.bfloat16      # PB_DTYPE=fp16: the same points rounded to fp16


def bf(x):
    return x.to(LOW).float()


def vit_forward(w, cfg, frames, on, layers_on=None):
    """frames [n,3,H,W] fp32 normalised -> descriptors [n,out] (not normalised).  `on` = set of POINTS; `layers_on` = set of layer indices
    the per-layer roundings apply to (None: all)."""
    r = lambda name, x, layer=None: bf(x) if (name in on and (layer is None or layers_on is None or layer in layers_on)) else x
    W = lambda name, layer=None: r("W", r("W" + name.split(".")[-2], w[name], layer), layer)
    n, d = frames.shape[0], cfg.width
    ps = cfg.patch_size
    g = cfg.image_size // ps
    x = frames.reshape(n, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(n, g * g, 3 * ps * ps)
    x = r("px", x) @ W("patch.weight").reshape(d, -1).t() + w["patch.bias"]
    x = torch.cat([w["cls"].reshape(1, 1, d).expand(n, 1, d), x], dim=1) + w["pos"].reshape(1, -1, d)
    h_, dh, t = cfg.heads, d // cfg.heads, x.shape[1]
    for i in range(cfg.layers):
        b = f"blocks.{i}."
        y = r("y", F.layer_norm(x, (d,), w[b + "ln1.weight"], w[b + "ln1.bias"], cfg.ln_eps), i)
        qkv = r("qkv", y @ W(b + "qkv.weight", i).t() + w[b + "qkv.bias"], i)
        q, k, v = qkv.reshape(n, t, 3, h_, dh).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        pr = r("p", e, i)                                   # the kernel packs exp() to bf16 and divides by the sum of the ROUNDED values
        a = (pr @ v) / pr.sum(dim=-1, keepdim=True)
        a = r("ctx", a.permute(0, 2, 1, 3).reshape(n, t, d), i)
        x = x + a @ W(b + "proj.weight", i).t() + w[b + "proj.bias"]
        y = r("y", F.layer_norm(x, (d,), w[b + "ln2.weight"], w[b + "ln2.bias"], cfg.ln_eps), i)
        hh = r("h", F.gelu(y @ W(b + "fc1.weight", i).t() + w[b + "fc1.bias"]), i)
        x = x + hh @ W(b + "fc2.weight", i).t() + w[b + "fc2.bias"]
    tok = F.layer_norm(x, (d,), w["ln_post.weight"], w["ln_post.bias"], cfg.ln_eps)
    pooled = tok.clamp(min=1e-6).pow(cfg.gem_p).mean(dim=1).pow(1.0 / cfg.gem_p)
    return pooled @ w["head.weight"].t() + w["head.bias"]


def evaluate(g, data, desc_vit, desc_swin=None):
    import gen_uap_e2e_golden as gen
    from vsc.metrics import CandidatePair, average_precision
    gen.PCA_DIM = int(g["pca_components"].shape[0])
    cands, _, _, _ = gen.chain(data, [g["desc_swin"] if desc_swin is None else desc_swin, desc_vit])
    ref = list(zip(g["cand_query"].tolist(), g["cand_ref"].tolist(), g["cand_score"].tolist()))
    uap = average_precision([CandidatePair(q, r, 1.0) for q, r in data["gt"]], [CandidatePair(q, r, s) for q, r, s in cands]).ap
    pos = {(q, r): i for i, (q, r, _) in enumerate(cands)}
    top = [(q, r, s) for q, r, s in ref[:200] if (q, r) in pos]
    order = [pos[(q, r)] for q, r, _ in top]
    inv = sum(1 for i in range(len(order)) for j in range(i + 1, len(order)) if order[i] > order[j])
    serr = max(abs(s - cands[pos[(q, r)]][2]) for q, r, s in top)
    return uap, inv, serr


def main(argv):
    from vsc_hip.config import get_config
    torch.set_num_threads(8)
    g = np.load(os.path.join(ROOT, "tests", "golden", "uap_e2e.npz"))
    data = synth_videos.make(int(g["seed"]))
    assert data["fingerprint"] == str(g["fingerprint"])
    cfg = get_config(str(g["vit_preset"]))
    w = {k: torch.from_numpy(v) for k, v in synth.encoder_weights(int(g["vit_weights_seed"]), cfg).items()}
    allf = np.concatenate([f for grp in ("refs", "norm", "queries") for _, f in data[grp]])
    x8 = synth_videos.resize_u8(allf, cfg.image_size)
    gold = g["desc_vit"]
    gn = gold / np.linalg.norm(gold, axis=1, keepdims=True)
    for spec in argv or ["none", "all"]:
        spec, _, lay = spec.partition("@")                    # e.g.  y,h@0-3  = those roundings in layers 0..3 only
        on = set(POINTS) if spec == "all" else set() if spec == "none" else set(spec.split(","))
        assert on <= set(POINTS), on
        layers_on = None
        if lay:
            lo, _, hi = lay.partition("-")
            layers_on = set(range(int(lo), int(hi or lo) + 1))
        outs = []
        with torch.no_grad():
            for lo in range(0, len(x8), 24):
                t = torch.from_numpy(x8[lo:lo + 24]).permute(0, 3, 1, 2).float().div(255.0)
                outs.append(vit_forward(w, cfg, (t - 0.5) / 0.5, on, layers_on).numpy())
        d = np.concatenate(outs)
        dn = d / np.linalg.norm(d, axis=1, keepdims=True)
        uap, inv, serr = evaluate(g, data, d)
        print(f"{spec + ('@' + lay if lay else ''):24s} desc max {np.abs(dn - gn).max():.2e} mean {np.abs(dn - gn).mean():.2e} | top-200 max score d {serr:.2e} "
              f"inversions {inv:4d} | uAP {uap:.6f} d {uap - float(g['uap']):+.2e}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
