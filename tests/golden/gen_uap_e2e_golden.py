"""Generates tests/golden/uap_e2e.npz: the descriptor track's chain END TO END in fp32 through the reference's own classes, on the
miniature data set of tools/synth_videos.py -- the yardstick of tests/test_gpu_uap_e2e.py (|uAP_hip - uAP_fp32| <= 1e-3, the
criterion `north_star` ends on).  Build container only (executes reference definitions: VSC_RUN_REFERENCE_CODE=1):

    VSC_RUN_REFERENCE_CODE=1 python tests/golden/gen_uap_e2e_golden.py [--cache /tmp/uap_e2e_desc.npz]

The chain (reference files under /root/reference/VSC22-Descriptor-Track-1st/):
  frames u8 -> Resize(bicubic) + ToTensor + Normalize(0.5, 0.5)                       infer/extract_query_feats.py:106-129, infer/src/transform.py:37-42
  -> SwinTransformerV2 (Swin-V2-B / 256)                                              train/train_v115/torch2scripts.py:70-657   (the class itself, from source)
  -> VIT (HF ViTModel + GeM + output_proj, ViT-B/16 / 224)                            train/train_v115/.../backbones/vit.py:12-54 (the class itself, from source)
  refs:    normalize per model, concatenate, PCA                                      infer/concat_pca_sn.py:42-68 (sklearn PCA fitted on the other split's references)
           ref_score_normalize against the other split                                infer/vsc/baseline/score_normalization.py:150-192
  queries: normalize per model, concatenate, drop near-duplicate frames, PCA          infer/extract_query_feats.py:176-211
           query_score_normalize (nk = 1, beta = 1.2, low-variance dimension)         infer/vsc/baseline/score_normalization.py:108-148, extract_query_feats.py:247-250
  search:  global-threshold k-NN over all frame pairs, max per video pair, best first infer/vsc/index.py:100-165, infer/vsc/candidates.py:24-40,
           first 25 per query                                                          infer/vsc/baseline/sscd_baseline.py:89-103
  uAP:     the reference's own average_precision (imported from its file)             infer/vsc/metrics.py:423-494
faiss is absent from this image: its Flat inner-product search is the float32 matrix product below (`oracle.knn_oracle`, pinned on the
reference's unit-test vectors); sklearn's `normalize` / `PCA` are the reference's own calls.  The video-score gate is not part of this
fixture (every video is treated as accepted; the gate has its own fixture, tests/golden/vsm_tiny_vsm.npz).

Stored: seeds + the frames' sha256, the fitted PCA (the "checkpoint" the HIP run loads), the fp32 candidate list, uAP, and the fp32
per-model descriptors of every frame (for error statistics on the same frames)."""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "vsc22-submission_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from tools import synth, synth_videos  # noqa: E402

SEED = 2022
SWIN_PRESET, SWIN_WEIGHTS = "swinv2_base_256", 5
VIT_PRESET, VIT_WEIGHTS = "vit_b16_224", 7
PCA_DIM = 128
FRAME_THRESHOLD = 0.975        # extract_query_feats.py:55
NK, BETA = 1, 1.2              # :56-57
RETRIEVE_PER_QUERY, CANDIDATES_PER_QUERY = 1200.0, 25.0    # sscd_baseline.py:91-92
OUT = os.path.join(HERE, "uap_e2e.npz")


def reference_models():
    import _reference_classes as refc
    import check_golden_against_reference as chk
    from vsc_hip.config import get_config
    from vsc_hip.swin_config import get_swin_config
    scfg = get_swin_config(SWIN_PRESET)
    ns = refc.load_definitions(refc.SWIN_SRC)
    swin = ns["SwinTransformerV2"](img_size=scfg.image_size, patch_size=scfg.patch_size, window_size=scfg.window_size,
                                   num_heads=list(scfg.heads), embed_dim=scfg.embed_dim, depths=list(scfg.depths),
                                   pretrained_window_sizes=list(scfg.pretrained_window_sizes), mlp_ratio=float(scfg.mlp_ratio),
                                   drop_path_rate=0.2, pretrained=None, output_dim=scfg.out_dim, p=scfg.gem_p).eval()
    res = swin.load_state_dict({k: chk._t(v) for k, v in synth.swin_weights(SWIN_WEIGHTS, scfg).items()}, strict=False)
    assert not res.unexpected_keys
    vcfg = get_config(VIT_PRESET)
    vit = chk.reference_vit(vcfg, synth.encoder_weights(VIT_WEIGHTS, vcfg))
    return [(swin, scfg.image_size), (vit, vcfg.image_size)]


def encode(model, size, frames_u8, batch=16):
    x8 = synth_videos.resize_u8(frames_u8, size)
    outs = []
    with torch.no_grad():
        for lo in range(0, len(x8), batch):
            t = torch.from_numpy(x8[lo:lo + batch]).permute(0, 3, 1, 2).float().div(255.0)     # ToTensor
            outs.append(model((t - 0.5) / 0.5).numpy())                                       # Normalize(0.5, 0.5)
    return np.concatenate(outs)


def descriptors(data, cache):
    if cache and os.path.exists(cache):
        z = np.load(cache)
        if str(z["fingerprint"]) == data["fingerprint"]:
            return [z["swin"], z["vit"]]
    models = reference_models()
    allf = np.concatenate([f for grp in ("refs", "norm", "queries") for _, f in data[grp]])
    out = [encode(m, s, allf) for m, s in models]
    if cache:
        np.savez(cache, fingerprint=data["fingerprint"], swin=out[0], vit=out[1])
    return out


def greedy_select(sim):
    """extract_query_feats.py:200-207 (sim: frame x frame cosine with the diagonal removed)"""
    removed = []
    for i in sim.mean(0).argsort()[::-1]:
        if i in removed:
            continue
        for j in np.where(sim[i] > FRAME_THRESHOLD)[0]:
            removed.append(j)
    return [i for i in range(len(sim)) if i not in removed]


def chain(data, desc):
    """fp32 descriptors of every frame (per model, in refs / norm / queries order) -> candidates, as documented above."""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import normalize
    from oracle import knn_oracle
    F = synth_videos.FRAMES
    nr, nn_, nq = len(data["refs"]), len(data["norm"]), len(data["queries"])
    cat = np.concatenate([normalize(d) for d in desc], axis=1)                    # concat_pca_sn.py:59-60 / extract_query_feats.py:176-181
    ref_cat, norm_cat, q_cat = cat[:nr * F], cat[nr * F:(nr + nn_) * F], cat[(nr + nn_) * F:]
    pca = PCA(n_components=PCA_DIM, random_state=2023).fit(norm_cat)             # concat_pca_sn.py:42-54: fitted on the train references
    refs, norm = pca.transform(ref_cat).astype(np.float32), pca.transform(norm_cat).astype(np.float32)
    low_var_dim = int(norm.var(axis=0).argmin())                                  # score_normalization.py:167-168, src/utils.py:2-5
    drop = lambda x: normalize(np.delete(x, low_var_dim, axis=1))                 # :169-178
    refs_sn = np.concatenate([drop(refs), np.ones((len(refs), 1), np.float32)], axis=1)       # :186-190
    norm_bank = drop(norm)
    q_feats, q_owner, kept = [], [], {}
    for v in range(nq):
        f = q_cat[v * F:(v + 1) * F]
        feat = f / np.linalg.norm(f, axis=1, keepdims=True)                       # extract_query_feats.py:198
        sim = np.matmul(feat, feat.T) - np.eye(len(feat))
        keep = greedy_select(sim)
        kept[data["queries"][v][0]] = keep
        x = drop(pca.transform(f[keep]).astype(np.float32))                       # :210, then score_normalization.py:121-131
        sims, _ = knn_oracle.knn_ip(np.ascontiguousarray(x, np.float32), np.ascontiguousarray(norm_bank, np.float32), NK)
        bias = -BETA * sims[:, :NK].mean(axis=1, keepdims=True)                   # :141-146
        q_feats.append(np.concatenate([x, bias], axis=1).astype(np.float32))
        q_owner.extend([v] * len(keep))
    Q = np.ascontiguousarray(np.concatenate(q_feats), np.float32)
    R = np.ascontiguousarray(refs_sn, np.float32)
    S = knn_oracle.ip_matrix(Q, R)                                                # Flat inner-product index
    global_k = int(RETRIEVE_PER_QUERY * nq)                                       # sscd_baseline.py:97
    flat = S.ravel()
    order = np.argsort(-flat, kind="stable")[:global_k]                           # index.py:150-163: the global_k best frame pairs
    qi, ri = np.unravel_index(order, S.shape)
    best = {}
    for a, b, s in zip(qi, ri, flat[order]):                                      # candidates.py:24-26: max per video pair
        key = (q_owner[a], b // F)
        if key not in best or s > best[key]:
            best[key] = s
    pairs = sorted(best.items(), key=lambda kv: kv[1], reverse=True)              # candidates.py:38
    pairs = pairs[: int(CANDIDATES_PER_QUERY * nq)]                               # sscd_baseline.py:99-100
    cands = [(data["queries"][q][0], data["refs"][r][0], float(s)) for (q, r), s in pairs]
    return cands, pca, kept, low_var_dim


def reference_uap(cands, gt):
    spec = importlib.util.spec_from_file_location("ref_vsc_metrics", "/root/reference/VSC22-Descriptor-Track-1st/infer/vsc/metrics.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    P = [ref.CandidatePair(q, r, s) for q, r, s in cands]
    G = [ref.CandidatePair(q, r, 1.0) for q, r in gt]
    ap = ref.average_precision(G, P)
    return float(ap.ap), float(ap.simple_ap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cache", default="")
    ap.add_argument("--dry", action="store_true", help="print the numbers, do not write the fixture")
    ap.add_argument("--large", action="store_true", help="the second fixture (uap_e2e_large.npz): 160 references, 192 queries of which 150 hold copies "
                    "-- 3.4 x the positives, so one adjacent swap moves the uAP by a third as much; per-frame descriptors are not stored")
    args = ap.parse_args()
    torch.set_num_threads(8)
    sizes = dict(n_ref=160, n_norm=60, n_query=192, n_positive=150) if args.large else {}
    data = synth_videos.make(SEED + (1 if args.large else 0), **sizes)
    desc = descriptors(data, args.cache)
    cands, pca, kept, low_var_dim = chain(data, desc)
    uap, simple = reference_uap(cands, data["gt"])
    gtset = set(data["gt"])
    ranks = [i for i, (q, r, _) in enumerate(cands) if (q, r) in gtset]
    print(f"{len(cands)} candidates, {len(ranks)} of {len(gtset)} ground-truth pairs among them; uAP {uap:.6f} (simple {simple:.6f})")
    print("ranks of the ground-truth pairs:", ranks)
    print("low-variance dimension", low_var_dim, "; videos that lost a frame to the duplicate filter:",
          sum(len(k) < synth_videos.FRAMES for k in kept.values()))
    if args.dry:
        return
    rows = np.arange(0, len(desc[0]), 67)          # --large: a sample of the per-frame descriptors (every 67th frame), for check_uape2e
    extra = {"sample_rows": rows, "desc_swin_sample": desc[0][rows].astype(np.float32), "desc_vit_sample": desc[1][rows].astype(np.float32)} if args.large \
        else {"desc_swin": desc[0].astype(np.float32), "desc_vit": desc[1].astype(np.float32)}
    out_path = OUT.replace(".npz", "_large.npz") if args.large else OUT
    np.savez_compressed(
        out_path, seed=SEED + (1 if args.large else 0), sizes=np.array([len(data["refs"]), len(data["norm"]), len(data["queries"]), len(data["gt"])]),
        fingerprint=data["fingerprint"], swin_preset=SWIN_PRESET, swin_weights_seed=SWIN_WEIGHTS,
        vit_preset=VIT_PRESET, vit_weights_seed=VIT_WEIGHTS, pca_mean=pca.mean_.astype(np.float32),
        pca_components=pca.components_.astype(np.float32), pca_explained_variance=pca.explained_variance_.astype(np.float32),
        cand_query=np.array([c[0] for c in cands]), cand_ref=np.array([c[1] for c in cands]),
        cand_score=np.array([c[2] for c in cands], np.float32), uap=uap, simple_ap=simple, low_var_dim=low_var_dim,
        kept_counts=np.array([len(kept[q]) for q, _ in data["queries"]]), **extra)
    print(f"{out_path}: {os.path.getsize(out_path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
