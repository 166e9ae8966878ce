"""Generates tests/golden/uap_reference.json: inputs (pairs with TIED scores, ground truth with pairs nobody
predicted) and the outputs of the REFERENCE's own vsc.metrics.average_precision
(/root/reference/VSC22-Descriptor-Track-1st/infer/vsc/metrics.py:423-494: `.ap` = drivendata_average_precision over sklearn's
tie-grouped average_precision_score, `.simple_ap`).  Build container only (imports the reference file); the json travels.

    python tests/golden/gen_uap_golden.py
"""
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference/VSC22-Descriptor-Track-1st/infer/vsc/metrics.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "uap_reference.json")


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_vsc_metrics", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_case(rng, n_pairs, n_gt_hit, n_gt_missed, decimals):
    q = rng.integers(0, 60, 4 * n_pairs)
    r = rng.integers(0, 90, 4 * n_pairs)
    keys = sorted({(int(a), int(b)) for a, b in zip(q, r)})[:n_pairs]
    rng.shuffle(keys)
    keys = [tuple(map(int, k)) for k in keys]
    scores = rng.random(len(keys))
    if decimals is not None:
        scores = np.round(scores, decimals)          # duplicated frames → tied max-aggregated scores
    hit = [keys[i] for i in rng.choice(len(keys), n_gt_hit, replace=False)]
    missed = [(1000 + i, 7) for i in range(n_gt_missed)]
    return {"pred": [[a, b, float(s)] for (a, b), s in zip(keys, scores)],
            "gt": [[int(a), int(b)] for a, b in hit + missed], "decimals": decimals}


def main():
    ref = load_reference()
    rng = np.random.default_rng(20220930)
    cases = [make_case(rng, n, h, m, d) for n, h, m, d in
             [(400, 60, 9, None), (400, 60, 9, 2), (400, 60, 9, 1), (400, 80, 0, 0), (50, 50, 0, 1), (30, 1, 5, 1)]]
    for c in cases:
        P = [ref.CandidatePair(f"Q{a:06d}", f"R{b:06d}", s) for a, b, s in c["pred"]]
        G = [ref.CandidatePair(f"Q{a:06d}", f"R{b:06d}", 1.0) for a, b in c["gt"]]
        ap = ref.average_precision(G, P)
        c["ap"] = float(ap.ap)
        c["simple_ap"] = float(ap.simple_ap)
        c["curve_recalls_sum"] = float(np.sum(ap.pr_curve.recalls))
        c["curve_precisions_sum"] = float(np.sum(ap.pr_curve.precisions))
        print(c["decimals"], len(P), c["ap"], c["simple_ap"])
    with open(OUT, "w") as f:
        json.dump(cases, f)


if __name__ == "__main__":
    main()
