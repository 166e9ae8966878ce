"""CandidateGeneration.query at a synthetic scale of the descriptor track's eval step (sscd_baseline.search: global_k = 1200 per query
video): [n_q_videos] x 20 frames against [n_r_videos] x 25 frames of 512-d descriptors.   (run on the GPU box)
    python tools/micro/candidates_bench.py [n_q_videos] [n_r_videos]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tools import synth
from vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc.index import VideoFeature
nqv = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nrv = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(0)


def vids(prefix, n, frames, seed):
    x = rng.standard_normal((n * frames, 512), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return [VideoFeature(video_id=f"{prefix}{i:06d}", feature=x[i * frames:(i + 1) * frames], timestamps=np.arange(frames, dtype=np.float32)) for i in range(n)]


refs, queries = vids("R", nrv, 25, 1), vids("Q", nqv, 20, 2)
for i in range(0, nqv, 10):      # every tenth query video holds copies of reference frames: true matches far above the noise
    queries[i].feature[:5] = refs[(7 * i) % nrv].feature[:5]
t0 = time.perf_counter()
cg = CandidateGeneration(refs, MaxScoreAggregation())
t1 = time.perf_counter()
cands = cg.query(queries, global_k=1200 * nqv, limit=25 * nqv)      # what sscd_baseline.search asks for
t2 = time.perf_counter()
print(f"{nqv} x 20 query frames, {nrv} x 25 reference frames, global_k {1200 * nqv}: index {t1 - t0:.2f} s, query {t2 - t1:.2f} s -> {len(cands)} candidate pairs, best {cands[0].score:.4f}")
