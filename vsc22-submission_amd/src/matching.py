"""Matching-track candidate features: per-pair frame x frame similarity matrices on the HIP path.

Mirrors VSC22-Matching-Track-1st/infer/src/utils.py (same function names -- including the reference's
spellings -- arguments and return structure) for the part of `infer_matching.py` that sits between the
candidate search and the classifier / refinement networks:

  * ``generate_candidates_classfiy_feature``  utils.py:21-51   similarity maps fed to the match classifier
  * ``generate_matching_feature``             utils.py:54-77   (query view, reference) feature pairs for refinement
  * ``calclualte_low_var_dim``                utils.py:7-10
  * ``transform_features``                    utils.py:12-16
  * ``search_candidate_pairs``                infer_matching.py:229-262   candidate (query video, reference video) search
  * ``MatchRefineDataset``                    src/dataset.py:127-144      padded maps fed to the refinement networks
  * ``match_classify`` / ``match_refine``     infer_matching.py:158-204   the MobileNetV3 pair classifier and the HRNet
                                              refinement net (vsc_hip/cnn.py: fp32 convolutions on the HIP path)

The reference runs one ``np.matmul(qfeat, rfeat.T)`` per candidate on the host (twice: once to pick the query
view, once for the map).  Here every candidate of a call goes through ONE ``vsc_pair_similarity_f32`` launch
over concatenated frame banks; each score is an independent ascending-k fp32 chain, so the map of the chosen
view is a row slice of the full product -- bit for bit -- and nothing is multiplied twice.

``generate_matching_result`` (utils.py:80-116) -- CPU post-processing of the refinement networks' probability
maps, cv2 connected components + sklearn RANSAC in the reference -- is mirrored as host code with
``scipy.ndimage.label`` in cv2's place (cv2 is not a dependency here); it is not part of the GPU path.
"""
import dataclasses
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

TOP_ROWS = 10  # rows whose maxima are averaged to score a query view (utils.py:41)
SEARCH_THRESHOLD = -0.1  # infer_matching.py:62


def calclualte_low_var_dim(score_norm_refs) -> int:
    """Index of the descriptor dimension with the smallest variance over all reference frames."""
    sn_features = np.concatenate([ref.feature for ref in score_norm_refs], axis=0)
    return int(sn_features.var(axis=0).argmin())


def transform_features(features, transform: Callable[[np.ndarray], np.ndarray]):
    return [dataclasses.replace(feature, feature=transform(feature.feature)) for feature in features]


def _hip_video_pair_max(q_bank, q_video, n_q_videos, r_bank, r_video, n_r_videos, threshold):
    import torch

    from vsc_hip import _lib, ops
    _lib.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    lims, rv, sc = ops.video_pair_max(torch.from_numpy(q_bank).to(dev), torch.from_numpy(q_video).to(dev), n_q_videos,
                                      torch.from_numpy(r_bank).to(dev), torch.from_numpy(r_video).to(dev), n_r_videos,
                                      threshold)
    return lims.cpu().numpy(), rv.cpu().numpy(), sc.cpu().numpy()


def _bank_with_video_index(videos):
    feats = [np.ascontiguousarray(v.feature, dtype=np.float32) for v in videos]
    dim = feats[0].shape[1] if feats else 0
    bank = np.concatenate(feats, axis=0) if feats else np.zeros((0, dim), np.float32)
    index = np.repeat(np.arange(len(feats), dtype=np.int32), [len(f) for f in feats])
    return bank, index


def search_candidate_pairs(sn_query_list, sn_refs, threshold: float = SEARCH_THRESHOLD,
                           video_pair_max: Optional[Callable] = None) -> List[Tuple[str, str, np.float32]]:
    """Candidate (query_id, ref_id, score) triples, best first (infer_matching.py:229-262).

    The reference searches the flat inner-product index per query video for the top 1024 references of every
    frame, re-runs ``range_search`` for the frames whose 1024th score still exceeds the threshold, and keeps the
    best frame score of every (query video, reference video) in a dict.  Both branches together select exactly
    the frame pairs scoring above ``threshold``, so one ``vsc_video_pair_max_f32`` sweep over all query frames
    and all reference frames yields the dict's contents; what remains here is the sort.  Order: descending
    score; pairs with bit-equal scores follow (query video, reference video) order where the reference keeps
    their first-encounter order -- the only difference, and one that needs an exact fp32 tie across pairs.
    ``video_pair_max(q_bank, q_video, n_q, r_bank, r_video, n_r, threshold) -> (lims, ref_video, score)`` is a
    test seam; the default is the HIP path."""
    if len(sn_query_list) == 0 or len(sn_refs) == 0:
        return []
    q_bank, q_video = _bank_with_video_index(sn_query_list)
    r_bank, r_video = _bank_with_video_index(sn_refs)
    if len(q_bank) == 0 or len(r_bank) == 0:
        return []
    lims, ref_video, score = (video_pair_max or _hip_video_pair_max)(
        q_bank, q_video, len(sn_query_list), r_bank, r_video, len(sn_refs), float(threshold))
    query_video = np.repeat(np.arange(len(sn_query_list)), np.diff(lims))
    order = np.lexsort((ref_video, query_video, -score.astype(np.float64)))
    return [(sn_query_list[query_video[i]].video_id, sn_refs[ref_video[i]].video_id, score[i]) for i in order]


def _hip_pair_similarity(q_bank: np.ndarray, r_bank: np.ndarray, pairs: np.ndarray):
    import torch

    from vsc_hip import _lib, ops
    _lib.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    flat, offsets = ops.pair_similarity(torch.from_numpy(q_bank).to(dev), torch.from_numpy(r_bank).to(dev), pairs)
    return flat.cpu().numpy(), offsets


class _Banks:
    """Concatenated frame banks of the videos a candidate list touches + the pair table of the launch."""

    def __init__(self, query: Dict[str, np.ndarray], ref: Dict[str, np.ndarray], candidates: Sequence[Tuple]):
        self.q_rows: Dict[str, Tuple[int, int]] = {}
        self.r_rows: Dict[str, Tuple[int, int]] = {}
        q_parts, r_parts = [], []
        nq = nr = 0
        for qid, rid, _ in candidates:
            if qid not in self.q_rows:
                f = np.ascontiguousarray(query[qid], dtype=np.float32)
                self.q_rows[qid] = (nq, len(f))
                q_parts.append(f)
                nq += len(f)
            if rid not in self.r_rows:
                f = np.ascontiguousarray(ref[rid], dtype=np.float32)
                self.r_rows[rid] = (nr, len(f))
                r_parts.append(f)
                nr += len(f)
        dim = q_parts[0].shape[1] if q_parts else 0
        self.q_bank = np.concatenate(q_parts, axis=0) if q_parts else np.zeros((0, dim), np.float32)
        self.r_bank = np.concatenate(r_parts, axis=0) if r_parts else np.zeros((0, dim), np.float32)
        self.pairs = np.array([[*self.q_rows[qid], *self.r_rows[rid]] for qid, rid, _ in candidates],
                              dtype=np.int64).reshape(-1, 4)


def pair_similarity_matrices(query: Dict[str, np.ndarray], ref: Dict[str, np.ndarray], candidates: Sequence[Tuple],
                             pair_similarity: Optional[Callable] = None) -> List[np.ndarray]:
    """[q_frames, r_frames] similarity matrix of every (qid, rid, score) candidate, one launch for all of them.
    ``pair_similarity(q_bank, r_bank, pairs) -> (flat, offsets)`` is a test seam; the default is the HIP path."""
    if len(candidates) == 0:
        return []
    banks = _Banks(query, ref, candidates)
    flat, off = (pair_similarity or _hip_pair_similarity)(banks.q_bank, banks.r_bank, banks.pairs)
    return [flat[off[i]:off[i + 1]].reshape(int(banks.pairs[i, 1]), int(banks.pairs[i, 3]))
            for i in range(len(candidates))]


def _best_view(sim_mat: np.ndarray, num_data: int) -> int:
    """Start row of the query view (block of ``num_data`` rows) whose ten best row maxima have the largest mean
    (utils.py:35-44); the first such view on ties, as np.argmax."""
    best, best_score = 0, None
    for start in range(0, sim_mat.shape[0], num_data):
        maxs = np.sort(sim_mat[start:start + num_data].max(1))
        score = maxs[-TOP_ROWS:].mean()
        if best_score is None or score > best_score:
            best, best_score = start, score
    return best


def generate_candidates_classfiy_feature(query, ref, candidate_list, query_video_len_map, pair_similarity=None):
    """-> (features, infos): per candidate the [q, r] map and, wrapped in a list as in the reference, its [r, q]
    transpose; infos repeats [qid, rid, score] for both."""
    mats = pair_similarity_matrices(query, ref, candidate_list, pair_similarity)
    features, infos = [], []
    for (qid, rid, score), sim_mat in zip(candidate_list, mats):
        num_data = query_video_len_map[qid]
        if num_data != sim_mat.shape[0]:
            start = _best_view(sim_mat, num_data)
            sim_mat = sim_mat[start:start + num_data]
        features.append(np.ascontiguousarray(sim_mat))
        infos.append([qid, rid, score])
        features.append([np.ascontiguousarray(sim_mat.T)])
        infos.append([qid, rid, score])
    return features, infos


def generate_matching_feature(query, ref, query_video_len_map, candidate_score_list, pair_similarity=None):
    """-> [[qid, rid, qfeat (the selected view's rows), rfeat], ...]"""
    mats = pair_similarity_matrices(query, ref, candidate_score_list, pair_similarity)
    res_list = []
    for (qid, rid, _), sim_mat in zip(candidate_score_list, mats):
        num_data = query_video_len_map[qid]
        qfeat = query[qid]
        if num_data != len(qfeat):
            start = _best_view(sim_mat, num_data)
            qfeat = qfeat[start:start + num_data]
        res_list.append([qid, rid, qfeat, ref[rid]])
    return res_list


class MatchClassifyDataset:
    """Similarity maps cropped / zero-padded to ``resolution`` and repeated on 3 channels (src/dataset.py:103-124)."""

    def __init__(self, features, infos, resolution=(160, 160)):
        self.features, self.infos, self.resolution = features, infos, resolution

    def __len__(self):
        return len(self.features)

    def __getitem__(self, item):
        feature = self.features[item]
        if isinstance(feature, list):
            feature = feature[0]
        h, w = min(feature.shape[0], self.resolution[0]), min(feature.shape[1], self.resolution[1])
        canvas = np.zeros(self.resolution, dtype=np.float32)
        canvas[:h, :w] = feature[:h, :w]
        return np.stack([canvas] * 3), self.infos[item][0], self.infos[item][1]


class MatchRefineDataset:
    """(query view, reference) similarity maps cropped / zero-padded to ``resolution``, repeated on 3 channels, with
    the valid height and width (src/dataset.py:127-144).  The reference multiplies ``qfeat @ rfeat.T`` per item on
    the host; here the maps of all items come from one ``vsc_pair_similarity_f32`` launch on first access."""

    def __init__(self, meta, resolution=(160, 160), pair_similarity: Optional[Callable] = None):
        self.meta, self.resolution, self._pair_similarity = meta, resolution, pair_similarity
        self._maps: Optional[List[np.ndarray]] = None

    def __len__(self):
        return len(self.meta)

    def _similarity_maps(self) -> List[np.ndarray]:
        if self._maps is None:
            query = {i: m[2] for i, m in enumerate(self.meta)}
            ref = {i: m[3] for i, m in enumerate(self.meta)}
            self._maps = pair_similarity_matrices(query, ref, [(i, i, 0.0) for i in range(len(self.meta))],
                                                  self._pair_similarity)
        return self._maps

    def __getitem__(self, item):
        qid, rid = self.meta[item][0], self.meta[item][1]
        sim_mat = self._similarity_maps()[item]
        h, w = min(sim_mat.shape[0], self.resolution[0]), min(sim_mat.shape[1], self.resolution[1])
        feat = np.zeros(self.resolution, dtype=np.float32)
        feat[:h, :w] = sim_mat[:h, :w]
        return np.stack([feat, feat, feat]), qid, rid, h, w


# ---- the two networks between candidate search and localisation (infer_matching.py:158-204) -------------------
MATCH_CLS_RESOLUTION, MATCH_REFINE_RESOLUTION = (160, 160), (224, 224)   # infer_matching.py:159,178
MATCH_CLS_BATCH, MATCH_REFINE_BATCH = 2048, 16                           # infer_matching.py:160,179


def load_match_models(cls_state_dicts, refine_state_dicts, device="cuda"):
    """State dicts of the reference's ClassifyModel / HRnet modules (train/models.py; timm parameter names; a
    torch.jit.load(...).state_dict() of the shipped submit_cls_model*.pt / submit_match_model*.pt has them) ->
    (classifier models, refinement models) on the HIP path."""
    from vsc_hip import cnn
    return ([cnn.MobileNetV3SmallHip(sd, device) for sd in cls_state_dicts],
            [cnn.HRNetRefineHip(sd, device) for sd in refine_state_dicts])


def match_classify(cls_models, match_cls_feature, match_cls_info, batch_size: int = MATCH_CLS_BATCH, device="cuda"):
    """Main.match_classify (infer_matching.py:158-175): probability of "this candidate pair holds a copy" per candidate,
    averaged over the classifier models.  -> list of (query_id, ref_id, prob) in candidate order (the reference builds a
    DataFrame with these three columns)."""
    import torch

    from vsc_hip import cnn
    data = MatchClassifyDataset(match_cls_feature, match_cls_info, MATCH_CLS_RESOLUTION)
    out = []
    for lo in range(0, len(data), batch_size):
        items = [data[i] for i in range(lo, min(lo + batch_size, len(data)))]
        feature = torch.from_numpy(np.stack([it[0] for it in items])).to(device)
        prob = cnn.match_classify_probability(cls_models, feature).cpu().numpy()
        out.extend((it[1], it[2], float(p)) for it, p in zip(items, prob))
    return out


def match_refine(refine_models, match_meta, batch_size: int = MATCH_REFINE_BATCH, device="cuda", pair_similarity=None):
    """Main.match_refine (infer_matching.py:177-204): per candidate the copy-probability map over (query frame, ref frame),
    averaged over the refinement models and over each model's transposed pass, cropped to the valid h x w.
    -> [[qid, rid, probability map [h, w], similarity map [h, w]], ...], the input of generate_matching_result."""
    import torch

    from vsc_hip import cnn
    data = MatchRefineDataset(match_meta, MATCH_REFINE_RESOLUTION, pair_similarity)
    res_list = []
    for lo in range(0, len(data), batch_size):
        items = [data[i] for i in range(lo, min(lo + batch_size, len(data)))]
        feature = torch.from_numpy(np.stack([it[0] for it in items]))
        pred = cnn.match_refine_probability(refine_models, feature.to(device)).cpu().numpy()
        for i, (fea, qid, rid, h, w) in enumerate(items):
            res_list.append([qid, rid, pred[i][1][:h, :w], fea[0][:h, :w]])
    return res_list


# ---- localisation from the refinement maps (host code, utils.py:80-116) ---------------------------------------
MIN_COMPONENT = 10      # a connected component needs more pixels than this to be fitted on its own
RANSAC_TRIALS, RANSAC_SEED, RANSAC_RESIDUAL = 200, 2023, 2


def _components8(mask: np.ndarray):
    """8-connected components of a boolean map -> (labels, count); 0 = background, components numbered in raster
    order of their first pixel (cv2 may number them differently; nothing downstream depends on the numbering)."""
    from scipy import ndimage
    labels, count = ndimage.label(mask, structure=np.ones((3, 3), dtype=np.int32))
    return labels.astype(np.int32), int(count)


def _fit_segment(prob_map: np.ndarray, x: np.ndarray, y: np.ndarray, std_ratio: float):
    """Weighted RANSAC line y = f(x) through the pixels (x = query frame, y = reference frame) -> the matched
    segment [q_start, r_start, q_end, r_end, score] or None."""
    from sklearn.linear_model import RANSACRegressor
    if len(set(x)) <= 3:
        return None
    ransac = RANSACRegressor(max_trials=RANSAC_TRIALS, random_state=RANSAC_SEED, residual_threshold=RANSAC_RESIDUAL)
    ransac.fit(x[:, np.newaxis], y[:, np.newaxis], sample_weight=np.square(prob_map[x, y]))
    on_line = abs(y - ransac.predict(x[:, np.newaxis]).flatten()) < 1
    slope = ransac.estimator_.coef_[0][0]
    if slope <= 0:
        return None
    slope = max(1 / slope, slope)
    xs, ys = x[on_line], y[on_line]
    if not (on_line.sum() > 5 and len(set(xs)) > 3 and len(set(ys)) > 3):
        return None
    top = prob_map[xs, ys]
    return [xs[0], ys[0], xs[-1], ys[-1], top.max() - top.std() * std_ratio - abs(slope - 1) / 10]


def generate_matching_result(res_list, threshold=0.05, std_ratio=2):
    """[[qid, rid, probability map, similarity map], ...] -> [[qid, rid, q_start, r_start, q_end, r_end, score], ...].

    Per map: pixels above ``threshold`` are split into 8-connected components; each component of more than ten
    pixels is fitted together with ALL pixels of the small components (or, when there is no large component, the
    small ones are fitted as one set); a fit yields a segment when its slope is positive and enough distinct
    frames lie within one frame of the line.  Same sklearn estimator, seed and thresholds as the reference."""
    match_res = []
    for qid, rid, prob_map, _ in res_list:
        loose = prob_map > threshold                 # ends up holding the pixels of the small components only
        labels, count = _components8(loose)
        large = []
        for i in range(1, count + 1):
            member = labels == i
            if member.sum() > MIN_COMPONENT:
                large.append(member)
                loose = loose & ~member
        groups = [member | loose for member in large] if large else [loose]
        for group in groups:
            x, y = np.where(group)
            seg = _fit_segment(prob_map, x, y, std_ratio)
            if seg is not None:
                match_res.append([qid, rid, *seg])
    return match_res
