"""Thin torch-tensor wrappers over the C ABI building blocks (used by the parity
tests and by vsc/index.py).  torch here is device memory + streams only."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, current_stream, ptr


_PRECISION = "bf16"


class operands:
    """with ops.operands("fp16"): ...  -- the kernel-level wrappers below then call libvsc_hip_f16.so and their 16-bit tensors are
    torch.float16 (vsc_operand_dtype, include/vsc_hip.h).  Default: the bf16 library."""

    def __init__(self, precision: str):
        assert precision in _lib.LIB_PATHS, precision
        self.precision, self._saved = precision, None

    def __enter__(self):
        global _PRECISION
        self._saved, _PRECISION = _PRECISION, self.precision
        return self

    def __exit__(self, *exc):
        global _PRECISION
        _PRECISION = self._saved
        return False


def lp_dtype():
    """torch dtype of the current library's 16-bit operands"""
    return torch.float16 if _PRECISION == "fp16" else torch.bfloat16


def _rd():
    return _lib.require_device(_PRECISION)


def _dev(t: torch.Tensor, dtype) -> torch.Tensor:
    assert t.is_cuda, "operand must live on the GPU"
    return t.to(dtype).contiguous()


def gemm_bf16(a, w, bias=None, *, epilogue=_lib.EPI_BF16, aux=None, tokens=0, out=None):
    """out = epi(a @ w.T + bias); a [M,K] bf16, w [N,K] bf16."""
    lib = _rd()
    a, w = _dev(a, lp_dtype()), _dev(w, lp_dtype())
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    bias = None if bias is None else _dev(bias, torch.float32)
    aux = None if aux is None else _dev(aux, torch.float32)
    if out is None:
        if epilogue in (_lib.EPI_BF16, _lib.EPI_GELU_BF16, _lib.EPI_QGELU_BF16):
            out = torch.empty((m, n), dtype=lp_dtype(), device=a.device)
        elif epilogue in (_lib.EPI_RESADD_F32, _lib.EPI_F32):
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        else:
            frames = m // (tokens - 1)
            out = torch.zeros((frames * tokens, n), dtype=torch.float32, device=a.device)
    check(lib.vsc_gemm_bf16(ptr(a), ptr(w), ptr(bias), ptr(aux), ptr(out), m, n, k, epilogue, tokens,
                            current_stream()))
    return out


def attention_bf16(qkv, frames: int, tokens: int, heads: int):
    lib = _rd()
    qkv = _dev(qkv, lp_dtype())
    assert qkv.shape == (frames * tokens, 3 * heads * 64)
    out = torch.empty((frames * tokens, heads * 64), dtype=lp_dtype(), device=qkv.device)
    check(lib.vsc_attention_bf16(ptr(qkv), ptr(out), frames, tokens, heads, current_stream()))
    return out


def layernorm(x, gamma, beta, eps: float, out_f32: bool = False):
    lib = _rd()
    x, gamma, beta = (_dev(t, torch.float32) for t in (x, gamma, beta))
    rows, width = x.shape
    out = torch.empty((rows, width), dtype=torch.float32 if out_f32 else lp_dtype(), device=x.device)
    check(lib.vsc_layernorm_f32(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, width, eps,
                                int(out_f32), current_stream()))
    return out


def patchify_bf16(frames, patch: int, kpad: int):
    lib = _rd()
    frames = _dev(frames, torch.float32)
    n, c, h, w = frames.shape
    assert h == w
    g = h // patch
    out = torch.empty((n * g * g, kpad), dtype=lp_dtype(), device=frames.device)
    check(lib.vsc_patchify_bf16(ptr(frames), ptr(out), n, c, h, patch, kpad, current_stream()))
    return out


def l2_normalize_(x):
    """In place, sklearn.preprocessing.normalize semantics."""
    lib = _rd()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    if x.shape[0]:
        check(lib.vsc_l2_normalize_f32(ptr(x), x.shape[0], x.shape[1], current_stream()))
    return x


def knn_ip(q, r, k: int, ref_id_offset: int = 0, floor=None):
    """Exact inner-product top-k.  q [nq,d], r [nr,d] float32 on the GPU ->
    (scores [nq,k] float32 descending, ids [nq,k] int64).  Empty inputs follow
    faiss: nq == 0 -> empty outputs; nr == 0 -> all (-FLT_MAX, -1).
    floor [nq] float32: only references with <q, r> >= floor[q] (vsc_knn_ip_floor_f32; unused slots (-FLT_MAX, -1))."""
    lib = _rd()
    q, r = _dev(q, torch.float32), _dev(r, torch.float32)
    nq, d = q.shape
    nr = r.shape[0]
    assert r.shape[1] == d, "query / reference dimension mismatch"
    scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    if nq == 0:
        return scores, ids
    if nr == 0:
        scores.fill_(torch.finfo(torch.float32).min)
        ids.fill_(-1)
        return scores, ids
    if floor is not None:
        floor = _dev(floor, torch.float32)
        assert floor.shape == (nq,)
        check(lib.vsc_knn_ip_floor_f32(ptr(q), nq, ptr(r), nr, d, k, ref_id_offset, ptr(floor), ptr(scores), ptr(ids), current_stream()))
        return scores, ids
    check(lib.vsc_knn_ip_f32(ptr(q), nq, ptr(r), nr, d, k, ref_id_offset, ptr(scores), ptr(ids),
                             current_stream()))
    return scores, ids


def knn_merge_parts(scores, ids):
    """scores / ids [parts, nq, k]: per-shard results of knn_ip (each with its ref_id_offset) -> the k best of the union, in the
    search's order (score descending, equal scores by ascending id)."""
    lib = _rd()
    scores, ids = _dev(scores, torch.float32), _dev(ids, torch.int64)
    parts, nq, k = scores.shape
    assert ids.shape == scores.shape and 1 <= parts <= 64
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    check(lib.vsc_knn_merge_parts_f32(ptr(scores), ptr(ids), parts, nq, k, ptr(out_s), ptr(out_i), current_stream()))
    return out_s, out_i


def range_search_ip(q, r, radius: float, ref_id_offset: int = 0, capacity: int = 1 << 20):
    """All pairs with <q, r> > radius.  -> (lims [nq+1] int64, scores, ids), hits of query i in
    lims[i]:lims[i+1], ascending reference id (faiss range_search layout)."""
    import ctypes
    lib = _rd()
    q, r = _dev(q, torch.float32), _dev(r, torch.float32)
    nq, d = q.shape
    nr = r.shape[0]
    assert r.shape[1] == d, "query / reference dimension mismatch"
    lims = torch.zeros(nq + 1, dtype=torch.int64, device=q.device)
    empty = (lims, torch.empty(0, dtype=torch.float32, device=q.device),
             torch.empty(0, dtype=torch.int64, device=q.device))
    if nq == 0 or nr == 0:
        return empty
    total = ctypes.c_int64(0)
    while True:
        scores = torch.empty(capacity, dtype=torch.float32, device=q.device)
        ids = torch.empty(capacity, dtype=torch.int64, device=q.device)
        check(lib.vsc_range_search_ip_f32(ptr(q), nq, ptr(r), nr, d, float(radius), ref_id_offset, ptr(lims),
                                          ptr(scores), ptr(ids), capacity, ctypes.byref(total),
                                          current_stream()))
        if total.value <= capacity:
            return lims, scores[: total.value], ids[: total.value]
        capacity = int(total.value)


def range_count_ip(q, r, radius: float) -> int:
    """Number of pairs with <q, r> > radius: the counting pass of vsc_range_search_ip_f32 alone (capacity 0)."""
    import ctypes
    lib = _rd()
    q, r = _dev(q, torch.float32), _dev(r, torch.float32)
    nq, d = q.shape
    nr = r.shape[0]
    assert r.shape[1] == d, "query / reference dimension mismatch"
    if nq == 0 or nr == 0:
        return 0
    lims = torch.zeros(nq + 1, dtype=torch.int64, device=q.device)
    total = ctypes.c_int64(0)
    check(lib.vsc_range_search_ip_f32(ptr(q), nq, ptr(r), nr, d, float(radius), 0, ptr(lims), None, None, 0,
                                      ctypes.byref(total), current_stream()))
    return int(total.value)


def pair_similarity(q, r, pairs):
    """Frame x frame similarity matrices of candidate (query video, reference video) pairs.
    q [nq, d], r [nr, d]: frame banks; pairs: int64 [n, 4] rows (q_row0, q_rows, r_row0, r_rows) on the host.
    -> (flat f32 device tensor, offsets int64 numpy [n + 1]); matrix p = flat[off[p]:off[p+1]].view(q_rows, r_rows)."""
    import numpy as np
    lib = _rd()
    q, r = _dev(q, torch.float32), _dev(r, torch.float32)
    assert q.dim() == 2 and r.dim() == 2 and q.shape[1] == r.shape[1], "query / reference dimension mismatch"
    pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int64).reshape(-1, 4))
    n = pairs.shape[0]
    offsets = np.zeros(n + 1, dtype=np.int64)
    total = int((pairs[:, 1] * pairs[:, 3]).sum())
    out = torch.empty(total, dtype=torch.float32, device=q.device)
    if q.shape[0] == 0 or r.shape[0] == 0:
        if total:
            raise ValueError("pairs reference rows of an empty bank")
        return out, offsets
    check(lib.vsc_pair_similarity_f32(ptr(q), q.shape[0], ptr(r), r.shape[0], q.shape[1], pairs.ctypes.data, n,
                                      offsets.ctypes.data, ptr(out) if total else None, total, current_stream()))
    return out, offsets


def video_pair_max(q, q_video, n_q_videos: int, r, r_video, n_r_videos: int, threshold: float, capacity: int = 1 << 20):
    """Largest frame score above ``threshold`` per (query video, reference video).
    q [nq, d], r [nr, d] float32; q_video [nq], r_video [nr] int32 video index of every row.
    -> (lims [n_q_videos + 1] int64, ref_video int32, score float32): pairs of query video v in
    lims[v]:lims[v+1], ascending reference video."""
    import ctypes
    lib = _rd()
    q, r = _dev(q, torch.float32), _dev(r, torch.float32)
    assert q.dim() == 2 and r.dim() == 2 and q.shape[1] == r.shape[1], "query / reference dimension mismatch"
    q_video, r_video = _dev(q_video, torch.int32), _dev(r_video, torch.int32)
    assert q_video.shape == (q.shape[0],) and r_video.shape == (r.shape[0],)
    # the sweep indexes a dense [n_q_videos, n_r_videos] table with these ids (atomic max): reject ids outside it here
    if q.shape[0] and (int(q_video.min()) < 0 or int(q_video.max()) >= n_q_videos):
        raise ValueError(f"q_video ids outside [0, {n_q_videos})")
    if r.shape[0] and (int(r_video.min()) < 0 or int(r_video.max()) >= n_r_videos):
        raise ValueError(f"r_video ids outside [0, {n_r_videos})")
    lims = torch.zeros(n_q_videos + 1, dtype=torch.int64, device=q.device)
    if q.shape[0] == 0 or r.shape[0] == 0 or n_q_videos == 0 or n_r_videos == 0:
        return (lims, torch.empty(0, dtype=torch.int32, device=q.device),
                torch.empty(0, dtype=torch.float32, device=q.device))
    total = ctypes.c_int64(0)
    while True:
        rv = torch.empty(capacity, dtype=torch.int32, device=q.device)
        sc = torch.empty(capacity, dtype=torch.float32, device=q.device)
        check(lib.vsc_video_pair_max_f32(ptr(q), q.shape[0], ptr(q_video), n_q_videos, ptr(r), r.shape[0], ptr(r_video),
                                         n_r_videos, q.shape[1], float(threshold), ptr(lims), ptr(rv), ptr(sc), capacity,
                                         ctypes.byref(total), current_stream()))
        if total.value <= capacity:
            return lims, rv[: total.value], sc[: total.value]
        capacity = int(total.value)


def window_attention_bf16(qkv, bias, scale, frames: int, res: int, window: int, shift: int, heads: int, bounded: bool = False):
    """Swin-V2 windowed cosine attention (head_dim 32) on image-ordered tokens.  bounded: fold every head's logit upper bound
    scale + max(bias) into its table and pass -scale where the head's logits span <= 69 (vsc_hip.h: the kernel then skips the
    softmax's row maximum), as vsc_swin_finalize does for its own tables."""
    lib = _rd()
    qkv = _dev(qkv, lp_dtype())
    bias, scale = _dev(bias, torch.float32), _dev(scale, torch.float32)
    if bounded:
        bmax, bmin = bias.max(dim=1).values, bias.min(dim=1).values
        ok = (2 * scale + (bmax - bmin)) <= 69.0
        bias = torch.where(ok[:, None], bias - (bmax + scale)[:, None], bias).contiguous()
        scale = torch.where(ok, -scale, scale).contiguous()
    assert qkv.shape == (frames * res * res, 3 * heads * 32)
    assert bias.shape == (heads, (2 * window - 1) ** 2) and scale.shape == (heads,)   # compact table
    out = torch.empty((frames * res * res, heads * 32), dtype=lp_dtype(), device=qkv.device)
    check(lib.vsc_window_attention_bf16(ptr(qkv), ptr(out), ptr(bias), ptr(scale), frames, res, window, shift, heads,
                                        current_stream()))
    return out


def ln_residual(t, gamma, beta, eps: float, x_in=None):
    """-> (x fp32, xb bf16) with x = (x_in or 0) + LayerNorm(t)."""
    lib = _rd()
    t, gamma, beta = (_dev(a, torch.float32) for a in (t, gamma, beta))
    x_in = None if x_in is None else _dev(x_in, torch.float32)
    rows, width = t.shape
    x = torch.empty_like(t)
    xb = torch.empty((rows, width), dtype=lp_dtype(), device=t.device)
    check(lib.vsc_ln_residual_f32(ptr(t), ptr(gamma), ptr(beta), ptr(x_in), ptr(x), ptr(xb), rows, width, eps,
                                  current_stream()))
    return x, xb


def gemm_ln_bf16(a, w, bias, gamma, beta, eps: float, x_in=None):
    """-> (x fp32, xb bf16) with x = (x_in or 0) + LayerNorm(a @ w.T + bias); w is [n, k], n in {128, 256, 512}."""
    lib = _rd()
    a, w = _dev(a, lp_dtype()), _dev(w, lp_dtype())
    gamma, beta = _dev(gamma, torch.float32), _dev(beta, torch.float32)
    bias = None if bias is None else _dev(bias, torch.float32)
    x_in = None if x_in is None else _dev(x_in, torch.float32)
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    x = torch.empty((m, n), dtype=torch.float32, device=a.device)
    xb = torch.empty((m, n), dtype=lp_dtype(), device=a.device)
    check(lib.vsc_gemm_ln_bf16(ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(beta), ptr(x_in), ptr(x), ptr(xb), m, n, k,
                               eps, current_stream()))
    return x, xb


def swin_mlp_bf16(x, w1, b1, w2, b2, gamma, beta, eps: float):
    """Fused Swin-V2 MLP, widths 128 / 256 / 512: -> (x + LayerNorm(gelu(bf16(x) @ w1.T + b1) @ w2.T + b2), its bf16 shadow).
    w1 [4c, c], w2 [c, 4c] as the module holds them (the hidden-axis reordering the kernel wants is done here)."""
    import numpy as np
    lib = _rd()
    x = _dev(x, torch.float32).clone()
    m, c = x.shape
    xb = x.to(lp_dtype())
    w2h = np.ascontiguousarray(w2.detach().float().cpu().numpy())
    assert w2h.shape == (c, 4 * c) and tuple(w1.shape) == (4 * c, c)
    w2p = np.empty_like(w2h)
    check(lib.vsc_swin_mlp_permute_hidden_f32(w2h.ctypes.data, w2p.ctypes.data, c))
    w1d = _dev(w1.to(x.device), lp_dtype())
    w2d = torch.from_numpy(w2p).to(x.device).to(lp_dtype())
    b1, b2 = _dev(b1.to(x.device), torch.float32), _dev(b2.to(x.device), torch.float32)
    gamma, beta = _dev(gamma.to(x.device), torch.float32), _dev(beta.to(x.device), torch.float32)
    check(lib.vsc_swin_mlp_bf16(ptr(w1d), ptr(b1), ptr(w2d), ptr(b2), ptr(gamma), ptr(beta), ptr(x), ptr(xb), m, c, eps,
                                current_stream()))
    return x, xb


def swin_proj_mlp_bf16(x, att, wp, bp, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, eps: float):
    """The second half of a Swin-V2 block in one launch, widths 128 / 256 / 512:
    x1 = x + LN(att @ wp.T + bp) * gamma1 + beta1;  -> (x1 + LN(gelu(bf16(x1) @ w1.T + b1) @ w2.T + b2) * gamma2 + beta2, its bf16 shadow).
    att [m, c] (rounded to bf16 here); weights as the module holds them."""
    import numpy as np
    lib = _rd()
    x = _dev(x, torch.float32).clone()
    m, c = x.shape
    dev = x.device
    xb = torch.empty((m, c), dtype=lp_dtype(), device=dev)
    attd = _dev(att.to(dev), lp_dtype())
    w2h = np.ascontiguousarray(w2.detach().float().cpu().numpy())
    w2p = np.empty_like(w2h)
    check(lib.vsc_swin_mlp_permute_hidden_f32(w2h.ctypes.data, w2p.ctypes.data, c))
    wpd, w1d = _dev(wp.to(dev), lp_dtype()), _dev(w1.to(dev), lp_dtype())
    w2d = torch.from_numpy(w2p).to(dev).to(lp_dtype())
    f = [_dev(t.to(dev), torch.float32) for t in (bp, gamma1, beta1, b1, b2, gamma2, beta2)]
    check(lib.vsc_swin_proj_mlp_bf16(ptr(attd), ptr(wpd), ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(w1d), ptr(f[3]), ptr(w2d), ptr(f[4]), ptr(f[5]),
                                     ptr(f[6]), ptr(x), ptr(xb), m, c, eps, current_stream()))
    return x, xb


def swin_proj_mlp_qkv_bf16(x, att, wp, bp, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, wq, bq, eps: float):
    """swin_proj_mlp_bf16 with the next block's qkv Linear behind it (width 512): -> (x_out fp32, qkv_next = bf16(x_out) @ wq.T + bq as bf16)."""
    import numpy as np
    lib = _rd()
    x = _dev(x, torch.float32).clone()
    m, c = x.shape
    dev = x.device
    qkv = torch.empty((m, 3 * c), dtype=lp_dtype(), device=dev)
    attd = _dev(att.to(dev), lp_dtype())
    w2h = np.ascontiguousarray(w2.detach().float().cpu().numpy())
    w2p = np.empty_like(w2h)
    check(lib.vsc_swin_mlp_permute_hidden_f32(w2h.ctypes.data, w2p.ctypes.data, c))
    wpd, w1d, wqd = (_dev(t.to(dev), lp_dtype()) for t in (wp, w1, wq))
    w2d = torch.from_numpy(w2p).to(dev).to(lp_dtype())
    f = [_dev(t.to(dev), torch.float32) for t in (bp, gamma1, beta1, b1, b2, gamma2, beta2, bq)]
    check(lib.vsc_swin_proj_mlp_qkv_bf16(ptr(attd), ptr(wpd), ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(w1d), ptr(f[3]), ptr(w2d), ptr(f[4]), ptr(f[5]),
                                         ptr(f[6]), ptr(wqd), ptr(f[7]), ptr(x), ptr(qkv), m, c, eps, current_stream()))
    return x, qkv


def merge_gather_bf16(xb, frames: int, res: int):
    lib = _rd()
    xb = _dev(xb, lp_dtype())
    c = xb.shape[1]
    assert xb.shape[0] == frames * res * res
    out = torch.empty((frames * (res // 2) ** 2, 4 * c), dtype=lp_dtype(), device=xb.device)
    check(lib.vsc_merge_gather_bf16(ptr(xb), ptr(out), frames, res, c, current_stream()))
    return out
