"""ctypes face of oracle/knn_oracle.c (TEST INFRASTRUCTURE ONLY -- see the C header)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libknn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libknn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ip_matrix(q, r):
    q, r = _f32(q), _f32(r)
    out = np.empty((q.shape[0], r.shape[0]), dtype=np.float32)
    _load().oracle_ip_matrix(_p(q), ctypes.c_int64(q.shape[0]), _p(r), ctypes.c_int64(r.shape[0]),
                             ctypes.c_int(q.shape[1]), _p(out))
    return out


def knn_ip(q, r, k):
    """-> (D [nq,k] float32 descending, I [nq,k] int64); ties: lower index first."""
    q, r = _f32(q), _f32(r)
    assert q.shape[1] == r.shape[1]
    D = np.empty((q.shape[0], k), dtype=np.float32)
    I = np.empty((q.shape[0], k), dtype=np.int64)
    _load().oracle_knn_ip(_p(q), ctypes.c_int64(q.shape[0]), _p(r), ctypes.c_int64(r.shape[0]),
                          ctypes.c_int(q.shape[1]), ctypes.c_int(k), _p(D), _p(I))
    return D, I


def range_search_ip(q, r, radius):
    """-> (lims [nq+1] int64, D, I): all pairs with <q,r> > radius, ascending ref index."""
    q, r = _f32(q), _f32(r)
    nq = q.shape[0]
    counts = np.zeros(nq, dtype=np.int64)
    lib = _load()
    lib.oracle_range_count_ip(_p(q), ctypes.c_int64(nq), _p(r), ctypes.c_int64(r.shape[0]),
                              ctypes.c_int(q.shape[1]), ctypes.c_float(radius), _p(counts))
    lims = np.zeros(nq + 1, dtype=np.int64)
    np.cumsum(counts, out=lims[1:])
    D = np.empty(int(lims[-1]), dtype=np.float32)
    I = np.empty(int(lims[-1]), dtype=np.int64)
    lib.oracle_range_fill_ip(_p(q), ctypes.c_int64(nq), _p(r), ctypes.c_int64(r.shape[0]),
                             ctypes.c_int(q.shape[1]), ctypes.c_float(radius), _p(lims), _p(D), _p(I))
    return lims, D, I


def l2_normalize(x):
    x = _f32(x).copy()
    _load().oracle_l2_normalize(_p(x), ctypes.c_int64(x.shape[0]), ctypes.c_int(x.shape[1]))
    return x
