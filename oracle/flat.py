"""ctypes front end of oracle/flat_oracle.c.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (rag_inference_pipeline_amd) must never do so.  Parity status of the oracle
itself: see the header of flat_oracle.c ("parity unpinned": the reference's arithmetic lives in
faiss-cpu 1.13.1, absent here, and the reference's tests pin no search result).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

METRIC_IP = 0
METRIC_L2 = 1


def build(force: bool = False) -> str:
    """Compile liboracle.so with the Makefile next to this file (gcc only, seconds)."""
    src = os.path.join(_HERE, "flat_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B" if force else "-s"], check=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        _lib.rago_dot.restype = C.c_float
        _lib.rago_dot.argtypes = [f32p, f32p, C.c_int32]
        _lib.rago_dot_f64.restype = C.c_double
        _lib.rago_dot_f64.argtypes = [f32p, f32p, C.c_int32]
        _lib.rago_scores.restype = C.c_int
        _lib.rago_scores.argtypes = [f32p, C.c_int64, C.c_int32, f32p, C.c_int32, C.c_int32, f32p]
        _lib.rago_search.restype = C.c_int
        _lib.rago_search.argtypes = [f32p, C.c_int64, C.c_int32, C.c_int32, f32p, C.c_int32,
                                     C.c_int32, C.c_int64, C.c_int32, f32p, i64p]
        _lib.rago_merge.restype = C.c_int
        _lib.rago_merge.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, i64p, f32p, i64p]
        _lib.rago_synth_rows.restype = C.c_int
        _lib.rago_synth_rows.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, f32p]
        _lib.rago_num_threads.restype = C.c_int
    return _lib


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def dot(x: np.ndarray, q: np.ndarray) -> float:
    x, q = _f32(x), _f32(q)
    return float(lib().rago_dot(_p(x, C.c_float), _p(q, C.c_float), x.shape[0]))


def scores(X: np.ndarray, Q: np.ndarray, metric: int = METRIC_IP) -> np.ndarray:
    """nq x N matrix of canonical ranking scores (small inputs only)."""
    X, Q = _f32(X), _f32(Q)
    out = np.empty((Q.shape[0], X.shape[0]), dtype=np.float32)
    lib().rago_scores(_p(X, C.c_float), X.shape[0], X.shape[1], _p(Q, C.c_float), Q.shape[0],
                      metric, _p(out, C.c_float))
    return out


def search(X: np.ndarray, Q: np.ndarray, k: int, metric: int = METRIC_IP, id_offset: int = 0,
           nthreads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """Exact top-k: the oracle for FAISSStore.search (reference faiss_store.py:113-158)."""
    X, Q = _f32(X), _f32(Q)
    if X.ndim != 2 or Q.ndim != 2 or (X.shape[0] and X.shape[1] != Q.shape[1]):
        raise ValueError("shape mismatch")
    D = np.empty((Q.shape[0], k), dtype=np.float32)
    I = np.empty((Q.shape[0], k), dtype=np.int64)
    rc = lib().rago_search(_p(X, C.c_float), X.shape[0], Q.shape[1], metric, _p(Q, C.c_float),
                           Q.shape[0], k, id_offset, nthreads, _p(D, C.c_float), _p(I, C.c_int64))
    if rc != 0:
        raise RuntimeError("rago_search failed")
    return D, I


def merge(scores_: np.ndarray, ids: np.ndarray, metric: int = METRIC_IP) -> tuple[np.ndarray, np.ndarray]:
    """Merge (n_shards, nq, k) per-shard lists into (nq, k)."""
    s, i = _f32(scores_), np.ascontiguousarray(ids, dtype=np.int64)
    g, nq, k = s.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = lib().rago_merge(metric, g, nq, k, _p(s, C.c_float), _p(i, C.c_int64), _p(D, C.c_float),
                          _p(I, C.c_int64))
    if rc != 0:
        raise RuntimeError("rago_merge failed")
    return D, I


def ivf_search(centroids: np.ndarray, quantizer_metric: int, rows: np.ndarray, ids: np.ndarray, offsets: np.ndarray,
               Q: np.ndarray, k: int, nprobe: int, metric: int = METRIC_L2, nthreads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """Oracle of the IVFFlat `nprobe` mode (faiss IndexIVF::search as the reference uses it: index.nprobe set at load,
    faiss_store.py:84-92; index.search, faiss_store.py:152; file written by scripts/create_test_docs.py:83-104).
      1. quantizer.search(Q, nprobe): the flat oracle over the centroids under the quantizer's metric;
      2. per query, the candidates are the rows of those lists (`rows` / `ids` in list order, list l =
         [offsets[l], offsets[l + 1]));
      3. exact top-k of the candidates with the flat oracle's canonical score, ranked (score, ascending stored id).
    faiss leaves the order among equal scores to its heap's visiting order; the candidate set is what nprobe defines.
    Parity status: unpinned against faiss itself, like the flat oracle.  Pure numpy over rago_search; small cases."""
    centroids, rows, Q = _f32(centroids), _f32(rows), _f32(Q)
    nlist = centroids.shape[0]
    nprobe = min(int(nprobe), nlist)
    _, probe = search(centroids, Q, nprobe, quantizer_metric, nthreads=nthreads)
    D = np.full((Q.shape[0], k), np.finfo(np.float32).max if metric == METRIC_L2 else -np.finfo(np.float32).max,
                dtype=np.float32)
    I = np.full((Q.shape[0], k), -1, dtype=np.int64)
    for qi in range(Q.shape[0]):
        sel = np.concatenate([np.arange(offsets[l], offsets[l + 1]) for l in probe[qi] if l >= 0] or [np.zeros(0, int)])
        if sel.size == 0:
            continue
        sel = sel[np.argsort(ids[sel], kind="stable")]        # ascending stored id: the local row number then breaks
        Dq, Iq = search(rows[sel], Q[qi:qi + 1], k, metric, nthreads=nthreads)   # ties exactly as the stored id would
        ok = Iq[0] >= 0
        D[qi, ok] = Dq[0, ok]
        I[qi, ok] = ids[sel][Iq[0, ok]]
    return D, I


def synth_rows(seed: int, row0: int, n: int, d: int) -> np.ndarray:
    """Rows [row0, row0+n) of the deterministic synthetic corpus (see rago_synth_rows)."""
    out = np.empty((n, d), dtype=np.float32)
    lib().rago_synth_rows(seed, row0, n, d, _p(out, C.c_float))
    return out


def num_threads() -> int:
    return int(lib().rago_num_threads())
