"""IVFFlatIndex — Python handle on the IVFFlat `nprobe` mode of the C ABI (rag_ivf_* in include/rag_amd.h).

Plays the role of the `faiss.IndexIVFFlat` object the reference's FAISSStore holds when FAISS_INDEX_PATH names the
file its own generator writes (reference scripts/create_test_docs.py:83-104; nprobe set at load,
src/pipeline/components/faiss_store.py:84-92; searched at :152).  All arithmetic happens in the HIP kernels.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from .index_io import IVFFlatLists


class IVFFlatIndex:
    """Rows grouped in inverted lists on one MI355X; a search visits the `nprobe` lists nearest to each query."""

    def __init__(self, lists: IVFFlatLists, device: int = 0, nprobe: int | None = None) -> None:
        self._lib = _native.lib()
        self._h = C.c_void_p()
        self.d = int(lists.centroids.shape[1])
        self.metric = int(lists.metric)
        self.device = int(device)
        self.nprobe = int(nprobe if nprobe is not None else max(1, lists.nprobe))
        _native.check(self._lib.rag_ivf_create(self.d, self.metric, int(lists.quantizer_metric), self.device,
                                               C.byref(self._h)))
        f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        cent = np.ascontiguousarray(lists.centroids, dtype=np.float32)
        rows = np.ascontiguousarray(lists.rows, dtype=np.float32)
        ids = np.ascontiguousarray(lists.ids, dtype=np.int64)
        off = np.ascontiguousarray(lists.offsets, dtype=np.int64)
        try:
            _native.check(self._lib.rag_ivf_set_lists(self._h, cent.ctypes.data_as(f32p), cent.shape[0],
                                                      rows.ctypes.data_as(f32p), ids.ctypes.data_as(i64p),
                                                      off.ctypes.data_as(i64p)))
        except Exception:
            self.close()
            raise

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rag_ivf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:  # best effort
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self) -> int:
        return int(self._lib.rag_ivf_ntotal(self._h)) if self._h else 0

    @property
    def two_stage(self) -> bool:
        """True when searches with k <= 100 go through the fp16 screening pass + exact second stage (identical results)."""
        return bool(self._h) and bool(self._lib.rag_ivf_two_stage(self._h))

    def screen_stats(self, reset: bool = False) -> dict:
        q, f, r = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        _native.check(self._lib.rag_ivf_screen_stats(self._h, C.byref(q), C.byref(f), C.byref(r), 1 if reset else 0))
        return {"queries": int(q.value), "fallbacks": int(f.value), "max_err_ratio": float(r.value)}

    @property
    def nlist(self) -> int:
        return int(self._lib.rag_ivf_nlist(self._h)) if self._h else 0

    def search(self, queries: np.ndarray, k: int, nprobe: int | None = None) -> tuple[np.ndarray, np.ndarray]:
        """(D, I) as faiss returns them: best first, int64 stored ids, -1 padded."""
        if not self._h:
            raise RuntimeError("IVFFlatIndex is closed")
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"queries must have shape (nq, {self.d}), got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _native.check(self._lib.rag_ivf_search(self._h, q.ctypes.data_as(C.POINTER(C.c_float)), nq, int(k),
                                               int(nprobe if nprobe is not None else self.nprobe),
                                               D.ctypes.data_as(C.POINTER(C.c_float)),
                                               I.ctypes.data_as(C.POINTER(C.c_int64))))
        return D, I

    def search_device(self, q_ptr: int, nq: int, k: int, scores_ptr: int, ids_ptr: int, stream: int = 0,
                      nprobe: int | None = None) -> None:
        """The same search on device pointers, enqueued on `stream` (rag_ivf_search_device): queries [nq][d] fp32,
        scores [nq][k] fp32 and ids [nq][k] int64 in device memory; no host round trip."""
        if not self._h:
            raise RuntimeError("IVFFlatIndex is closed")
        _native.check(self._lib.rag_ivf_search_device(self._h, C.c_void_p(q_ptr), int(nq), int(k),
                                                      int(nprobe if nprobe is not None else self.nprobe),
                                                      C.c_void_p(scores_ptr), C.c_void_p(ids_ptr), C.c_void_p(stream)))

    def search_from_device(self, q_ptr: int, nq: int, k: int, stream: int = 0,
                           nprobe: int | None = None) -> tuple[np.ndarray, np.ndarray]:
        """(D, I) on the host for queries that are already in device memory (an embedder's device-resident result on
        `stream`): rag_ivf_search_device_host_out — one read-back per array, one wait."""
        if not self._h:
            raise RuntimeError("IVFFlatIndex is closed")
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _native.check(self._lib.rag_ivf_search_device_host_out(
            self._h, C.c_void_p(q_ptr), int(nq), int(k), int(nprobe if nprobe is not None else self.nprobe),
            D.ctypes.data_as(C.POINTER(C.c_float)), I.ctypes.data_as(C.POINTER(C.c_int64)), C.c_void_p(stream)))
        return D, I

    def search_gather_device(self, comm: C.c_void_p, q_ptr: int, nq: int, k: int, mode: int, c_args: tuple, stream: int,
                             comm_stream: int | None) -> None:
        """The shard step of a sharded IVF index as ONE call (rag_ivf_search_gather_device): local nprobe search ->
        all-gather on `comm` -> merge.  `mode` (the flat index's fallback choice) has no meaning here: the list is final."""
        if not self._h:
            raise RuntimeError("IVFFlatIndex is closed")
        _native.check(self._lib.rag_ivf_search_gather_device(
            self._h, comm, C.c_void_p(q_ptr), int(nq), int(k), int(self.nprobe), *c_args, C.c_void_p(stream),
            C.c_void_p(comm_stream) if comm_stream is not None else None))
