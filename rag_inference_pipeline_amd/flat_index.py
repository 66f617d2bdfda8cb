"""FlatIndex — Python handle on the C-ABI flat index (rag_index_* in include/rag_amd.h).

Plays the role of the `faiss.Index` object that the reference's FAISSStore holds
(reference src/pipeline/components/faiss_store.py:37, :66-69, :152): `search(x, k) -> (D, I)`,
`ntotal`, `d`.  All arithmetic happens in the HIP kernels; this file only marshals buffers.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from ._native import (METRIC_INNER_PRODUCT, METRIC_L2,  # noqa: F401  (re-exported)
                      SEARCH_DEFAULT, SEARCH_DEFER_FALLBACK, SEARCH_EXACT_ONE_PASS)

SCREEN_OFF, SCREEN_FP16, SCREEN_INACTIVE = 0, 1, 2  # include/rag_amd.h RAG_SCREEN_*


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class FlatIndex:
    """Exhaustive fp32 index on one MI355X.  ids are insertion row numbers (+ id_offset)."""

    def __init__(self, d: int, metric: int = METRIC_INNER_PRODUCT, device: int = 0) -> None:
        self._lib = _native.lib()
        self._h = C.c_void_p()
        _native.check(self._lib.rag_index_create(int(d), int(metric), int(device), C.byref(self._h)))
        self.d = int(d)
        self.metric = int(metric)
        self.device = int(device)

    # -- lifecycle ------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rag_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:  # best effort
        try:
            self.close()
        except Exception:
            pass

    def _handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("FlatIndex is closed")
        return self._h

    # -- building -------------------------------------------------------------------------
    def reserve(self, n_total: int) -> None:
        _native.check(self._lib.rag_index_reserve(self._handle(), int(n_total)))

    def add(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.d:
            raise ValueError(f"rows must have shape (n, {self.d}), got {rows.shape}")
        _native.check(self._lib.rag_index_add(self._handle(), _f32p(rows), rows.shape[0]))

    def add_device(self, dev_ptr: int, n: int, stream: int = 0) -> None:
        """Append n rows already resident on this device (e.g. a torch tensor's data_ptr())."""
        _native.check(self._lib.rag_index_add_device(self._handle(), C.c_void_p(dev_ptr), int(n),
                                                     C.c_void_p(stream)))

    def add_synthetic(self, n: int, seed: int = 1234, row_number_offset: int = 0) -> None:
        _native.check(self._lib.rag_index_add_synthetic(self._handle(), int(n), int(seed),
                                                        int(row_number_offset)))

    def set_id_offset(self, offset: int) -> None:
        _native.check(self._lib.rag_index_set_id_offset(self._handle(), int(offset)))

    @property
    def ntotal(self) -> int:
        return int(self._lib.rag_index_ntotal(self._handle()))

    def get_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), dtype=np.float32)
        _native.check(self._lib.rag_index_get_rows(self._handle(), int(row0), int(n), _f32p(out)))
        return out

    # -- search ---------------------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """(D, I): D (nq, k) float32 best-first, I (nq, k) int64, -1 padded (faiss convention)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"queries must have shape (nq, {self.d}), got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _native.check(self._lib.rag_index_search(self._handle(), _f32p(q), nq, int(k), _f32p(D),
                                                 I.ctypes.data_as(C.POINTER(C.c_int64))))
        return D, I

    def search_device(self, q_ptr: int, nq: int, k: int, out_scores_ptr: int, out_ids_ptr: int,
                      stream: int = 0) -> None:
        """Asynchronous search on device buffers (pointers as ints, stream = hipStream_t)."""
        _native.check(self._lib.rag_index_search_device(
            self._handle(), C.c_void_p(q_ptr), int(nq), int(k), C.c_void_p(out_scores_ptr),
            C.c_void_p(out_ids_ptr), C.c_void_p(stream)))

    def search_device_ex(self, q_ptr: int, nq: int, k: int, out_scores_ptr: int, out_ids_ptr: int,
                         mode: int = SEARCH_DEFAULT, flag_ptr: int = 0, stream: int = 0) -> None:
        """search_device with the two-stage fallback policy chosen by the caller (rag_index_search_device_ex):
        SEARCH_DEFER_FALLBACK leaves out the fallback launches and reports through the device word at
        `flag_ptr` (1 = repeat the batch with SEARCH_EXACT_ONE_PASS)."""
        _native.check(self._lib.rag_index_search_device_ex(
            self._handle(), C.c_void_p(q_ptr), int(nq), int(k), C.c_void_p(out_scores_ptr), C.c_void_p(out_ids_ptr),
            int(mode), C.c_void_p(flag_ptr or None), C.c_void_p(stream)))

    def search_gather_device(self, comm: C.c_void_p, q_ptr: int, nq: int, k: int, mode: int, c_args: tuple, stream: int,
                             comm_stream: int | None) -> None:
        """The whole shard step as ONE call (rag_index_search_gather_device): local search -> all-gather on `comm` ->
        flagged merge.  `c_args`: the slot's (pack, gathered, out_scores, out_ids, any_flag, host_mirror) pointers."""
        _native.check(self._lib.rag_index_search_gather_device(
            self._handle(), comm, C.c_void_p(q_ptr), int(nq), int(k), int(mode), *c_args, C.c_void_p(stream),
            C.c_void_p(comm_stream) if comm_stream is not None else None))

    def search_from_device(self, q_ptr: int, nq: int, k: int, stream: int = 0) -> tuple[np.ndarray, np.ndarray]:
        """(D, I) on the host for queries that are already in device memory (an embedder's device-resident
        result on `stream`): rag_index_search_device_host_out — one read-back per array, one sync."""
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _native.check(self._lib.rag_index_search_device_host_out(
            self._handle(), C.c_void_p(q_ptr), int(nq), int(k), _f32p(D), I.ctypes.data_as(C.POINTER(C.c_int64)),
            C.c_void_p(stream)))
        return D, I

    # -- two-stage exact search ------------------------------------------------------------
    def set_screening(self, mode: int | bool = SCREEN_FP16) -> None:
        """Keep a scaled fp16 copy of the corpus and answer k <= 100 searches by screening it, then
        re-scoring the candidates in fp32 under a per-query exactness certificate (failures fall back
        to the fp32 scan on the device).  Results are identical in either mode."""
        _native.check(self._lib.rag_index_set_screening(self._handle(), int(mode)))

    @property
    def screening(self) -> int:
        """SCREEN_OFF, SCREEN_FP16 (active) or SCREEN_INACTIVE (requested, corpus out of range)."""
        return int(self._lib.rag_index_screening(self._handle()))

    def screen_stats(self, reset: bool = False) -> dict:
        q, f, r = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        _native.check(self._lib.rag_index_screen_stats(self._handle(), C.byref(q), C.byref(f), C.byref(r),
                                                       1 if reset else 0))
        return {"queries": int(q.value), "fallbacks": int(f.value), "max_err_ratio": float(r.value)}

    def set_cu_budget(self, n_cus: int = 0) -> None:
        """How many compute units a search may count on (the persistent scan kernel's grid); 0 = the whole device.
        For searches enqueued on a stream restricted to a share of the chip (create_masked_stream)."""
        _native.check(self._lib.rag_index_set_cu_budget(self._handle(), int(n_cus)))

    # -- profiling ------------------------------------------------------------------------
    def profile_enable(self, on: bool = True) -> None:
        _native.check(self._lib.rag_index_profile_enable(self._handle(), 1 if on else 0))

    def profile(self, reset: bool = False) -> tuple[float, int]:
        """(accumulated scan-kernel milliseconds from HIP events, number of scan launches)."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        _native.check(self._lib.rag_index_profile(self._handle(), C.byref(ms), C.byref(n),
                                                  1 if reset else 0))
        return float(ms.value), int(n.value)

    @staticmethod
    def max_k(d: int, nq: int = 32) -> int:
        return int(_native.lib().rag_index_max_k(int(d), int(nq)))


def merge_topk_device(device: int, metric: int, n_shards: int, nq: int, k: int, scores_ptr: int,
                      ids_ptr: int, out_scores_ptr: int, out_ids_ptr: int, stream: int = 0) -> None:
    """Merge all-gathered per-shard lists on the device (rag_merge_topk_device)."""
    _native.check(_native.lib().rag_merge_topk_device(
        int(device), int(metric), int(n_shards), int(nq), int(k), C.c_void_p(scores_ptr),
        C.c_void_p(ids_ptr), C.c_void_p(out_scores_ptr), C.c_void_p(out_ids_ptr), C.c_void_p(stream)))


def merge_topk_packed_flagged_device(device: int, metric: int, n_shards: int, nq: int, k: int, packed_ptr: int,
                                     shard_stride_bytes: int, scores_offset_bytes: int, flag_offset_bytes: int,
                                     out_scores_ptr: int, out_ids_ptr: int, any_flag_ptr: int, host_mirror_ptr: int = 0,
                                     stream: int = 0) -> None:
    """Merge out of the all-gather receive buffer and OR the shards' "not final" words into one device word
    (rag_merge_topk_packed_flagged_device); with `host_mirror_ptr` (pinned host memory, one block's layout) the
    kernel writes the result to the host as well."""
    _native.check(_native.lib().rag_merge_topk_packed_flagged_device(
        int(device), int(metric), int(n_shards), int(nq), int(k), C.c_void_p(packed_ptr), int(shard_stride_bytes),
        int(scores_offset_bytes), int(flag_offset_bytes), C.c_void_p(out_scores_ptr), C.c_void_p(out_ids_ptr),
        C.c_void_p(any_flag_ptr), C.c_void_p(host_mirror_ptr or None), C.c_void_p(stream)))


def merge_topk_packed_device(device: int, metric: int, n_shards: int, nq: int, k: int, packed_ptr: int,
                             shard_stride_bytes: int, scores_offset_bytes: int, out_scores_ptr: int,
                             out_ids_ptr: int, stream: int = 0) -> None:
    """Merge straight out of the all-gather receive buffer (rag_merge_topk_packed_device)."""
    _native.check(_native.lib().rag_merge_topk_packed_device(
        int(device), int(metric), int(n_shards), int(nq), int(k), C.c_void_p(packed_ptr), int(shard_stride_bytes),
        int(scores_offset_bytes), C.c_void_p(out_scores_ptr), C.c_void_p(out_ids_ptr), C.c_void_p(stream)))


def create_masked_stream(device: int, first_cu: int, n_cus: int) -> int:
    """A hipStream_t (as an integer) whose kernels run on the CUs [first_cu, first_cu + n_cus) of the CU-mask enumeration
    (rag_stream_create_masked); wrap it with torch.cuda.ExternalStream to use it from torch.  destroy_stream releases it."""
    lib = _native.lib()
    st = C.c_void_p()
    _native.check(lib.rag_stream_create_masked(int(device), int(first_cu), int(n_cus), C.byref(st)))
    return int(st.value)


def destroy_stream(device: int, stream: int) -> None:
    _native.check(_native.lib().rag_stream_destroy(int(device), C.c_void_p(int(stream))))


def stream_wait(device: int, waiter: int, signaler: int) -> None:
    """Work enqueued on `waiter` from now on starts when what is on `signaler` now has finished (rag_stream_wait)."""
    _native.check(_native.lib().rag_stream_wait(int(device), C.c_void_p(int(waiter)), C.c_void_p(int(signaler))))


def device_cu_count(device: int) -> int:
    n = C.c_int32(0)
    _native.check(_native.lib().rag_device_cu_count(int(device), C.byref(n)))
    return int(n.value)
