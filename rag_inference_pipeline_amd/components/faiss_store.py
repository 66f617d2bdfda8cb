"""FAISSStore — drop-in for the reference's vector-index component, backed by the HIP flat index.

Same constructor, methods, properties and exception types/messages as the reference class
(reference src/pipeline/components/faiss_store.py:22-189; messages pinned by
tests/test_components.py:47, :92, :112, :133 and tests/test_retrieval_service.py:263):

    FAISSStore(settings).load() / .search(embeddings, k) -> (distances, indices) / .unload()
    .is_loaded / .index_size

What differs is only what sits underneath: rows live in HBM and `search` runs the gfx950
scan + top-k kernels through the C ABI (rag_index_search).  There is no CPU path; on a machine
without a HIP device `load()` raises.
"""

from __future__ import annotations

import gc
import logging
from pathlib import Path

import numpy as np

from .. import index_io
from ..config import PipelineSettings

logger = logging.getLogger(__name__)

_ADD_CHUNK_ROWS = 1 << 18  # rows per host->device copy when loading a file


class FAISSStore:
    """Exhaustive (flat) vector index resident on one MI355X."""

    def __init__(self, settings: PipelineSettings) -> None:
        self.settings = settings
        self.index_path = Path(settings.faiss_index_path)
        self._index = None  # rag_inference_pipeline_amd.flat_index.FlatIndex
        self._is_loaded = False

    def load(self) -> None:
        """Read the index file and pin its rows in HBM (reference load(): faiss_store.py:40-111)."""
        if self._is_loaded:
            logger.info("FAISS index already loaded")
            return
        if not self.index_path.exists():
            raise FileNotFoundError(f"FAISS index not found at {self.index_path}")
        logger.info("Loading FAISS index from %s", self.index_path)
        try:
            from ..flat_index import FlatIndex  # raises if librag_amd.so is missing: no fallback

            default_metric = index_io.metric_from_name(getattr(self.settings, "faiss_metric", "ip"))
            rows, metric = index_io.read_index_file(
                self.index_path, default_metric, mmap=bool(getattr(self.settings, "faiss_use_mmap", False)))
            n, d = rows.shape
            index = FlatIndex(d, metric, device=int(getattr(self.settings, "gpu_device", 0)))
            index.reserve(n)
            for lo in range(0, n, _ADD_CHUNK_ROWS):
                index.add(np.ascontiguousarray(rows[lo:lo + _ADD_CHUNK_ROWS], dtype=np.float32))
            self._index = index
            self._is_loaded = True
            logger.info("FAISS index loaded successfully: %d vectors, dimension=%d", n, d)
            # nprobe / precomputed tables (reference :84-100) have no meaning for an exhaustive
            # scan: every row is visited, which is the nprobe == nlist limit.
            if n > 0:  # warm-up search, as the reference does (:103-107)
                index.search(np.zeros((1, d), dtype=np.float32), 1)
        except FileNotFoundError:
            raise
        except Exception as exc:
            logger.exception("Failed to load FAISS index")
            self._index = None
            self._is_loaded = False
            raise RuntimeError(f"FAISS index loading failed: {exc}") from exc

    def search(self, embeddings: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """(distances, indices), each (batch, k); reference search(): faiss_store.py:113-158."""
        if not self._is_loaded or self._index is None:
            raise RuntimeError("FAISS index not loaded. Call load() first.")
        if embeddings.ndim != 2:
            raise ValueError(f"Embeddings must be 2D array, got shape {embeddings.shape}")
        if embeddings.shape[1] != self.settings.faiss_dim:
            raise ValueError(
                f"Embedding dimension mismatch: expected {self.settings.faiss_dim}, got {embeddings.shape[1]}")
        embeddings = embeddings.astype("float32")
        logger.debug("Searching FAISS index with %d queries, k=%d", embeddings.shape[0], k)
        try:
            return self._index.search(embeddings, k)
        except Exception:
            logger.exception("FAISS search failed")
            raise

    def unload(self) -> None:
        if self._is_loaded:
            logger.info("Unloading FAISS index")
            if self._index is not None:
                self._index.close()
            self._index = None
            self._is_loaded = False
            gc.collect()

    @property
    def is_loaded(self) -> bool:
        return self._is_loaded

    @property
    def index_size(self) -> int:
        if not self._is_loaded or self._index is None:
            return 0
        return int(self._index.ntotal)

    def __repr__(self) -> str:
        status = "loaded" if self._is_loaded else "not loaded"
        size = self.index_size if self._is_loaded else "unknown"
        return f"FAISSStore(path={self.index_path}, status={status}, size={size})"
