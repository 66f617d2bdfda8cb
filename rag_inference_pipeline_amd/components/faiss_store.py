"""FAISSStore — drop-in for the reference's vector-index component, backed by the HIP flat index.

Same constructor, methods, properties and exception types/messages as the reference class
(reference src/pipeline/components/faiss_store.py:22-189; messages pinned by
tests/test_components.py:47, :92, :112, :133 and tests/test_retrieval_service.py:263):

    FAISSStore(settings).load() / .search(embeddings, k) -> (distances, indices) / .unload()
    .is_loaded / .index_size

What differs is only what sits underneath: rows live in HBM and `search` runs the gfx950
scan + top-k kernels through the C ABI (rag_index_search).  There is no CPU path; on a machine
without a HIP device `load()` raises.

Multi-GPU: when the process belongs to an initialised torch.distributed group of more than one
rank (one process per GPU), `load()` keeps only this rank's contiguous row range of the file
(sharded.shard_range) and `search()` becomes collective — rank 0 serves requests exactly as before,
the other ranks call `serve_forever()` and follow (the reference itself has no
multi-device path).
"""

from __future__ import annotations

import gc
import threading
import logging
from pathlib import Path

import numpy as np

from .. import _native, index_io
from ..config import PipelineSettings, resolve_gpu_device
from ..device_embeddings import DeviceEmbeddings

logger = logging.getLogger(__name__)

_ADD_CHUNK_ROWS = 1 << 18  # rows per host->device copy when loading a file
_TWO_STAGE_MAX_D = 2048    # rag_index_set_screening covers d <= 2048


class FAISSStore:
    """Exhaustive (flat) vector index resident on one MI355X."""

    accepts_device_embeddings = True  # search() takes EmbeddingGenerator.encode_device's handle

    def __init__(self, settings: PipelineSettings) -> None:
        self.settings = settings
        self.index_path = Path(settings.faiss_index_path)
        self._index = None  # rag_inference_pipeline_amd.flat_index.FlatIndex
        self._sharded = None  # rag_inference_pipeline_amd.sharded.ShardedFlatIndex when world > 1
        self._ivf = None  # rag_inference_pipeline_amd.ivf_index.IVFFlatIndex in the nprobe mode of an IwFl file
        self._ntotal = 0
        self._is_loaded = False
        self._share_stream = 0     # the search stream of a partitioned chip (settings.encoder_cus), made at first use
        self._share_checked = False
        self._share_lock = threading.Lock()

    def _search_share(self) -> int:
        """settings.encoder_cus > 0 on one GPU: searches that follow the embedder on the device run on a stream that owns
        the CUs the encoder's stream does not, and the index plans its launches for that many (0: no partition)."""
        if not self._share_checked:
            with self._share_lock:   # (batches run on pool threads: the first two may arrive together)
                if not self._share_checked:
                    share = int(getattr(self.settings, "encoder_cus", 0) or 0)
                    # (only reached from the device hand-off branch of search(): a store that never sees a
                    # DeviceEmbeddings batch never partitions.  Once it has, host-array searches keep planning for the
                    # index's share as well — a 6 % slower scan at 224 of 256 CUs — rather than flip the budget per call.)
                    if share > 0 and self._sharded is None and self._index is not None:
                        from ..flat_index import create_masked_stream, device_cu_count
                        total = device_cu_count(self._index.device)
                        if 0 < share < total:
                            self._share_stream = create_masked_stream(self._index.device, share, total - share)
                            self._index.set_cu_budget(total - share)
                    self._share_checked = True
        return self._share_stream

    def load(self) -> None:
        """Read the index file and pin its rows in HBM (reference load(): faiss_store.py:40-111)."""
        if self._is_loaded:
            logger.info("FAISS index already loaded")
            return
        if not self.index_path.exists():
            raise FileNotFoundError(f"FAISS index not found at {self.index_path}")
        logger.info("Loading FAISS index from %s", self.index_path)
        try:
            from ..flat_index import SCREEN_FP16, FlatIndex  # raises if librag_amd.so is missing: no fallback

            default_metric = index_io.metric_from_name(getattr(self.settings, "faiss_metric", "ip"))
            if self._load_ivf_nprobe_mode():
                return
            rows, metric = index_io.read_index_file(
                self.index_path, default_metric, mmap=bool(getattr(self.settings, "faiss_use_mmap", False)))
            n, d = rows.shape
            device = resolve_gpu_device(self.settings)  # LOCAL_RANK / current device when one process per GPU
            rank, world = self._dist_rank_world()
            row_lo, row_hi = 0, n
            if world > 1:
                from ..sharded import shard_range

                row_lo, row_hi = shard_range(n, rank, world)
            index = FlatIndex(d, metric, device=device)
            index.reserve(row_hi - row_lo)
            for lo in range(row_lo, row_hi, _ADD_CHUNK_ROWS):
                index.add(np.ascontiguousarray(rows[lo:min(lo + _ADD_CHUNK_ROWS, row_hi)], dtype=np.float32))
            index.set_id_offset(row_lo)
            if bool(getattr(self.settings, "faiss_two_stage", True)) and d <= _TWO_STAGE_MAX_D:
                try:  # an optimisation, never a reason not to serve: e.g. no room for the extra fp16 copy
                    index.set_screening(SCREEN_FP16)
                    if index.screening != SCREEN_FP16:
                        logger.warning("two-stage search inactive: corpus values outside the range its error "
                                       "bound covers; searches use the one-pass fp32 scan")
                except _native.RagAmdError as exc:
                    logger.warning("two-stage search not enabled (%s); searches use the one-pass fp32 scan", exc)
            self._index = index
            self._ntotal = n
            if world > 1:
                from ..sharded import ShardedFlatIndex

                self._sharded = ShardedFlatIndex(index, metric, device=device, dim=d,
                                                 max_batch=max(1, int(getattr(self.settings, "retrieval_batch_size", 32))))
            self._is_loaded = True
            logger.info("FAISS index loaded successfully: %d vectors, dimension=%d (rows %d..%d on this rank)",
                        n, d, row_lo, row_hi)
            # nprobe / precomputed tables (reference :84-100) have no meaning for an exhaustive
            # scan: every row is visited, which is the nprobe == nlist limit.
            if row_hi > row_lo:  # warm-up search, as the reference does (:103-107); local, not collective
                index.search(np.zeros((1, d), dtype=np.float32), 1)
        except FileNotFoundError:
            raise
        except Exception as exc:
            logger.exception("Failed to load FAISS index")
            self._index = None
            self._is_loaded = False
            raise RuntimeError(f"FAISS index loading failed: {exc}") from exc

    def _load_ivf_nprobe_mode(self) -> bool:
        """RAG_AMD_IVF_MODE=nprobe on an IndexIVFFlat file: keep the inverted lists and search as the reference does with
        index.nprobe = FAISS_NPROBE (faiss_store.py:84-92).  False: the caller loads the file exhaustively."""
        mode = str(getattr(self.settings, "faiss_ivf_mode", "exhaustive")).strip().lower()
        if mode not in ("exhaustive", "nprobe"):
            raise ValueError(f"RAG_AMD_IVF_MODE must be 'exhaustive' or 'nprobe', got {mode!r}")
        if mode != "nprobe":
            return False
        with open(self.index_path, "rb") as fh:
            if fh.read(4) != b"IwFl":
                return False   # not an IVF file: nprobe has no meaning, as for the reference's flat indexes (:84)
        from ..ivf_index import IVFFlatIndex

        lists = index_io.read_ivfflat_lists(self.index_path)
        nprobe = max(1, int(getattr(self.settings, "faiss_nprobe", 64)))
        device = resolve_gpu_device(self.settings)
        rank, world = self._dist_rank_world()
        # a sharded deployment: every rank keeps the centroids and its share of EVERY list (IVFFlatLists.shard), probes
        # the same lists and the per-rank top-k lists are merged as in the flat mode (one all-gather per batch)
        local = lists.shard(rank, world) if world > 1 else lists
        self._ivf = IVFFlatIndex(local, device=device, nprobe=nprobe)
        self._ntotal = lists.ntotal
        if world > 1:
            from ..sharded import ShardedFlatIndex

            self._sharded = ShardedFlatIndex(self._ivf, lists.metric, device=device, dim=int(lists.centroids.shape[1]),
                                             max_batch=max(1, int(getattr(self.settings, "retrieval_batch_size", 32))))
        self._is_loaded = True
        logger.info("Set FAISS nprobe=%d", min(nprobe, lists.nlist))
        logger.info("IVF list scan: %s", "two-stage (fp16 screening pass + exact second stage, identical results)"
                    if self._ivf.two_stage else "one-stage fp32 (RAG_AMD_IVF_TWO_STAGE=0, or the corpus is outside the "
                                                "range the screening pass's error bound covers)")
        logger.info("FAISS index loaded successfully: %d vectors, dimension=%d (IVFFlat, %d lists; %d rows on this rank)",
                    lists.ntotal, lists.centroids.shape[1], lists.nlist, local.ntotal)
        if lists.ntotal:
            self._ivf.search(np.zeros((1, lists.centroids.shape[1]), dtype=np.float32), 1)   # warm-up (:103-107)
        return True

    def search(self, embeddings: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """(distances, indices), each (batch, k); reference search(): faiss_store.py:113-158."""
        if not self._is_loaded or (self._index is None and self._ivf is None):
            raise RuntimeError("FAISS index not loaded. Call load() first.")
        if embeddings.ndim != 2:
            raise ValueError(f"Embeddings must be 2D array, got shape {embeddings.shape}")
        if embeddings.shape[1] != self.settings.faiss_dim:
            raise ValueError(
                f"Embedding dimension mismatch: expected {self.settings.faiss_dim}, got {embeddings.shape[1]}")
        logger.debug("Searching FAISS index with %d queries, k=%d", embeddings.shape[0], k)
        try:
            if self._ivf is not None and self._sharded is None:
                if isinstance(embeddings, DeviceEmbeddings):
                    if embeddings.device == self._ivf.device:   # the embedder's result never left HBM (as below)
                        res = self._ivf.search_from_device(embeddings.data_ptr, embeddings.shape[0], k, embeddings.stream)
                        embeddings.settled()
                        if embeddings.valid():
                            return res
                    embeddings = embeddings.numpy()
                return self._ivf.search(np.ascontiguousarray(embeddings, dtype=np.float32), k)
            if isinstance(embeddings, DeviceEmbeddings):
                # the embedder's result never left HBM: the search is enqueued behind it on its stream and only
                # ids and scores come back (a sharded deployment ships host bytes in its request message)
                if self._sharded is None and embeddings.device == self._index.device:
                    stream = embeddings.stream
                    if self._search_share():
                        # the search runs on the index's share of the chip, behind the point the embedder's stream has
                        # reached: the embedder's next batch can run beside it (settings.encoder_cus)
                        from ..flat_index import stream_wait
                        stream_wait(self._index.device, self._share_stream, embeddings.stream)
                        stream = self._share_stream
                    res = self._index.search_from_device(embeddings.data_ptr, embeddings.shape[0], k, stream)
                    embeddings.settled()  # the call has waited for the stream (and, through it, the embedder's)
                    if embeddings.valid():
                        return res
                    # the encoder's pass left fp16's range (device_embeddings.py): the host path repeats it exactly
                    embeddings = embeddings.numpy()
                else:
                    embeddings = embeddings.numpy()
            # the reference converts unconditionally (faiss_store.py:147); a C-contiguous fp32 array is already
            # what the C ABI reads, so it is passed as it is (search never writes to it)
            if not (isinstance(embeddings, np.ndarray) and embeddings.dtype == np.float32 and embeddings.flags.c_contiguous):
                embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
            if self._sharded is not None:
                return self._sharded.leader_search(embeddings, k)
            return self._index.search(embeddings, k)
        except Exception:
            logger.exception("FAISS search failed")
            raise

    @staticmethod
    def _dist_rank_world() -> tuple[int, int]:
        try:
            import torch.distributed as dist
        except ImportError:
            return 0, 1
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    @property
    def shard_link(self):
        """The serving channel of a sharded deployment (None on one GPU): what `Reranker.attach_shard_link`
        takes so that rerank batches are split over the same ranks."""
        return self._sharded

    def serve_forever(self, reranker=None) -> int:
        """Ranks other than 0 of a sharded deployment: join the leader's searches — and, when this
        rank's loaded `reranker` is passed, its query-sharded rerank passes — until it unloads."""
        if not self._is_loaded or self._sharded is None:
            raise RuntimeError("FAISS index not loaded in sharded mode. Call load() first.")
        if reranker is not None:
            reranker.attach_shard_link(self._sharded)
        return self._sharded.follower_loop()

    def unload(self) -> None:
        if self._is_loaded:
            logger.info("Unloading FAISS index")
            if self._sharded is not None:
                self._sharded.shutdown()
                self._sharded.close()   # the own RCCL communicator(s), once nothing is in flight
                self._sharded = None
            dev = self._index.device if self._index is not None else 0
            if self._index is not None:
                self._index.close()
            if self._ivf is not None:
                self._ivf.close()
                self._ivf = None
            if self._share_stream:
                from ..flat_index import destroy_stream
                destroy_stream(dev, self._share_stream)
                self._share_stream = 0
            self._share_checked = False
            self._index = None
            self._is_loaded = False
            gc.collect()

    @property
    def is_loaded(self) -> bool:
        return self._is_loaded

    @property
    def index_size(self) -> int:
        if not self._is_loaded or (self._index is None and self._ivf is None):
            return 0
        return int(self._ntotal)

    def __repr__(self) -> str:
        status = "loaded" if self._is_loaded else "not loaded"
        size = self.index_size if self._is_loaded else "unknown"
        return f"FAISSStore(path={self.index_path}, status={status}, size={size})"
