"""RetrievalExecutor — the batch function that drives the hot path.

Counterpart of the reference's RetrievalExecutor (src/pipeline/services/retrieval/api.py:295-598):
requests are grouped by a BatchScheduler; each batch runs, on a worker thread,

    embeddings (precomputed or encode())  ->  index search  ->  document fetch  ->  optional rerank

and yields one RetrievalResponseItem per request, in request order.  Semantics kept:
  * the FIRST request decides between the precomputed-embedding path and encode(); a later request
    without an embedding on the precomputed path is ValueError("Missing embedding in batch")
    (api.py:69-84, :351-374)
  * components are looked up by canonical alias in the registry; only the index is mandatory
    (api.py:218-238, :361-363, :434, :537-542)
  * without a document store, or in `id_only` mode, documents are id-only stubs (api.py:436-451)
  * scores are zipped onto documents without a length check (api.py:491, strict=False)
  * with a reranker the returned list and scores are the reranker's (api.py:495-514)
  * unless DISABLE_CACHE_FOR_PROFILING is set, index results are cached per embedding (api.py:392-425)
What is different: rerank runs as ONE batched call over the whole batch (rerank_batch) instead of
a thread pool of per-query calls (api.py:579-589) — the GPU cross-encoder batches all pairs.
In `compressed` payload mode (api.py:516-523) the documents leave as an LZ4 frame of compact JSON —
the bytes lz4.frame.decompress + msgspec.json.decode read back on the generation node
(generation/service.py:429-431) — and `docs` is empty.
"""

from __future__ import annotations

import asyncio
import gc
import json
import hashlib
import logging
import threading
import time
from collections.abc import Sequence
from typing import Any

import numpy as np

from .batch_scheduler import Batch, BatchScheduler
from .cache import LRUCache
from .component_registry import ComponentRegistry
from .components.document_store import Document as StoreDocument
from .components.schemas import Document, fast_constructor
from .config import PipelineSettings, get_settings
from .schemas import PendingRequest, RetrievalDocument, RetrievalRequestItem, RetrievalResponseItem
from .telemetry import STAGE_DOCUMENT_FETCH, STAGE_EMBEDDING, STAGE_FAISS_SEARCH, stage_timers

logger = logging.getLogger(__name__)

_new_retrieval_doc = fast_constructor(RetrievalDocument)


def extract_embeddings_from_requests(requests: Sequence[Any]) -> np.ndarray:
    """(batch, d) float32 from the requests' precomputed embeddings: the binary form (`embedding_f32`,
    little-endian fp32 bytes) when a request carries it, else the reference's list of floats."""
    rows = []
    for req in requests:
        raw = getattr(req, "embedding_f32", None)
        if raw is not None:
            rows.append(np.frombuffer(raw, dtype="<f4"))
        elif req.embedding is not None:
            rows.append(req.embedding)
        else:
            raise ValueError("Missing embedding in batch")
    return np.array(rows).astype("float32")


class RetrievalExecutor:
    def __init__(self, registry: ComponentRegistry, settings: PipelineSettings | None = None) -> None:
        self.registry = registry
        self.settings = settings or get_settings()
        self.scheduler: BatchScheduler[RetrievalResponseItem] = BatchScheduler(
            batch_size=self.settings.retrieval_batch_size,
            max_batch_delay_ms=self.settings.retrieval_max_batch_delay_ms,
            process_batch_fn=self._process_batch,
            service_name="retrieval",
            enable_adaptive=getattr(self.settings, "enable_adaptive_batching", True),
        )
        # per-embedding result cache (reference api.py:310-315); bypassed in measurement mode
        self.cache: LRUCache[str, tuple[list[int], list[float]]] = LRUCache(
            capacity=self.settings.retrieval_cache_capacity, ttl=self.settings.cache_max_ttl,
            name="retrieval_faiss_cache")
        self._lock = threading.Lock()
        self._shard_link_checked = False
        self._gc_frozen = False

    def _attach_shard_link(self, reranker: Any) -> None:
        """One process per GPU: rerank batches are split by query over the ranks that shard the index.
        Checked at the first rerank, when the components' load hooks have run."""
        self._shard_link_checked = True
        link = getattr(self.registry.get("faiss_store"), "shard_link", None)
        if link is not None and hasattr(reranker, "attach_shard_link"):
            reranker.attach_shard_link(link)

    async def start(self) -> None:
        await self.scheduler.start()

    async def stop(self) -> None:
        await self.scheduler.stop()

    async def process_request(self, item: RetrievalRequestItem) -> RetrievalResponseItem:
        req = PendingRequest(request_id=item.request_id, query=item.query, embedding=item.embedding,
                             embedding_f32=getattr(item, "embedding_f32", None), timestamp=time.time())
        return await self.scheduler.enqueue(req)

    async def _process_batch(self, batch: Batch[RetrievalResponseItem]) -> list[RetrievalResponseItem]:
        # worker thread: the C ABI releases the GIL and the event loop keeps accepting requests
        return await asyncio.get_running_loop().run_in_executor(None, self._process_batch_sync, batch)

    # -- stages -------------------------------------------------------------------------------
    def _get_embeddings(self, batch: Batch[RetrievalResponseItem]) -> np.ndarray:
        first = batch.requests[0]
        if first.embedding is not None or getattr(first, "embedding_f32", None) is not None:
            return extract_embeddings_from_requests(batch.requests)
        embedder = self.registry.get("embedding_generator")
        if not embedder:
            raise RuntimeError("Embedding generator not available")
        with stage_timers.track(STAGE_EMBEDDING):
            # measurement mode (no embedding-keyed result cache): an embedder and an index that can hand the batch
            # over in device memory do so — the embeddings are never needed on the host
            if (getattr(self.settings, "disable_cache_for_profiling", True) and hasattr(embedder, "encode_device")
                    and getattr(self.registry.get("faiss_store"), "accepts_device_embeddings", False)):
                return embedder.encode_device([req.query for req in batch.requests])
            return embedder.encode([req.query for req in batch.requests])

    def clear_cache(self) -> None:
        with self._lock:
            self.cache.clear()

    def _search(self, embeddings: np.ndarray) -> tuple[list[list[int]], list[list[float]]]:
        """Index search, with the reference's optional result cache keyed by sha256 of the embedding
        bytes (api.py:376-425): hits skip the index, misses are searched as one smaller batch."""
        index = self.registry.get("faiss_store")
        k = self.settings.retrieval_k
        if getattr(self.settings, "disable_cache_for_profiling", True):
            with stage_timers.track(STAGE_FAISS_SEARCH):
                distances, indices = index.search(embeddings, k)
            return [row.tolist() for row in indices], [row.tolist() for row in distances]
        n = embeddings.shape[0]
        ids_out: list[list[int]] = [[] for _ in range(n)]
        dist_out: list[list[float]] = [[] for _ in range(n)]
        keys = [hashlib.sha256(embeddings[i].tobytes()).hexdigest() for i in range(n)]
        missing = []
        with self._lock:
            for i, key in enumerate(keys):
                hit = self.cache.get(key)
                if hit:
                    ids_out[i], dist_out[i] = hit
                else:
                    missing.append(i)
        if missing:
            with stage_timers.track(STAGE_FAISS_SEARCH):
                distances, indices = index.search(np.array([embeddings[i] for i in missing]), k)
            with self._lock:
                for row, i in enumerate(missing):
                    ids_out[i], dist_out[i] = indices[row].tolist(), distances[row].tolist()
                    self.cache.put(keys[i], (ids_out[i], dist_out[i]))
        return ids_out, dist_out

    def _fetch_documents(self, doc_ids_batch: list[list[int]]) -> list[list[StoreDocument]]:
        store = self.registry.get("document_store")
        mode = getattr(self.settings, "documents_payload_mode", "full")
        if not store or mode == "id_only":
            return [[StoreDocument(doc_id=i, title="", content="") for i in ids] for ids in doc_ids_batch]
        with stage_timers.track(STAGE_DOCUMENT_FETCH):
            return store.fetch_documents_batch(doc_ids_batch, truncate_length=self.settings.truncate_length)

    @staticmethod
    def _to_retrieval_docs(docs: list[StoreDocument], scores: list[float]) -> list[RetrievalDocument]:
        return [_new_retrieval_doc(doc_id=d.doc_id, title=d.title, content=d.content, category=d.category or "",
                                   score=float(s))
                for d, s in zip(docs, scores)]  # no strict length check, as the reference

    def _freeze_gc_once(self) -> None:
        """After the first served batch: everything alive now (modules, models, tokenizer tables, the components) is
        long-lived — take it out of the cyclic collector's sight so that a later full collection costs what the
        per-batch garbage costs, not 40-60 ms (settings.gc_freeze_after_warmup; no counterpart in the reference, whose
        batches take seconds on the CPU)."""
        with self._lock:
            if self._gc_frozen:
                return
            self._gc_frozen = True
        if getattr(self.settings, "gc_freeze_after_warmup", True):
            gc.collect()
            gc.freeze()

    def _process_batch_sync(self, batch: Batch[RetrievalResponseItem]) -> list[RetrievalResponseItem]:
        out = self._process_batch_now(batch)
        if not self._gc_frozen:
            self._freeze_gc_once()
        return out

    def _process_batch_now(self, batch: Batch[RetrievalResponseItem]) -> list[RetrievalResponseItem]:
        if not self.registry.get("faiss_store"):
            raise RuntimeError("FAISS store not available")
        reranker = self.registry.get("reranker")
        embeddings = self._get_embeddings(batch)
        doc_ids_batch, distances_batch = self._search(embeddings)
        documents_batch = self._fetch_documents(doc_ids_batch)
        if reranker:
            if not self._shard_link_checked:
                self._attach_shard_link(reranker)
            # The reference wraps every fetched row three times on this path (RetrievalDocument -> Document ->
            # RerankedDocument -> RetrievalDocument, api.py:475-514); at top-100 that is 12 800 pydantic
            # objects per batch, more host time than the cross-encoder takes on the GPU.  A reranker that
            # declares `accepts_rows` (this build's does) only reads doc_id / title / content / category and
            # takes the fetched rows as they are, cut to the number of scores as the zip with the scores
            # did; any other reranker gets the reference's Document objects.
            if getattr(reranker, "accepts_rows", False):
                inputs = [docs[:len(scores)] for docs, scores in zip(documents_batch, distances_batch)]
            else:
                inputs = [[Document.model_construct(doc_id=d.doc_id, title=d.title, content=d.content,
                                                    category=d.category or "") for d, _ in zip(docs, scores)]
                          for docs, scores in zip(documents_batch, distances_batch)]
            queries = [req.query for req in batch.requests]
            if getattr(reranker, "accepts_rows", False):
                # this build's reranker builds the response documents itself, once (the reference's
                # RerankedDocument -> RetrievalDocument re-wrap is 3200 more objects per top-100 batch)
                per_request = reranker.rerank_batch(queries, inputs, result_factory=_new_retrieval_doc)
            else:
                reranked = reranker.rerank_batch(queries, inputs)
                per_request = [[_new_retrieval_doc(doc_id=d.doc_id, title=d.title, content=d.content,
                                                   category=d.category or "", score=d.score) for d in docs]
                               for docs in reranked]
        else:
            per_request = [self._to_retrieval_docs(docs, scores)
                           for docs, scores in zip(documents_batch, distances_batch)]
        if getattr(self.settings, "documents_payload_mode", "full") == "compressed":
            from . import lz4frame  # native block compressor behind the standard frame format

            return [RetrievalResponseItem.model_construct(
                        request_id=req.request_id, docs=[],
                        compressed_docs=lz4frame.compress(json.dumps(
                            [{"doc_id": d.doc_id, "title": d.title, "content": d.content, "category": d.category,
                              "score": d.score} for d in docs],
                            separators=(",", ":"), ensure_ascii=False).encode("utf-8")))
                    for req, docs in zip(batch.requests, per_request)]
        return [RetrievalResponseItem(request_id=req.request_id, docs=docs, compressed_docs=None)
                for req, docs in zip(batch.requests, per_request)]
