"""Reranker — drop-in for the reference's cross-encoder component, on the HIP transformer.

Same surface and error contract as the reference class (src/pipeline/components/reranker.py:37-308;
messages pinned by tests/test_components.py:387 and tests/test_generation_service.py:217):

    Reranker(settings).load() / .rerank(query, documents, top_n=None) -> list[RerankedDocument]
    .rerank_batch(queries, documents_batch, top_n=None) / .unload() / .is_loaded

`rerank` reproduces :237-272: pairs [query, doc.content] -> tokenizer(truncation, max_length =
settings.truncate_length) -> classifier logit -> sigmoid -> stable descending sort -> [:top_n].
Differences, all on the side of the hardware: the pairs of a whole BATCH of queries go through one
packed forward pass (the reference loops over queries, :303-306, and pads each to its longest pair);
arithmetic is fp32 (the reference uses fp16 on a GPU, :91-93).

Multi-GPU: with one process per GPU the batch is sharded BY QUERY — query i and its
documents go to rank i mod G, every rank scores its share with its own copy of the model, the scores
come back to rank 0.  No tensor collective is involved; the texts travel as one scatter and the scores
as one gather over the serving channel of the sharded index (`attach_shard_link`).
"""

from __future__ import annotations

import gc
import logging
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ..bert import iter_token_budget, pack_sequences
from ..config import PipelineSettings, resolve_gpu_device
from .schemas import Document, RerankedDocument, fast_constructor

logger = logging.getLogger(__name__)

_MAX_TOKENS_PER_PASS = 131072
# Chunks: the first chunk's tokenisation is the only one the GPU waits for, and the GPU is the less efficient the
# smaller a pass is (a 160-pair pass takes 2.0 ms, 640 pairs 1.5, 3200 pairs 5.8 in fp16 mode: the chip is not
# covered below ~50 k tokens).  Round 3's native pair encoder needs 0.7 us per pair, so a quarter of the batch
# (0.6 ms of tokenisation at 3200 pairs) goes first and the rest follows in one piece, tokenised under that pass.
_PAIRS_PER_CHUNK = 4096
_FIRST_FRACTION = 4
_MIN_FIRST_CHUNK = 256
_new_reranked = fast_constructor(RerankedDocument)


class Reranker:
    # rerank / rerank_batch read only doc_id, title, content and category of what they are given: callers
    # may pass any row objects with those attributes (retrieval_executor.py does, to skip re-wrapping)
    accepts_rows = True

    def __init__(self, settings: PipelineSettings) -> None:
        self.settings = settings
        self.model_name = settings.reranker_model_name
        self._device_index = resolve_gpu_device(settings)
        self.device = f"cuda:{self._device_index}"
        self.tokenizer = None
        self.model = None
        self._max_len = 512
        self._loaded = False
        self._link = None  # sharded.ShardedFlatIndex serving channel when the batch is split over ranks
        self._tok_pool: ThreadPoolExecutor | None = None
        logger.info("Reranker initialized (device: %s)", self.device)

    def load(self) -> None:
        if self._loaded:
            logger.warning("Reranker already loaded")
            return
        logger.info("Loading reranker model: %s", self.model_name)
        t0 = time.time()
        try:
            from ..bert import BertModel
            from ..model_source import resolve_model

            cfg, weights, tokenizer, max_len = resolve_model(self.model_name, "reranker")
            cfg.gemm_dtype = str(getattr(self.settings, "reranker_dtype", "f32")).lower().replace("fp", "f")
            self._device_index = resolve_gpu_device(self.settings)  # the group may have been set up since
            self.device = f"cuda:{self._device_index}"
            self.model = BertModel(cfg, weights, device=self._device_index)
            self.tokenizer, self._max_len = tokenizer, max_len
            self._loaded = True
            self._score_pairs(["query " * 10], ["document " * 100])  # warm-up (:156-166)
            logger.info("Reranker model loaded in %.2f seconds", time.time() - t0)
        except Exception:
            logger.exception("Failed to load reranker model")
            self.model = None
            self.tokenizer = None
            self._loaded = False
            raise

    def unload(self) -> None:
        if not self._loaded:
            return
        logger.info("Unloading reranker model")
        if self.model is not None:
            self.model.close()
        if self._tok_pool is not None:
            self._tok_pool.shutdown(wait=True)
            self._tok_pool = None
        self.model = None
        self.tokenizer = None
        gc.collect()
        self._loaded = False

    @property
    def is_loaded(self) -> bool:
        return self._loaded

    def _score_pairs(self, queries: list[str], docs: list[str]) -> list[float]:
        """Sigmoid scores of (query, document) pairs.  Pairs go through in chunks: while the GPU scores one
        chunk (the C call releases the GIL) a worker thread tokenises AND packs the next, so the calling
        thread does nothing between two GPU passes but hand over three arrays — host tokenisation and the
        cross-encoder pass overlap instead of adding up.  The first chunk is a quarter of the batch: its
        tokenisation is the only one the GPU has to wait for, and small passes use the GPU badly."""
        max_len = min(int(self.settings.truncate_length), self._max_len)
        with_types = self.model.cfg.type_vocab > 1
        n = len(docs)
        bounds, lo = [], 0
        while lo < n:
            first = max(_MIN_FIRST_CHUNK, n // _FIRST_FRACTION)
            hi = min(lo + (first if lo == 0 and n >= 2 * _MIN_FIRST_CHUNK else _PAIRS_PER_CHUNK), n)
            bounds.append((lo, hi))
            lo = hi

        def prepare(lo: int, hi: int):
            packed_fn = getattr(self.tokenizer, "encode_pairs_packed", None)
            packed = packed_fn(queries[lo:hi], docs[lo:hi], max_len, with_types) if packed_fn is not None else None
            if packed is not None:   # one native call: already the arrays the model takes
                ids, types, cu = packed
                lens = np.diff(cu)
                if int(cu[-1]) <= _MAX_TOKENS_PER_PASS:
                    return [(ids, types, cu)]
                return [(ids[cu[a]:cu[b]], types[cu[a]:cu[b]] if types is not None else None, cu[a:b + 1] - cu[a])
                        for a, b in iter_token_budget(lens.tolist(), _MAX_TOKENS_PER_PASS)]
            ids, types = self.tokenizer.encode_pairs(queries[lo:hi], docs[lo:hi], max_len)
            parts = []
            for a, b in iter_token_budget([len(s) for s in ids], _MAX_TOKENS_PER_PASS):
                parts.append(pack_sequences(ids[a:b], types[a:b] if with_types else None))
            return parts

        scores: list[float] = []
        if not bounds:
            return scores
        if self._tok_pool is None:
            self._tok_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rerank-tokenise")
        pending = self._tok_pool.submit(prepare, *bounds[0])
        for i in range(len(bounds)):
            parts = pending.result()
            if i + 1 < len(bounds):
                pending = self._tok_pool.submit(prepare, *bounds[i + 1])
            for ids, types, cu in parts:
                probs = self.model.classify_packed(ids, types, cu, sigmoid=True)
                scores.extend(probs[:, 0].tolist())
        return scores

    @staticmethod
    def _ranked(documents: list[Document], scores: list[float], top_n: int | None, make=_new_reranked) -> list:
        # fields come from validated Document objects (or fetched rows): no second validation pass, and the
        # result objects are built once, by `make` — RerankedDocument unless the caller asked for its own
        # result class (retrieval_executor.py does: it would only re-wrap every document otherwise)
        order = sorted(range(len(documents)), key=scores.__getitem__, reverse=True)  # stable: ties keep retrieval order
        if top_n is not None:
            order = order[:top_n]  # objects only for what is returned
        out = []
        for i in order:
            d = documents[i]
            out.append(make(doc_id=d.doc_id, title=d.title, content=d.content, category=d.category or "",
                            score=float(scores[i])))
        return out

    def rerank(self, query: str, documents: list[Document], top_n: int | None = None) -> list[RerankedDocument]:
        if not self._loaded or self.model is None or self.tokenizer is None:
            raise RuntimeError("Reranker model not loaded")
        if not documents:
            return []
        scores = self._score_pairs([query] * len(documents), [d.content for d in documents])
        return self._ranked(documents, scores, top_n)

    # -- query-sharded scoring over the ranks of a process group ---------------------------------------
    def attach_shard_link(self, link) -> None:
        """Use `link` (the ShardedFlatIndex of this rank's FAISSStore: `store.shard_link`) to split
        rerank batches over the group.  Rank 0 calls rerank_batch as before; the other ranks register
        their side of the exchange and serve it from the store's follower loop."""
        from ..sharded import OP_RERANK

        if link is not None and self._loaded and getattr(link.device, "type", "cpu") == "cuda" \
                and int(link.device.index or 0) != self._device_index:
            raise RuntimeError(f"reranker is on cuda:{self._device_index} but this rank's index shard is on "
                               f"{link.device}: both follow config.resolve_gpu_device()")
        self._link = link
        if link is not None and link.rank != 0:
            link.register_handler(OP_RERANK, self._follower_pass)

    def _score_work(self, work: list[tuple[int, str, list[str]]]) -> list[tuple[int, list[float]]]:
        flat_q = [q for _, q, docs in work for _ in docs]
        flat_d = [d for _, _, docs in work for d in docs]
        scores = self._score_pairs(flat_q, flat_d) if flat_d else []
        out, pos = [], 0
        for qi, _, docs in work:
            out.append((qi, scores[pos:pos + len(docs)]))
            pos += len(docs)
        return out

    def _follower_pass(self) -> None:
        dist, group = self._link._dist, self._link.group
        recv: list = [None]
        dist.scatter_object_list(recv, None, src=0, group=group)
        dist.gather_object(self._score_work(recv[0]), None, dst=0, group=group)

    def _sharded_scores(self, queries: list[str], documents_batch: list[list[Document]]) -> list[list[float]]:
        from ..sharded import OP_RERANK

        link = self._link
        dist, group, world = link._dist, link.group, link.world
        parts = [[(qi, queries[qi], [d.content for d in documents_batch[qi]]) for qi in range(r, len(queries), world)]
                 for r in range(world)]
        gathered: list = [None] * world
        with link.exclusive():  # head + scatter + gather are one request: batches run on several threads
            link.leader_call(OP_RERANK)
            recv: list = [None]
            dist.scatter_object_list(recv, parts, src=0, group=group)
            dist.gather_object(self._score_work(recv[0]), gathered, dst=0, group=group)
        scores: list[list[float]] = [[] for _ in queries]
        for part in gathered:
            for qi, sc in part:
                scores[qi] = sc
        return scores

    def rerank_batch(self, queries: list[str], documents_batch: list[list[Document]],
                     top_n: int | None = None, *, result_factory=None) -> list[list[RerankedDocument]]:
        # `result_factory` (keyword-only, not in the reference's signature): build the result objects with this
        # constructor instead of RerankedDocument — same five fields, passed as keywords
        if not self._loaded or self.model is None or self.tokenizer is None:
            raise RuntimeError("Reranker model not loaded")
        if len(queries) != len(documents_batch):
            raise ValueError(
                f"Queries ({len(queries)}) and documents ({len(documents_batch)}) must have same length")
        if self._link is not None and self._link.world > 1 and self._link.rank == 0 and len(queries) > 1:
            per_query = self._sharded_scores(queries, documents_batch)
            make = result_factory or _new_reranked
            return [self._ranked(docs, sc, top_n, make) if docs else [] for docs, sc in zip(documents_batch, per_query)]
        flat_q = [q for q, docs in zip(queries, documents_batch) for _ in docs]
        flat_d = [d.content for docs in documents_batch for d in docs]
        make = result_factory or _new_reranked
        out, pos = [], 0
        # (Building the result objects on the tokeniser's worker thread while the GPU scores was tried in round 4 and taken
        # out again: the worker holds the GIL when the pass returns and the calling thread waits out its switch interval —
        # config C went 18.8 -> 19.6 ms per batch.)
        scores = self._score_pairs(flat_q, flat_d) if flat_d else []
        for docs in documents_batch:
            out.append(self._ranked(docs, scores[pos:pos + len(docs)], top_n, make) if docs else [])
            pos += len(docs)
        return out
