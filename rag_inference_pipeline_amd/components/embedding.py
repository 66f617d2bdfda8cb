"""EmbeddingGenerator — drop-in for the reference's query-embedding component, on the HIP encoder.

Same surface and error contract as the reference class (src/pipeline/components/embedding.py:36-205;
messages pinned by tests/test_components.py:307, :318):

    EmbeddingGenerator(settings).load() / .encode(texts) -> (n, dim) float32, rows unit-norm
    .unload() / .is_loaded / .cache / .clear_cache()

`encode` reproduces `SentenceTransformer.encode(batch_size=32, convert_to_numpy=True,
normalize_embeddings=True)` (:127-133): tokenise -> BERT encoder -> pooling (mean or CLS, as the
checkpoint's sentence-transformers config says) -> L2 normalise.  The whole batch runs as ONE packed
forward pass (no padding tokens), on fp32 MFMA kernels.  The optional sha256-keyed text cache
(:135-175) is kept.  `settings.only_cpu` is ignored: this build has no CPU path.
"""

from __future__ import annotations

import gc
import hashlib
import logging
import threading

import numpy as np

from ..bert import iter_token_budget
from ..cache import LRUCache
from ..config import PipelineSettings, resolve_gpu_device

logger = logging.getLogger(__name__)

_MAX_TOKENS_PER_PASS = 65536  # bounds the activation workspace (~1.2 GB at hidden 768)


class EmbeddingGenerator:
    def __init__(self, settings: PipelineSettings) -> None:
        self.settings = settings
        self.model_name = settings.embedding_model_name
        self._device_index = resolve_gpu_device(settings)
        self.device = f"cuda:{self._device_index}"
        self._model = None
        self._tokenizer = None
        self._max_len = 512
        self._is_loaded = False
        self._lock = threading.Lock()
        self.cache: LRUCache[str, np.ndarray] = LRUCache(capacity=10000, ttl=settings.cache_max_ttl,
                                                         name="embedding_cache")

    def load(self) -> None:
        if self._is_loaded:
            logger.info("Embedding model already loaded")
            return
        logger.info("Loading embedding model: %s", self.model_name)
        try:
            from ..bert import BertModel
            from ..model_source import resolve_model

            cfg, weights, tokenizer, max_len = resolve_model(self.model_name, "embedding")
            self._device_index = resolve_gpu_device(self.settings)  # the group may have been set up since
            self.device = f"cuda:{self._device_index}"
            self._model = BertModel(cfg, weights, device=self._device_index)
            self._tokenizer, self._max_len = tokenizer, max_len
            self._share_stream = 0
            share = int(getattr(self.settings, "encoder_cus", 0) or 0)
            handoff = bool(getattr(self.settings, "disable_cache_for_profiling", True))
            if share > 0 and not handoff:
                # with the text cache on, batches go through encode() and the index searches on its own unmasked stream:
                # a share for the encoder would only slow it down (there is nothing beside it to make room for)
                logger.warning("RAG_AMD_ENCODER_CUS=%d ignored: the partition serves the device hand-off path, which is "
                               "active only with DISABLE_CACHE_FOR_PROFILING=true", share)
            if share > 0 and handoff and self._dist_world() == 1:
                # the encoder's share of the chip (settings.encoder_cus): its passes run on these CUs only, beside
                # whatever the index's search stream runs on the others (components/faiss_store.py)
                from ..flat_index import create_masked_stream
                self._share_stream = create_masked_stream(self._device_index, 0, share)
                self._model.set_stream(self._share_stream)
                self._model.set_cu_budget(share)
            self._is_loaded = True
            # warm-up on a longer text, as the reference does (:84-93)
            self._encode_uncached(["This is a test sentence for warmup. " * 10])
            logger.info("Embedding model loaded successfully on %s", self.device)
        except Exception:
            logger.exception("Failed to load embedding model")
            self._model = None
            self._is_loaded = False
            raise

    @staticmethod
    def _dist_world() -> int:
        try:
            import torch.distributed as dist
        except ImportError:
            return 1
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _encode_uncached(self, texts: list[str]) -> np.ndarray:
        ids, types = self._tokenizer.encode_batch(texts, self._max_len)
        out = np.empty((len(texts), self._model.cfg.hidden), dtype=np.float32)
        for lo, hi in iter_token_budget([len(s) for s in ids], _MAX_TOKENS_PER_PASS):
            out[lo:hi] = self._model.embed(ids[lo:hi], types[lo:hi], normalize=True)
        return out

    def encode_device(self, texts: list[str]):
        """encode() whose result stays in HBM: a DeviceEmbeddings handle (pointer + producer stream) that
        FAISSStore.search takes as it is, so a retrieval batch reads back its ids and scores and nothing else.
        Same error contract as encode(); batches beyond the per-pass token budget and the cached mode (keyed
        on host values) go through encode()."""
        if not self._is_loaded or self._model is None:
            raise RuntimeError("Model not loaded. Call load() first.")
        if not texts:
            raise ValueError("Cannot encode empty text list")
        if not getattr(self.settings, "disable_cache_for_profiling", True):
            return self.encode(texts)
        ids, types = self._tokenizer.encode_batch(list(texts), self._max_len)
        if sum(len(s) for s in ids) > _MAX_TOKENS_PER_PASS:
            return self.encode(texts)
        dev = self._model.embed_to_device(ids, types, normalize=True)
        texts = list(texts)
        dev.recompute = lambda: self._encode_uncached(texts)   # the host path: fp32's range whatever the values
        return dev

    def encode(self, texts: list[str]) -> np.ndarray:
        if not self._is_loaded or self._model is None:
            raise RuntimeError("Model not loaded. Call load() first.")
        if not texts:
            raise ValueError("Cannot encode empty text list")
        logger.debug("Encoding batch of %d texts", len(texts))
        if getattr(self.settings, "disable_cache_for_profiling", True):
            return self._encode_uncached(list(texts))
        results: list[np.ndarray | None] = [None] * len(texts)
        missing: list[int] = []
        with self._lock:
            for i, text in enumerate(texts):
                hit = self.cache.get(hashlib.sha256(text.strip().encode()).hexdigest())
                if hit is not None:
                    results[i] = hit
                else:
                    missing.append(i)
        if missing:
            fresh = self._encode_uncached([texts[i] for i in missing])
            with self._lock:
                for row, i in zip(fresh, missing):
                    results[i] = row
                    self.cache.put(hashlib.sha256(texts[i].strip().encode()).hexdigest(), row)
        return np.array(results)

    def clear_cache(self) -> None:
        with self._lock:
            self.cache.clear()

    def unload(self) -> None:
        if self._is_loaded:
            logger.info("Unloading embedding model")
            if self._model is not None:
                self._model.close()
            if getattr(self, "_share_stream", 0):
                from ..flat_index import destroy_stream
                destroy_stream(self._device_index, self._share_stream)
                self._share_stream = 0
            self._model = None
            self._tokenizer = None
            self._is_loaded = False
            gc.collect()

    @property
    def is_loaded(self) -> bool:
        return self._is_loaded

    def __repr__(self) -> str:
        status = "loaded" if self._is_loaded else "not loaded"
        return f"EmbeddingGenerator(model={self.model_name}, device={self.device}, status={status})"
