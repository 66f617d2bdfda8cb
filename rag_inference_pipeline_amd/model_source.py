"""Resolve a model *name* from the settings into (config, weights, tokenizer) without a network.

The reference hands hub names to SentenceTransformer / AutoTokenizer / AutoModel...from_pretrained
(embedding.py:80, reranker.py:84-111; defaults BAAI/bge-base-en-v1.5 and BAAI/bge-reranker-base,
config/__init__.py:317-325), which fetch on first use.  Here nothing is ever fetched:

  1. a local directory (config.json + model.safetensors|pytorch_model.bin + tokenizer files);
  2. a hub name already present in the local Hugging Face cache (snapshot lookup, local_files_only);
  3. `synthetic:<architecture>[:seed]` — seeded random weights of a named architecture with a
     deterministic hashing tokenizer.  This is what benchmarks and tests use: no checkpoint or
     vocabulary exists in the build environment, and throughput does not depend on weight values.
"""

from __future__ import annotations

import os
import re
import zlib
from typing import Any, Sequence

import numpy as np

from .bert import BertConfig, load_pretrained_dir, random_weights

_PRESETS = {
    "all-minilm-l6-v2": BertConfig.minilm_l6,
    "sentence-transformers/all-minilm-l6-v2": BertConfig.minilm_l6,
    "bge-base-en-v1.5": BertConfig.bge_base,
    "baai/bge-base-en-v1.5": BertConfig.bge_base,
    "ms-marco-minilm-l-6-v2": BertConfig.ms_marco_minilm_l6,
    "cross-encoder/ms-marco-minilm-l-6-v2": BertConfig.ms_marco_minilm_l6,
    "bge-reranker-base": BertConfig.bge_reranker_base,
    "baai/bge-reranker-base": BertConfig.bge_reranker_base,
}


class _PieceMemo(dict):
    """piece -> id, computed on first sight (crc32 into the id range above the special tokens)."""

    def __init__(self, first: int, span: int) -> None:
        super().__init__()
        self._first, self._span = first, span

    def __missing__(self, piece: str) -> int:
        pid = self._first + zlib.crc32(piece.encode("utf-8")) % self._span
        if len(self) < 200_000:
            self[piece] = pid
        return pid


class HashTokenizer:
    """Deterministic stand-in vocabulary for synthetic models: lower-cased word / punctuation
    pieces hashed (crc32) into the id range above the special tokens.  BERT-style framing
    [CLS] a [SEP] (b [SEP]) with token types 0/1, or RoBERTa-style <s> a </s></s> b </s>."""

    _piece = re.compile(r"\w+|[^\w\s]", re.UNICODE)

    def __init__(self, vocab_size: int, roberta: bool = False) -> None:
        self.vocab_size = vocab_size
        self.roberta = roberta
        self.cls_id, self.sep_id, self.pad_id = (0, 2, 1) if roberta else (101, 102, 0)
        self._first = 1000 if vocab_size > 2000 else 8
        self._memo = _PieceMemo(self._first, vocab_size - self._first)

    def _piece_id(self, piece: str) -> int:
        return self._memo[piece]

    def _ids(self, text: str) -> list[int]:
        # dict.__getitem__ with __missing__: no Python-level call for pieces seen before (a rerank batch
        # looks up ~80k pieces)
        return list(map(self._memo.__getitem__, self._piece.findall(text.lower())))

    def encode_batch(self, texts: Sequence[str], max_length: int) -> tuple[list[list[int]], list[list[int]]]:
        ids = [[self.cls_id] + self._ids(t)[: max(0, max_length - 2)] + [self.sep_id] for t in texts]
        return ids, [[0] * len(s) for s in ids]

    def encode_pairs(self, first: Sequence[str], second: Sequence[str], max_length: int
                     ) -> tuple[list[list[int]], list[list[int]]]:
        out_ids, out_types = [], []
        n_special = 4 if self.roberta else 3
        memo: dict[str, list[int]] = {}  # a rerank batch repeats each query once per document
        for a, b in zip(first, second):
            ia = memo.get(a)
            if ia is None:
                ia = memo[a] = self._ids(a)
            ia, ib = list(ia), self._ids(b)
            while len(ia) + len(ib) + n_special > max_length and (ia or ib):  # longest_first truncation
                if len(ib) >= len(ia):
                    ib.pop()
                else:
                    ia.pop()
            if self.roberta:
                ids = [self.cls_id] + ia + [self.sep_id, self.sep_id] + ib + [self.sep_id]
                types = [0] * len(ids)
            else:
                ids = [self.cls_id] + ia + [self.sep_id] + ib + [self.sep_id]
                types = [0] * (len(ia) + 2) + [1] * (len(ib) + 1)
            out_ids.append(ids)
            out_types.append(types)
        return out_ids, out_types


    def encode_pairs_packed(self, first: Sequence[str], second: Sequence[str], max_length: int, with_types: bool = True):
        """encode_pairs() straight into the packed (ids, token types, cu_seqlens) arrays the transformer takes, in one
        native call (rag_hash_encode_pairs).  None when a text is not ASCII or the library is not there: the caller
        then uses encode_pairs()."""
        import ctypes as C

        try:
            from . import _native
            lib = _native.lib()
        except Exception:  # noqa: BLE001
            return None
        n = len(first)
        try:
            fa = [t.encode("ascii") for t in first]
            fb = [t.encode("ascii") for t in second]
        except UnicodeEncodeError:
            return None
        la = np.fromiter(map(len, fa), dtype=np.int64, count=n)
        lb = np.fromiter(map(len, fb), dtype=np.int64, count=n)
        cap = int(la.sum() + lb.sum()) + 4 * n + 4
        ids = np.empty(cap, dtype=np.int32)
        types = np.empty(cap, dtype=np.int32) if with_types else None
        cu = np.empty(n + 1, dtype=np.int32)
        i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        total = lib.rag_hash_encode_pairs(
            (C.c_char_p * n)(*fa), la.ctypes.data_as(i64p), (C.c_char_p * n)(*fb), lb.ctypes.data_as(i64p), n,
            int(max_length), 1 if self.roberta else 0, int(self._first), int(self.vocab_size - self._first),
            int(self.cls_id), int(self.sep_id), ids.ctypes.data_as(i32p),
            types.ctypes.data_as(i32p) if types is not None else None, cu.ctypes.data_as(i32p), cap)
        if total < 0:
            return None
        return ids[:total], (types[:total] if types is not None else None), cu


class HFTokenizer:
    """A Hugging Face tokenizer loaded from local files (tokenizer.json / vocab.txt / sentencepiece)."""

    def __init__(self, path: str) -> None:
        from transformers import AutoTokenizer

        self._tok = AutoTokenizer.from_pretrained(path, local_files_only=True)

    @staticmethod
    def _types(enc: Any, ids: list[list[int]]) -> list[list[int]]:
        tt = enc.get("token_type_ids")
        return [list(t) for t in tt] if tt is not None else [[0] * len(s) for s in ids]

    def encode_batch(self, texts: Sequence[str], max_length: int) -> tuple[list[list[int]], list[list[int]]]:
        enc = self._tok(list(texts), truncation=True, max_length=max_length, padding=False)
        ids = [list(s) for s in enc["input_ids"]]
        return ids, self._types(enc, ids)

    def encode_pairs(self, first: Sequence[str], second: Sequence[str], max_length: int
                     ) -> tuple[list[list[int]], list[list[int]]]:
        # the reference call: tokenizer(pairs, padding=True, truncation=True, max_length=...) (reranker.py:240-246);
        # padding is dropped because sequences are packed
        enc = self._tok(list(first), list(second), truncation=True, max_length=max_length, padding=False)
        ids = [list(s) for s in enc["input_ids"]]
        return ids, self._types(enc, ids)


def _find_local(name: str) -> str | None:
    if os.path.isdir(name) and os.path.exists(os.path.join(name, "config.json")):
        return name
    try:  # a snapshot already in the Hugging Face cache; never touches the network
        from huggingface_hub import snapshot_download

        return snapshot_download(name, local_files_only=True)
    except Exception:
        return None


def _st_max_seq_length(path: str) -> int | None:
    import json

    p = os.path.join(path, "sentence_bert_config.json")
    if os.path.exists(p):
        with open(p) as fh:
            return int(json.load(fh).get("max_seq_length", 0)) or None
    return None


def resolve_model(name: str, role: str) -> tuple[BertConfig, dict[str, np.ndarray], Any, int]:
    """(config, weights, tokenizer, max_seq_length) for `role` in {"embedding", "reranker"}."""
    if name.startswith("synthetic:"):
        parts = name.split(":")
        arch = parts[1].lower()
        seed = int(parts[2]) if len(parts) > 2 else 0
        if arch not in _PRESETS:
            raise ValueError(f"unknown synthetic architecture {parts[1]!r}; known: {sorted(set(_PRESETS))}")
        cfg = _PRESETS[arch]()
        if role == "reranker" and cfg.head == "none":
            raise ValueError(f"{parts[1]} has no classifier head; pick a cross-encoder architecture")
        tok = HashTokenizer(cfg.vocab_size, roberta=cfg.pos_offset > 0)
        return cfg, random_weights(cfg, seed), tok, min(512, cfg.max_positions - cfg.pos_offset)
    path = _find_local(name)
    if path is None:
        raise RuntimeError(
            f"model {name!r} is not a local directory and is not in the local Hugging Face cache; this build "
            "never downloads. Point the setting at a local checkpoint directory, pre-populate the cache, or use "
            "'synthetic:<architecture>' for seeded random weights")
    cfg, weights = load_pretrained_dir(path, head=None if role == "reranker" else "none")
    if role == "reranker" and cfg.head == "none":
        raise RuntimeError(f"{name!r} is not a sequence-classification checkpoint")
    max_len = _st_max_seq_length(path) if role == "embedding" else None
    return cfg, weights, HFTokenizer(path), min(max_len or 512, cfg.max_positions - cfg.pos_offset)
