"""BertModel — Python handle on the C-ABI transformer (rag_bert_* in include/rag_amd.h).

Holds the weights as fp32 PyTorch-ROCm tensors (PyTorch's only job here) and hands their device
pointers to the HIP kernels; tokenised, PACKED sequences go in, pooled embeddings or classifier
scores come out.  Used by components/embedding.py (query encoder; reference
src/pipeline/components/embedding.py:127-133) and components/reranker.py (cross-encoder; reference
src/pipeline/components/reranker.py:248-252).

Weight names are a small canonical vocabulary (`CANONICAL_LAYER_KEYS` below); loaders translate
Hugging Face BERT / RoBERTa / XLM-RoBERTa state dicts into it.
"""

from __future__ import annotations

import ctypes as C
import threading
import itertools
import json
import os
from dataclasses import dataclass, field
from typing import Any, Iterable, Sequence

import numpy as np

from . import _native

EMB_KEYS = ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")
CANONICAL_LAYER_KEYS = ("qkv_w", "qkv_b", "attn_out_w", "attn_out_b", "ln1_g", "ln1_b",
                        "ffn_in_w", "ffn_in_b", "ffn_out_w", "ffn_out_b", "ln2_g", "ln2_b")
HEAD_KEYS = ("head_dense_w", "head_dense_b", "head_out_w", "head_out_b")

_GEMM_MODES = {"f32": 0, "f16": 1, "f32_strict": 2}  # RAG_GEMM_F32 / _F16 / _F32_STRICT
_ACTS = {"gelu": _native.ACT_GELU, "gelu_new": _native.ACT_GELU_TANH, "gelu_pytorch_tanh": _native.ACT_GELU_TANH,
         "relu": _native.ACT_RELU}


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden: int = 384
    n_layers: int = 6
    n_heads: int = 12
    intermediate: int = 1536
    max_positions: int = 512
    type_vocab: int = 2
    pos_offset: int = 0            # padding_idx + 1 for RoBERTa-family models
    act: str = "gelu"
    head: str = "none"             # "none" | "bert" | "roberta"
    n_labels: int = 1
    ln_eps: float = 1e-12
    pooling: str = "mean"          # sentence-embedding pooling: "mean" | "cls"
    # how big-batch GEMMs run (include/rag_amd.h RAG_GEMM_*): "f32" fp32 results on the bf16 matrix cores
    # (exact three-way bf16 split, six products), "f16" fp16 inputs (the reference's GPU reranker
    # precision), "f32_strict" everything on the fp32 MFMA
    gemm_dtype: str = "f32"
    extra: dict[str, Any] = field(default_factory=dict)

    # -- the model families BASELINE.json names (architecture facts are upstream model cards) ------
    @classmethod
    def minilm_l6(cls) -> "BertConfig":          # sentence-transformers/all-MiniLM-L6-v2
        return cls(hidden=384, n_layers=6, n_heads=12, intermediate=1536, pooling="mean")

    @classmethod
    def bge_base(cls) -> "BertConfig":           # BAAI/bge-base-en-v1.5
        return cls(hidden=768, n_layers=12, n_heads=12, intermediate=3072, pooling="cls")

    @classmethod
    def ms_marco_minilm_l6(cls) -> "BertConfig":  # cross-encoder/ms-marco-MiniLM-L-6-v2
        return cls(hidden=384, n_layers=6, n_heads=12, intermediate=1536, head="bert", n_labels=1)

    @classmethod
    def bge_reranker_base(cls) -> "BertConfig":   # BAAI/bge-reranker-base (XLM-RoBERTa base)
        return cls(vocab_size=250002, hidden=768, n_layers=12, n_heads=12, intermediate=3072,
                   max_positions=514, type_vocab=1, pos_offset=2, head="roberta", n_labels=1, ln_eps=1e-5)

    @classmethod
    def from_hf(cls, hf: dict[str, Any], head: str | None = None, pooling: str = "mean") -> "BertConfig":
        mt = hf.get("model_type", "bert")
        roberta = mt in ("roberta", "xlm-roberta", "camembert")
        if mt not in ("bert", "roberta", "xlm-roberta", "camembert"):
            raise ValueError(f"unsupported model_type {mt!r} (BERT / RoBERTa families only)")
        if hf.get("position_embedding_type", "absolute") != "absolute":
            raise ValueError("only absolute position embeddings are supported")
        if head is None:
            archs = " ".join(hf.get("architectures") or [])
            head = ("roberta" if roberta else "bert") if "SequenceClassification" in archs else "none"
        n_labels = len(hf["id2label"]) if head != "none" and "id2label" in hf else int(hf.get("num_labels", 1))
        return cls(vocab_size=int(hf["vocab_size"]), hidden=int(hf["hidden_size"]),
                   n_layers=int(hf["num_hidden_layers"]), n_heads=int(hf["num_attention_heads"]),
                   intermediate=int(hf["intermediate_size"]), max_positions=int(hf["max_position_embeddings"]),
                   type_vocab=int(hf.get("type_vocab_size", 0)),
                   pos_offset=(int(hf.get("pad_token_id", 1)) + 1) if roberta else 0,
                   act=str(hf.get("hidden_act", "gelu")), head=head, n_labels=max(1, n_labels),
                   ln_eps=float(hf.get("layer_norm_eps", 1e-12)), pooling=pooling)

    def to_struct(self) -> _native.BertConfigStruct:
        if self.act not in _ACTS:
            raise ValueError(f"unsupported activation {self.act!r}")
        if self.gemm_dtype not in _GEMM_MODES:
            raise ValueError(f"gemm_dtype must be one of {sorted(_GEMM_MODES)}, got {self.gemm_dtype!r}")
        head = {"none": _native.HEAD_NONE, "bert": _native.HEAD_BERT, "roberta": _native.HEAD_ROBERTA}[self.head]
        return _native.BertConfigStruct(self.vocab_size, self.hidden, self.n_layers, self.n_heads, self.intermediate,
                                        self.max_positions, self.type_vocab, self.pos_offset, _ACTS[self.act], head,
                                        self.n_labels, self.ln_eps, _GEMM_MODES[self.gemm_dtype])

    def weight_shapes(self) -> dict[str, tuple[int, ...]]:
        H, I = self.hidden, self.intermediate
        shapes: dict[str, tuple[int, ...]] = {
            "word_emb": (self.vocab_size, H), "pos_emb": (self.max_positions, H),
            "emb_ln_g": (H,), "emb_ln_b": (H,)}
        if self.type_vocab > 0:
            shapes["type_emb"] = (self.type_vocab, H)
        per = {"qkv_w": (3 * H, H), "qkv_b": (3 * H,), "attn_out_w": (H, H), "attn_out_b": (H,),
               "ln1_g": (H,), "ln1_b": (H,), "ffn_in_w": (I, H), "ffn_in_b": (I,),
               "ffn_out_w": (H, I), "ffn_out_b": (H,), "ln2_g": (H,), "ln2_b": (H,)}
        for l in range(self.n_layers):
            for k, s in per.items():
                shapes[f"layer{l}.{k}"] = s
        if self.head != "none":
            shapes.update({"head_dense_w": (H, H), "head_dense_b": (H,),
                           "head_out_w": (self.n_labels, H), "head_out_b": (self.n_labels,)})
        return shapes


def random_weights(cfg: BertConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Seeded synthetic weights of the architecture (there are no checkpoints offline).  Matrices
    N(0, 0.05), biases N(0, 0.02), LayerNorm gains around 1 — scaled so activations and attention
    logits are O(1), which exercises softmax / GELU / LayerNorm away from their trivial regimes."""
    rng = np.random.default_rng(seed)
    out: dict[str, np.ndarray] = {}
    for name, shape in cfg.weight_shapes().items():
        base = name.split(".")[-1]
        if base.endswith("_g"):
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif base.endswith("_b"):
            w = 0.02 * rng.standard_normal(shape)
        elif base.endswith("_emb"):
            w = 0.5 * rng.standard_normal(shape)
        else:
            w = 0.05 * rng.standard_normal(shape)
        out[name] = w.astype(np.float32)
    return out


def weights_from_hf_state_dict(cfg: BertConfig, sd: dict[str, Any]) -> dict[str, np.ndarray]:
    """Translate a Hugging Face BERT / (XLM-)RoBERTa state dict (tensors or arrays) to canonical names."""
    def arr(key: str) -> np.ndarray:
        v = sd[key]
        if hasattr(v, "detach"):
            v = v.detach().to("cpu").float().numpy()
        return np.ascontiguousarray(v, dtype=np.float32)

    prefix = ""
    for cand in ("bert.", "roberta.", "0.auto_model.", "model.", ""):
        if f"{cand}embeddings.word_embeddings.weight" in sd:
            prefix = cand
            break
    else:
        raise KeyError("no embeddings.word_embeddings.weight in the state dict")
    e = prefix + "embeddings."
    out = {"word_emb": arr(e + "word_embeddings.weight"), "pos_emb": arr(e + "position_embeddings.weight"),
           "emb_ln_g": arr(e + "LayerNorm.weight"), "emb_ln_b": arr(e + "LayerNorm.bias")}
    if cfg.type_vocab > 0:
        out["type_emb"] = arr(e + "token_type_embeddings.weight")
    for l in range(cfg.n_layers):
        p = f"{prefix}encoder.layer.{l}."
        out[f"layer{l}.qkv_w"] = np.concatenate([arr(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")])
        out[f"layer{l}.qkv_b"] = np.concatenate([arr(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")])
        out[f"layer{l}.attn_out_w"] = arr(p + "attention.output.dense.weight")
        out[f"layer{l}.attn_out_b"] = arr(p + "attention.output.dense.bias")
        out[f"layer{l}.ln1_g"] = arr(p + "attention.output.LayerNorm.weight")
        out[f"layer{l}.ln1_b"] = arr(p + "attention.output.LayerNorm.bias")
        out[f"layer{l}.ffn_in_w"] = arr(p + "intermediate.dense.weight")
        out[f"layer{l}.ffn_in_b"] = arr(p + "intermediate.dense.bias")
        out[f"layer{l}.ffn_out_w"] = arr(p + "output.dense.weight")
        out[f"layer{l}.ffn_out_b"] = arr(p + "output.dense.bias")
        out[f"layer{l}.ln2_g"] = arr(p + "output.LayerNorm.weight")
        out[f"layer{l}.ln2_b"] = arr(p + "output.LayerNorm.bias")
    if cfg.head == "bert":
        out["head_dense_w"], out["head_dense_b"] = arr(prefix + "pooler.dense.weight"), arr(prefix + "pooler.dense.bias")
        out["head_out_w"], out["head_out_b"] = arr("classifier.weight"), arr("classifier.bias")
    elif cfg.head == "roberta":
        out["head_dense_w"], out["head_dense_b"] = arr("classifier.dense.weight"), arr("classifier.dense.bias")
        out["head_out_w"], out["head_out_b"] = arr("classifier.out_proj.weight"), arr("classifier.out_proj.bias")
    return out


def load_pretrained_dir(path: str, head: str | None = None, pooling: str | None = None
                        ) -> tuple[BertConfig, dict[str, np.ndarray]]:
    """config.json + model.safetensors / pytorch_model.bin from a LOCAL directory (no network).
    A sentence-transformers layout (modules.json + 1_Pooling/config.json) sets the pooling mode."""
    with open(os.path.join(path, "config.json")) as fh:
        hf = json.load(fh)
    if pooling is None:
        pooling = "mean"
        pcfg = os.path.join(path, "1_Pooling", "config.json")
        if os.path.exists(pcfg):
            with open(pcfg) as fh:
                pj = json.load(fh)
            pooling = "cls" if pj.get("pooling_mode_cls_token") else "mean"
    cfg = BertConfig.from_hf(hf, head=head, pooling=pooling)
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file

        sd: dict[str, Any] = load_file(st)
    else:
        import torch

        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return cfg, weights_from_hf_state_dict(cfg, sd)


def pack_sequences(seqs: Sequence[Sequence[int]], type_seqs: Sequence[Sequence[int]] | None = None
                   ) -> tuple[np.ndarray, np.ndarray | None, np.ndarray]:
    """Concatenate token-id lists into the packed (ids, type_ids, cu_seqlens) the kernels take."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    cu = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    total = int(cu[-1])
    ids = np.fromiter(itertools.chain.from_iterable(seqs), dtype=np.int32, count=total)
    types = None
    if type_seqs is not None:
        types = np.fromiter(itertools.chain.from_iterable(type_seqs), dtype=np.int32, count=total)
    return ids, types, cu


class BertModel:
    """Weights on the GPU + the native forward pass."""

    def __init__(self, cfg: BertConfig, weights: dict[str, np.ndarray], device: int = 0) -> None:
        import torch

        self.cfg = cfg
        self.device = int(device)
        self._lib = _native.lib()
        self._flag_lock = threading.Lock()
        shapes = cfg.weight_shapes()
        missing = [k for k in shapes if k not in weights]
        if missing:
            raise KeyError(f"missing weights: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        if _native.device_count() <= self.device:
            raise _native.RagAmdError(_native.RAG_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)")
        dev = torch.device("cuda", self.device)
        self._tensors: dict[str, Any] = {}
        for name, shape in shapes.items():
            w = np.ascontiguousarray(weights[name], dtype=np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(w.shape)}")
            self._tensors[name] = torch.from_numpy(w).to(dev).contiguous()
        order = list(EMB_KEYS)
        for l in range(cfg.n_layers):
            order += [f"layer{l}.{k}" for k in CANONICAL_LAYER_KEYS]
        if cfg.head != "none":
            order += list(HEAD_KEYS)
        ptrs = [self._tensors[k].data_ptr() if k in self._tensors else 0 for k in order]
        if cfg.gemm_dtype not in _GEMM_MODES:
            raise ValueError(f"gemm_dtype must be one of {sorted(_GEMM_MODES)}, got {cfg.gemm_dtype!r}")
        table = (C.c_void_p * len(ptrs))(*ptrs)
        struct = cfg.to_struct()
        assert self._lib.rag_bert_weight_count(C.byref(struct)) == len(ptrs)
        self._h = C.c_void_p()
        torch.cuda.synchronize(dev)  # weights resident before the first forward on another stream
        _native.check(self._lib.rag_bert_create(C.byref(struct), table, len(ptrs), self.device, C.byref(self._h)))

    @classmethod
    def random_init(cls, cfg: BertConfig, seed: int = 0, device: int = 0) -> "BertModel":
        return cls(cfg, random_weights(cfg, seed), device)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rag_bert_destroy(self._h)
            self._h = C.c_void_p()
        self._tensors = {}

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def _forward(self, ids: np.ndarray, types: np.ndarray | None, cu: np.ndarray, out_kind: int, normalize: bool,
                 out_shape: tuple[int, ...]) -> np.ndarray:
        if not self._h:
            raise RuntimeError("BertModel is closed")
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        cu = np.ascontiguousarray(cu, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        tptr = None
        if types is not None:
            types = np.ascontiguousarray(types, dtype=np.int32)
            tptr = types.ctypes.data_as(i32p)
        out = np.empty(out_shape, dtype=np.float32)
        _native.check(self._lib.rag_bert_forward(self._h, ids.ctypes.data_as(i32p), tptr, cu.ctypes.data_as(i32p),
                                                 len(cu) - 1, out_kind, 1 if normalize else 0,
                                                 out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def forward_device(self, ids_ptr: int, types_ptr: int, cu_ptr: int, nseq: int, total_tokens: int,
                       max_seq_len: int, out_kind: int, normalize: bool, out_ptr: int, stream: int = 0,
                       range_flag_ptr: int = 0) -> None:
        """Asynchronous forward on device buffers (int32 ids / types / cu_seqlens, fp32 out); pointers as
        ints, `stream` a hipStream_t.  types_ptr may be 0.  range_flag_ptr: a zeroed pinned-host word that reads 1
        afterwards if an activation left fp16's range (the pass is then void; 0 = the handle's own word, see
        range_events)."""
        if not self._h:
            raise RuntimeError("BertModel is closed")
        _native.check(self._lib.rag_bert_forward_device(
            self._h, C.c_void_p(ids_ptr), C.c_void_p(types_ptr or None), C.c_void_p(cu_ptr), int(nseq),
            int(total_tokens), int(max_seq_len), int(out_kind), 1 if normalize else 0, C.c_void_p(out_ptr),
            C.c_void_p(range_flag_ptr or None), C.c_void_p(stream)))

    def _flag_word(self):
        """(pinned int32 view of one word, its address): a fresh zeroed word of a small ring for one asynchronous pass."""
        import torch

        with self._flag_lock:   # (batches run on pool threads: two passes must never share a word)
            if getattr(self, "_flag_ring", None) is None:
                self._flag_ring = torch.zeros(256, dtype=torch.int32).pin_memory()
                self._flag_next = 0
            i = self._flag_next
            self._flag_next = (i + 1) % 256
            word = self._flag_ring[i:i + 1]
            word.zero_()
            return word, self._flag_ring.data_ptr() + 4 * i

    def set_background(self, on: bool = True) -> None:
        """Small-batch GEMMs in their 32-KiB-LDS form, for running the encoder on a side stream under a long-running
        kernel such as the corpus scan (rag_bert_set_background)."""
        _native.check(self._lib.rag_bert_set_background(self._h, 1 if on else 0))

    def set_stream(self, stream: int = 0) -> None:
        """The stream embed() / embed_to_device() / classify() enqueue on instead of the model's own (0 restores it);
        the caller keeps it alive (rag_bert_set_stream)."""
        _native.check(self._lib.rag_bert_set_stream(self._h, C.c_void_p(int(stream)) if stream else None))

    def set_cu_budget(self, n_cus: int = 0) -> None:
        """How many compute units this model's launches may count on (0 = the whole device): for a caller that runs it
        on a stream restricted to a share of the chip (rag_stream_create_masked)."""
        _native.check(self._lib.rag_bert_set_cu_budget(self._h, int(n_cus)))

    def range_events(self, take_pending: bool = False) -> tuple[int, bool]:
        """(forward passes repeated on the three-plane bf16 path because an activation left fp16's range, whether an
        asynchronous pass has raised the flag since it was last taken) — rag_bert_range_events."""
        n, pend = C.c_int64(0), C.c_int32(0)
        _native.check(self._lib.rag_bert_range_events(self._h, C.byref(n), C.byref(pend) if take_pending else None))
        return int(n.value), bool(pend.value)

    def embed_to_device(self, seqs: Sequence[Sequence[int]], type_seqs: Sequence[Sequence[int]] | None = None,
                        normalize: bool = True, pooling: str | None = None):
        """embed() without the read-back: token ids go up, the forward pass is enqueued on the model's private
        stream and the call returns a DeviceEmbeddings handle at once (rag_bert_forward_to_device)."""
        import torch

        from .device_embeddings import DeviceEmbeddings

        if not self._h:
            raise RuntimeError("BertModel is closed")
        ids, types, cu = pack_sequences(seqs, type_seqs)
        kind = _native.BERT_OUT_CLS if (pooling or self.cfg.pooling) == "cls" else _native.BERT_OUT_MEAN
        i32p = C.POINTER(C.c_int32)
        out = torch.empty((len(seqs), self.cfg.hidden), dtype=torch.float32, device=torch.device("cuda", self.device))
        stream = C.c_void_p()
        word, word_ptr = self._flag_word()
        _native.check(self._lib.rag_bert_forward_to_device(
            self._h, ids.ctypes.data_as(i32p), types.ctypes.data_as(i32p) if types is not None else None,
            cu.ctypes.data_as(i32p), len(seqs), kind, 1 if normalize else 0, C.c_void_p(out.data_ptr()),
            C.c_void_p(word_ptr), C.byref(stream)))
        # (a closed model has waited for the device: nothing of its can still be writing to `out`)
        return DeviceEmbeddings(out, int(stream.value or 0), self.device, producer_alive=lambda: bool(self._h),
                                range_word=word)

    def embed(self, seqs: Sequence[Sequence[int]], type_seqs: Sequence[Sequence[int]] | None = None,
              normalize: bool = True, pooling: str | None = None) -> np.ndarray:
        """(n, hidden) sentence embeddings: pooled last hidden state, L2-normalised by default."""
        ids, types, cu = pack_sequences(seqs, type_seqs)
        kind = _native.BERT_OUT_CLS if (pooling or self.cfg.pooling) == "cls" else _native.BERT_OUT_MEAN
        return self._forward(ids, types, cu, kind, normalize, (len(seqs), self.cfg.hidden))

    def classify(self, seqs: Sequence[Sequence[int]], type_seqs: Sequence[Sequence[int]] | None = None,
                 sigmoid: bool = True) -> np.ndarray:
        """(n, n_labels) classifier outputs: sigmoid(logits) by default, as the reference reranker."""
        ids, types, cu = pack_sequences(seqs, type_seqs)
        kind = _native.BERT_OUT_PROBS if sigmoid else _native.BERT_OUT_LOGITS
        return self._forward(ids, types, cu, kind, False, (len(seqs), self.cfg.n_labels))

    def classify_packed(self, ids: np.ndarray, types: np.ndarray | None, cu: np.ndarray, sigmoid: bool = True
                        ) -> np.ndarray:
        """classify() on sequences that are already packed (pack_sequences): the caller packed them on
        another thread while the previous pass was running."""
        kind = _native.BERT_OUT_PROBS if sigmoid else _native.BERT_OUT_LOGITS
        return self._forward(ids, types, cu, kind, False, (len(cu) - 1, self.cfg.n_labels))

    def hidden_states(self, seqs: Sequence[Sequence[int]], type_seqs: Sequence[Sequence[int]] | None = None
                      ) -> np.ndarray:
        ids, types, cu = pack_sequences(seqs, type_seqs)
        return self._forward(ids, types, cu, _native.BERT_OUT_HIDDEN, False, (int(cu[-1]), self.cfg.hidden))


def iter_token_budget(lengths: Iterable[int], max_tokens: int) -> Iterable[tuple[int, int]]:
    """Split a list of sequence lengths into consecutive [lo, hi) chunks of at most max_tokens tokens."""
    lo, acc, i = 0, 0, 0
    for i, n in enumerate(lengths):
        if acc and acc + n > max_tokens:
            yield lo, i
            lo, acc = i, 0
        acc += n
    if acc:
        yield lo, i + 1
