"""CPU restatement of the transformer side of the hot path.  TEST INFRASTRUCTURE ONLY — imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.

What it follows (reference call sites; the arithmetic itself lives in un-vendored wheels —
sentence-transformers 5.1.2 / transformers 4.57.3 / torch 2.9.1, uv.lock:4263-4264, 4512-4513,
4439-4440):
  * query encoder: SentenceTransformer.encode(..., normalize_embeddings=True)
    (src/pipeline/components/embedding.py:127-133) = BERT encoder -> Pooling (mean over tokens for
    all-MiniLM-L6-v2, first token for bge-base-en-v1.5) -> L2 normalise;
  * cross-encoder: AutoModelForSequenceClassification(**inputs).logits.view(-1).float() -> sigmoid
    (src/pipeline/components/reranker.py:248-252), then a stable descending sort and a top_n slice
    (:270-272).

Pinning: tests/test_bert_oracle.py checks this restatement against `transformers`' own
BertModel / BertForSequenceClassification / XLMRobertaForSequenceClassification, built offline from
a config with seeded random weights (no checkpoint exists in this environment).  That pins the
*architecture semantics*; it cannot pin trained-checkpoint outputs ("parity unpinned" in that sense,
see DESIGN.md).

Plain fp32 torch on the CPU, written for clarity: padded batch, additive attention mask, softmax.
"""

from __future__ import annotations

import math
from typing import Any, Sequence

import numpy as np
import torch
import torch.nn.functional as F


_DTYPE = [torch.float32]  # hidden_states(dtype=torch.float64) evaluates the same graph in double (accuracy arbiter)


def _t(w: dict[str, np.ndarray], key: str) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(w[key], dtype=np.float32)).to(_DTYPE[0])


def _act(x: torch.Tensor, name: str) -> torch.Tensor:
    if name == "gelu":
        return F.gelu(x)
    if name in ("gelu_new", "gelu_pytorch_tanh"):
        return F.gelu(x, approximate="tanh")
    if name == "relu":
        return F.relu(x)
    raise ValueError(name)


@torch.no_grad()
def hidden_states(cfg: Any, w: dict[str, np.ndarray], seqs: Sequence[Sequence[int]],
                  type_seqs: Sequence[Sequence[int]] | None = None, dtype: torch.dtype = torch.float32
                  ) -> list[np.ndarray]:
    """Last hidden state of every sequence ([len_s, hidden] each), HF BertModel semantics."""
    _DTYPE[0] = dtype
    try:
        return _hidden_states(cfg, w, seqs, type_seqs)
    finally:
        _DTYPE[0] = torch.float32


def _hidden_states(cfg: Any, w: dict[str, np.ndarray], seqs: Sequence[Sequence[int]],
                   type_seqs: Sequence[Sequence[int]] | None = None) -> list[np.ndarray]:
    n, L = len(seqs), max(len(s) for s in seqs)
    ids = torch.zeros((n, L), dtype=torch.long)
    types = torch.zeros((n, L), dtype=torch.long)
    mask = torch.zeros((n, L), dtype=_DTYPE[0])
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(list(s), dtype=torch.long)
        mask[i, : len(s)] = 1.0
        if type_seqs is not None:
            types[i, : len(s)] = torch.tensor(list(type_seqs[i]), dtype=torch.long)
    pos = torch.arange(L).unsqueeze(0).expand(n, L) + cfg.pos_offset
    pos = pos.clamp(max=cfg.max_positions - 1)
    x = F.embedding(ids, _t(w, "word_emb")) + F.embedding(pos, _t(w, "pos_emb"))
    if cfg.type_vocab > 0:
        x = x + F.embedding(types, _t(w, "type_emb"))
    H, heads = cfg.hidden, cfg.n_heads
    dh = H // heads
    x = F.layer_norm(x, (H,), _t(w, "emb_ln_g"), _t(w, "emb_ln_b"), cfg.ln_eps)
    bias = (1.0 - mask)[:, None, None, :] * torch.finfo(torch.float32).min  # fp32's lowest, in either dtype
    for l in range(cfg.n_layers):
        p = f"layer{l}."
        qkv = F.linear(x, _t(w, p + "qkv_w"), _t(w, p + "qkv_b"))
        q, k, v = (t.view(n, L, heads, dh).transpose(1, 2) for t in qkv.split(H, dim=-1))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + bias, dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(n, L, H)
        x = F.layer_norm(F.linear(ctx, _t(w, p + "attn_out_w"), _t(w, p + "attn_out_b")) + x, (H,),
                         _t(w, p + "ln1_g"), _t(w, p + "ln1_b"), cfg.ln_eps)
        f = _act(F.linear(x, _t(w, p + "ffn_in_w"), _t(w, p + "ffn_in_b")), cfg.act)
        x = F.layer_norm(F.linear(f, _t(w, p + "ffn_out_w"), _t(w, p + "ffn_out_b")) + x, (H,),
                         _t(w, p + "ln2_g"), _t(w, p + "ln2_b"), cfg.ln_eps)
    return [x[i, : len(s)].numpy().copy() for i, s in enumerate(seqs)]


def embed(cfg: Any, w: dict[str, np.ndarray], seqs: Sequence[Sequence[int]],
          type_seqs: Sequence[Sequence[int]] | None = None, normalize: bool = True,
          pooling: str | None = None) -> np.ndarray:
    """Sentence embeddings: pooling over the last hidden state + optional L2 normalisation."""
    hs = hidden_states(cfg, w, seqs, type_seqs)
    mode = pooling or cfg.pooling
    pooled = np.stack([h[0] if mode == "cls" else h.mean(axis=0, dtype=np.float32) for h in hs]).astype(np.float32)
    if normalize:
        t = torch.from_numpy(pooled)
        pooled = F.normalize(t, p=2, dim=1).numpy()
    return pooled


def classify(cfg: Any, w: dict[str, np.ndarray], seqs: Sequence[Sequence[int]],
             type_seqs: Sequence[Sequence[int]] | None = None, sigmoid: bool = True) -> np.ndarray:
    """Classifier head on the first token: tanh(dense) -> out projection (-> sigmoid)."""
    hs = hidden_states(cfg, w, seqs, type_seqs)
    cls = torch.from_numpy(np.stack([h[0] for h in hs]))
    pooled = torch.tanh(F.linear(cls, _t(w, "head_dense_w"), _t(w, "head_dense_b")))
    logits = F.linear(pooled, _t(w, "head_out_w"), _t(w, "head_out_b"))
    return (torch.sigmoid(logits) if sigmoid else logits).numpy()


def rerank_order(scores: Sequence[float], top_n: int | None = None) -> list[int]:
    """Indices in the order reranker.py:270-272 returns documents: stable sort, score descending."""
    idx = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)
    return idx[: len(scores) if top_n is None else top_n]
