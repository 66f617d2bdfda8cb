"""Writes the transformer-side golden fixtures SURVEY.md §8(c) lists, with `transformers` itself as the source of
the expected values (the upstream dependency the reference's embedder and reranker delegate to:
src/pipeline/components/embedding.py:127-133, src/pipeline/components/reranker.py:237-272):

    encoder_minilm_tiny.npz    BertModel -> mean pooling -> L2 normalise   (all-MiniLM-L6-v2's pipeline)
    encoder_bge_cls_tiny.npz   BertModel -> first token  -> L2 normalise   (bge-base-en-v1.5's pipeline)
    rerank_order.npz           BertForSequenceClassification logits -> .view(-1).float() -> sigmoid -> stable
                               descending sort -> top_n slice, with two exactly equal scores in the list
    scheduler_traces.json      flush reasons / batch sizes of scripted arrival patterns (see below)

Run in the BUILD container only (it has `transformers`; the GPU box needs none of it once these files exist):

    python tests/golden/make_golden_transformer.py

No trained checkpoint or vocabulary exists offline, so the models are tiny architectures with seeded weights.
The weights are generated HERE (numpy, rounded to bf16-representable fp32 values so that a file stores two bytes per
weight, exactly), loaded into the `transformers` module through its own state-dict names, and the module's outputs
are recorded: a fixture is (config, weights, token ids) -> (what transformers computed).  tests/test_golden_transformer.py
holds oracle/bert.py to them on the CPU and the HIP kernels to them under `-m gpu`.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rag_inference_pipeline_amd.bert import BertConfig  # noqa: E402


def bf16_exact(a: np.ndarray) -> np.ndarray:
    """fp32 values whose low 16 mantissa bits are zero (round to nearest even): stored as uint16, restored exactly."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def tiny_weights(cfg: BertConfig, seed: int) -> dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in cfg.weight_shapes().items():
        base = name.split(".")[-1]
        if base.endswith("_g"):
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif base.endswith("_b"):
            w = 0.05 * rng.standard_normal(shape)
        elif base.endswith("_emb"):
            w = 0.5 * rng.standard_normal(shape)
        else:
            w = 0.12 * rng.standard_normal(shape)
        out[name] = bf16_exact(w.astype(np.float32))
    return out


def hf_state_dict(cfg: BertConfig, w: dict[str, np.ndarray], prefix: str) -> dict[str, torch.Tensor]:
    """Canonical names -> Hugging Face BERT state-dict names (the inverse of bert.weights_from_hf_state_dict)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    H = cfg.hidden
    sd = {prefix + "embeddings.word_embeddings.weight": t(w["word_emb"]),
          prefix + "embeddings.position_embeddings.weight": t(w["pos_emb"]),
          prefix + "embeddings.token_type_embeddings.weight": t(w["type_emb"]),
          prefix + "embeddings.LayerNorm.weight": t(w["emb_ln_g"]), prefix + "embeddings.LayerNorm.bias": t(w["emb_ln_b"])}
    for l in range(cfg.n_layers):
        p, q = f"{prefix}encoder.layer.{l}.", f"layer{l}."
        for i, n in enumerate(("query", "key", "value")):
            sd[p + f"attention.self.{n}.weight"] = t(w[q + "qkv_w"][i * H:(i + 1) * H])
            sd[p + f"attention.self.{n}.bias"] = t(w[q + "qkv_b"][i * H:(i + 1) * H])
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = t(w[q + "attn_out_w"]), t(w[q + "attn_out_b"])
        sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"] = t(w[q + "ln1_g"]), t(w[q + "ln1_b"])
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = t(w[q + "ffn_in_w"]), t(w[q + "ffn_in_b"])
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = t(w[q + "ffn_out_w"]), t(w[q + "ffn_out_b"])
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = t(w[q + "ln2_g"]), t(w[q + "ln2_b"])
    if cfg.head == "bert":
        sd[prefix + "pooler.dense.weight"], sd[prefix + "pooler.dense.bias"] = t(w["head_dense_w"]), t(w["head_dense_b"])
        sd["classifier.weight"], sd["classifier.bias"] = t(w["head_out_w"]), t(w["head_out_b"])
    return sd


def pad(seqs, types):
    L = max(len(s) for s in seqs)
    ids = torch.zeros((len(seqs), L), dtype=torch.long)
    tt = torch.zeros((len(seqs), L), dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = torch.tensor(s)
        tt[i, :len(s)] = torch.tensor(types[i])
        mask[i, :len(s)] = 1
    return ids, tt, mask


def save(name: str, cfg: BertConfig, w: dict[str, np.ndarray], seqs, types, **expected) -> None:
    cfg_json = json.dumps({k: getattr(cfg, k) for k in ("vocab_size", "hidden", "n_layers", "n_heads", "intermediate",
                                                        "max_positions", "type_vocab", "pos_offset", "act", "head",
                                                        "n_labels", "ln_eps", "pooling")})
    arrays = {"w." + k: (v.view(np.uint32) >> 16).astype(np.uint16) for k, v in w.items()}
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, config=np.array(cfg_json), lens=lens,
                        ids=np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs]),
                        types=np.concatenate([np.asarray(t, dtype=np.int32) for t in types]), **arrays, **expected)
    print("wrote", path, os.path.getsize(path), "bytes")


def make_encoder(name: str, pooling: str, seed: int, lens) -> None:
    import transformers

    cfg = BertConfig(vocab_size=120, hidden=64, n_layers=2, n_heads=2, intermediate=128, max_positions=48,
                     type_vocab=2, pooling=pooling)
    w = tiny_weights(cfg, seed)
    hf = transformers.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.n_layers,
                                 num_attention_heads=cfg.n_heads, intermediate_size=cfg.intermediate,
                                 max_position_embeddings=cfg.max_positions, type_vocab_size=cfg.type_vocab)
    model = transformers.BertModel(hf, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(hf_state_dict(cfg, w, ""), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed + 100)
    seqs = [rng.integers(3, cfg.vocab_size, size=int(n)).tolist() for n in lens]
    types = [[0] * len(s) for s in seqs]      # single sentences, as SentenceTransformer.encode feeds them
    ids, tt, mask = pad(seqs, types)
    with torch.no_grad():
        h = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).last_hidden_state
        m = mask[..., None].float()
        pooled = h[:, 0] if pooling == "cls" else (h * m).sum(1) / m.sum(1)   # sentence-transformers Pooling module
        emb = torch.nn.functional.normalize(pooled, p=2, dim=1)               # sentence-transformers Normalize module
    hidden = np.concatenate([h[i, :len(s)].numpy() for i, s in enumerate(seqs)])
    save(name, cfg, w, seqs, types, hidden=hidden.astype(np.float32), embedding=emb.numpy().astype(np.float32))


def make_rerank(name: str, seed: int) -> None:
    import transformers

    cfg = BertConfig(vocab_size=120, hidden=64, n_layers=2, n_heads=2, intermediate=128, max_positions=48,
                     type_vocab=2, head="bert", n_labels=1)
    w = tiny_weights(cfg, seed)
    hf = transformers.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.n_layers,
                                 num_attention_heads=cfg.n_heads, intermediate_size=cfg.intermediate,
                                 max_position_embeddings=cfg.max_positions, type_vocab_size=cfg.type_vocab, num_labels=1)
    model = transformers.BertForSequenceClassification(hf).eval()
    missing, unexpected = model.load_state_dict(hf_state_dict(cfg, w, "bert."), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed + 100)
    query = rng.integers(3, cfg.vocab_size, size=6).tolist()
    docs = [rng.integers(3, cfg.vocab_size, size=int(n)).tolist() for n in rng.integers(5, 30, size=11)]
    docs.insert(7, list(docs[2]))             # the same document twice: two EXACTLY equal scores -> the sort must be stable
    seqs = [query + d for d in docs]
    types = [[0] * len(query) + [1] * len(d) for d in docs]
    ids, tt, mask = pad(seqs, types)
    with torch.no_grad():   # centre and spread the logits (a flat or saturated sigmoid would make the order hang on rounding)
        raw = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits.view(-1).float().numpy()
    gain = 1.5 / max(float(raw.std()), 1e-6)
    w["head_out_w"] = bf16_exact(w["head_out_w"] * gain)
    w["head_out_b"] = bf16_exact(np.array([-gain * (float(np.median(raw)) - float(w["head_out_b"][0]))], dtype=np.float32))
    model.load_state_dict(hf_state_dict(cfg, w, "bert."), strict=False)
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits.view(-1).float()
        scores = torch.sigmoid(logits)        # reranker.py:252
    sc = scores.numpy().astype(np.float32)
    assert sc[2] == sc[7]
    order = sorted(range(len(sc)), key=lambda i: sc[i], reverse=True)   # reranker.py:270 (list.sort is stable)
    assert order.index(2) + 1 == order.index(7)
    gaps = np.abs(np.diff(np.sort(sc)))
    assert np.sort(gaps)[1] > 2e-4, gaps      # apart from the planted tie the order does not hang on rounding
    save(name, cfg, w, seqs, types, logits=logits.numpy().astype(np.float32), scores=sc,
         order=np.array(order, dtype=np.int32), top_n=np.array(5, dtype=np.int32),
         order_top_n=np.array(order[:5], dtype=np.int32), n_query_tokens=np.array(len(query), dtype=np.int32))


# Scheduler traces.  Each script is a list of (delay before the request in ms, request id); the expected flushes follow
# from the reference's rules (services/gateway/batch_scheduler.py): a batch is flushed when it reaches batch_size
# ("full", :201-203), when the timer started by its FIRST request fires ("timeout", :206-208, :221-236), or by stop()
# ("shutdown", :160-165); sizes and reasons below are those rules applied by hand to each script, mirroring the
# reference's own timing tests (tests/test_batch_scheduler.py:313-369: 3 requests at batch_size 3 -> one batch of 3;
# one request at batch_size 10, 50 ms -> one batch of 1).  Delays are chosen far from the timer so that a loaded CI
# machine does not blur them (gaps of >= 4x).
SCHEDULER_TRACES = [
    {"name": "full_batch_at_max_size", "batch_size": 3, "max_batch_delay_ms": 1000, "stop_after_ms": 0,
     "arrivals": [[0, "a0"], [0, "a1"], [0, "a2"]],
     "flushes": [{"size": 3, "reason": "full", "ids": ["a0", "a1", "a2"]}]},
    {"name": "single_request_times_out", "batch_size": 10, "max_batch_delay_ms": 50, "stop_after_ms": 200,
     "arrivals": [[0, "single"]],
     "flushes": [{"size": 1, "reason": "timeout", "ids": ["single"]}]},
    {"name": "nine_requests_batch_of_four", "batch_size": 4, "max_batch_delay_ms": 40, "stop_after_ms": 200,
     "arrivals": [[0, f"n{i}"] for i in range(9)],
     "flushes": [{"size": 4, "reason": "full", "ids": ["n0", "n1", "n2", "n3"]},
                 {"size": 4, "reason": "full", "ids": ["n4", "n5", "n6", "n7"]},
                 {"size": 1, "reason": "timeout", "ids": ["n8"]}]},
    {"name": "timer_runs_from_the_first_request", "batch_size": 8, "max_batch_delay_ms": 80, "stop_after_ms": 400,
     "arrivals": [[0, "t0"], [20, "t1"], [20, "t2"], [200, "t3"]],
     "flushes": [{"size": 3, "reason": "timeout", "ids": ["t0", "t1", "t2"]},
                 {"size": 1, "reason": "timeout", "ids": ["t3"]}]},
    {"name": "stop_flushes_what_is_pending", "batch_size": 8, "max_batch_delay_ms": 2000, "stop_after_ms": 20,
     "arrivals": [[0, "s0"], [0, "s1"]],
     "flushes": [{"size": 2, "reason": "shutdown", "ids": ["s0", "s1"]}]},
]
# AdaptiveBatchPolicy.update (batch_scheduler.py:28-76): delay = clamp(0.7 * previous + 0.3 * target), target = max at
# mean depth >= max_batch_size, else min + (mean depth / max_batch_size) * (max - min); mean over the last 10 depths;
# the first `previous` is min_delay.  Values computed here in float64 by that formula.
POLICY_TRACES = [
    {"max_batch_size": 8, "min_delay_sec": 0.01, "max_delay_sec": 0.2, "depths": [4, 4, 8, 16, 16, 1, 0, 0]},
    {"max_batch_size": 32, "min_delay_sec": 0.01, "max_delay_sec": 0.05, "depths": [1, 2, 3, 32, 32, 32, 32, 32, 32, 32, 32, 32, 1]},
]


def policy_expected(tr: dict) -> list[float]:
    lo, hi, cur, hist, out = tr["min_delay_sec"], tr["max_delay_sec"], tr["min_delay_sec"], [], []
    for depth in tr["depths"]:
        hist = (hist + [depth])[-10:]
        mean = sum(hist) / len(hist)
        target = hi if mean >= tr["max_batch_size"] else lo + min(mean / tr["max_batch_size"], 1.0) * (hi - lo)
        cur = 0.7 * cur + 0.3 * target
        out.append(max(lo, min(cur, hi)))
    return out


def make_scheduler_traces() -> None:
    doc = {"scheduler": SCHEDULER_TRACES,
           "adaptive_policy": [dict(tr, delays=policy_expected(tr)) for tr in POLICY_TRACES]}
    path = os.path.join(HERE, "scheduler_traces.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    make_encoder("encoder_minilm_tiny.npz", "mean", 11, [5, 12, 9, 31, 3, 17, 1, 24])
    make_encoder("encoder_bge_cls_tiny.npz", "cls", 12, [8, 20, 14, 40, 2, 11])
    make_rerank("rerank_order.npz", 13)
    make_scheduler_traces()
