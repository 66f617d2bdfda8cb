"""Host time of one query-encoding call (32 text queries, MiniLM-L6 architecture): tokenise / pack / the C call that
enqueues the forward pass (returns without waiting) / the GPU time of the pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
from rag_inference_pipeline_amd.bert import pack_sequences

WORDS = ("retrieval augmented generation pipeline vector index query document embedding transformer attention "
         "gpu memory bandwidth kernel matrix latency throughput batch scheduler cache shard merge score").split()
rng = np.random.default_rng(0)
s = PipelineSettings(embedding_model_name="synthetic:all-MiniLM-L6-v2", DISABLE_CACHE_FOR_PROFILING="true")
emb = EmbeddingGenerator(s); emb.load()
queries = [" ".join(rng.choice(WORDS, size=int(rng.integers(6, 16)))) + "?" for _ in range(32)]

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    el = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return el, r

ms, tok = t(lambda: emb._tokenizer.encode_batch(list(queries), emb._max_len))
print(f"tokenise 32 queries                      : {ms * 1e3:7.1f} us")
ids, types = tok
ms, _ = t(lambda: pack_sequences(ids, types))
print(f"pack_sequences                           : {ms * 1e3:7.1f} us")
def enq():
    d = emb._model.embed_to_device(ids, types, normalize=True)
    return d
ms, d = t(enq)
print(f"embed_to_device (host side, no wait)     : {ms * 1e3:7.1f} us   (includes pack_sequences and the output tensor)")
def enq_wait():
    d = emb._model.embed_to_device(ids, types, normalize=True); torch.cuda.synchronize(); return d
ms, d = t(enq_wait, 100)
print(f"embed_to_device + device synchronize     : {ms * 1e3:7.1f} us")
ms, d = t(lambda: emb.encode_device(queries))
print(f"encode_device (texts in, handle out)     : {ms * 1e3:7.1f} us")
