"""Where a 32 x 100 rerank batch's host time goes (reranker component, synthetic checkpoint)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.components.reranker import Reranker

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
WORDS = ("retrieval augmented generation pipeline vector index query document embedding transformer attention "
         "gpu memory bandwidth kernel matrix latency throughput batch scheduler cache shard merge score").split()
rng = np.random.default_rng(0)
s = PipelineSettings(reranker_model_name="synthetic:ms-marco-MiniLM-L-6-v2", RAG_AMD_RERANKER_DTYPE=dtype)
rr = Reranker(s); rr.load()
queries = [" ".join(rng.choice(WORDS, size=int(rng.integers(6, 16)))) + "?" for _ in range(32)]
docs = [" ".join(rng.choice(WORDS, size=25)) for _ in range(3200)]
fq = [q for q in queries for _ in range(100)]
max_len = min(int(s.truncate_length), rr._max_len)

def t(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r

ms, packed = t(lambda: rr.tokenizer.encode_pairs_packed(fq, docs, max_len, rr.model.cfg.type_vocab > 1))
print(f"tokenise + pack 3200 pairs, one native call : {ms:7.2f} ms")
ids, types, cu = packed
ms, _ = t(lambda: rr.model.classify_packed(ids, types, cu, sigmoid=True))
print(f"classify_packed, 3200 pairs in one pass      : {ms:7.2f} ms  ({int(cu[-1])} tokens)")
for lo, hi in ((0, 160), (0, 640)):
    sub = (ids[cu[lo]:cu[hi]], types[cu[lo]:cu[hi]] if types is not None else None, cu[lo:hi + 1] - cu[lo])
    ms, _ = t(lambda: rr.model.classify_packed(*sub, sigmoid=True))
    print(f"classify_packed, {hi - lo} pairs                  : {ms:7.2f} ms")
ms, sc = t(lambda: rr._score_pairs(fq, docs))
print(f"_score_pairs (chunked, tokeniser thread)     : {ms:7.2f} ms")
