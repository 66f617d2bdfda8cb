import os
"""Whole retrieval-node batch (BASELINE configs[1] / [2]) through the Python components and the
executor: text queries -> encoder -> 1M x 384 scan + top-k -> SQLite document fetch -> (cross-encoder).
Reports per-stage wall time from the executor's stage timers and batches/s; this is where host-side
costs (tokenisation, document objects) show up next to the kernels.

    python scripts/bench_pipeline.py [--rows 1000000] [--rerank] [--k 100] [--batches 10]
"""
import argparse, os, sqlite3, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_inference_pipeline_amd.batch_scheduler import Batch
from rag_inference_pipeline_amd.component_registry import ComponentRegistry
from rag_inference_pipeline_amd.components.document_store import DocumentStore
from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
from rag_inference_pipeline_amd.components.reranker import Reranker
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, FlatIndex
from rag_inference_pipeline_amd.retrieval_executor import RetrievalExecutor
from rag_inference_pipeline_amd.schemas import PendingRequest
from rag_inference_pipeline_amd.telemetry import stage_timers

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--rerank", action="store_true")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--batches", type=int, default=10)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--profile", action="store_true", help="cProfile one batch and print the top host functions")
ap.add_argument("--dim", type=int, default=384, help="index width; 768 switches the embedder to the bge-base architecture")
ap.add_argument("--threads", type=int, default=1, help="batches in flight (the scheduler runs batches concurrently)")
ap.add_argument("--encoder-cus", type=int, default=0, help="RAG_AMD_ENCODER_CUS: the encoder's share of the chip (0 = off)")
ap.add_argument("--one-pass", action="store_true", help="RAG_AMD_TWO_STAGE=false: the one-pass fp32 scan")
ap.add_argument("--no-gc", action="store_true", help="run the timed batches with the cyclic garbage collector off (diagnosis)")
ap.add_argument("--per-batch", action="store_true", help="print every timed batch's wall time")
a = ap.parse_args()

WORDS = ("retrieval augmented generation pipeline vector index query document embedding transformer attention "
         "gpu memory bandwidth kernel matrix latency throughput batch scheduler cache shard merge score").split()
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "documents"))
con = sqlite3.connect(os.path.join(tmp, "documents", "documents.db"))
con.execute("CREATE TABLE documents (doc_id INTEGER PRIMARY KEY, title TEXT, content TEXT, category TEXT)")
t0 = time.time()
CH = 100_000
for lo in range(0, a.rows, CH):
    n = min(CH, a.rows - lo)
    words = rng.choice(WORDS, size=(n, 25))  # ~25 words per synthetic doc (reference scripts/create_test_docs.py:47)
    con.executemany("INSERT INTO documents VALUES (?,?,?,?)",
                    ((lo + i, f"Document {lo + i}", " ".join(words[i]), "general") for i in range(n)))
con.commit(); con.close()
print(f"sqlite: {a.rows} docs in {time.time() - t0:.1f}s", flush=True)

settings = PipelineSettings(DOCUMENTS_DIR=os.path.join(tmp, "documents"), DOCUMENTS_PAYLOAD_MODE="full",
                            DISABLE_CACHE_FOR_PROFILING="true", faiss_dim=a.dim, retrieval_k=a.k,
                            embedding_model_name="synthetic:bge-base-en-v1.5" if a.dim == 768 else "synthetic:all-MiniLM-L6-v2",
                            reranker_model_name="synthetic:ms-marco-MiniLM-L-6-v2",
                            RAG_AMD_RERANKER_DTYPE=a.dtype, RAG_AMD_ENCODER_CUS=a.encoder_cus,
                            RAG_AMD_TWO_STAGE="false" if a.one_pass else "true")


class SyntheticStore(FAISSStore):  # the product's FAISSStore (its search(), two-stage default, device hand-off) over a
    def __init__(self, settings, n, d):   # synthetic corpus generated on the GPU (no 1.5 GB file round trip)
        super().__init__(settings)
        idx = FlatIndex(d); idx.add_synthetic(n, 1234)
        if getattr(settings, "faiss_two_stage", True):
            idx.set_screening(SCREEN_FP16)
        self._index, self._ntotal, self._is_loaded = idx, n, True

reg = ComponentRegistry()
emb = EmbeddingGenerator(settings); reg.register("embedding_generator", emb, emb.load)
reg.register("faiss_store", SyntheticStore(settings, a.rows, a.dim))
reg.register("document_store", DocumentStore(settings))
if a.rerank:
    rr = Reranker(settings); reg.register("reranker", rr, rr.load)
ex = RetrievalExecutor(reg, settings)
queries = [" ".join(rng.choice(WORDS, size=int(rng.integers(6, 16)))) + "?" for _ in range(32)]
mk = lambda: Batch(1, [PendingRequest(request_id=f"r{i}", query=q, timestamp=time.time()) for i, q in enumerate(queries)])
ex._process_batch_sync(mk())  # warm
stage_timers.reset()
extra = {"tokenize_pairs": 0.0}
if a.no_gc:
    import gc
    gc.collect(); gc.disable()
per = []
if a.threads > 1:   # several batches in flight, as the scheduler runs them (batch_scheduler.py:286-288)
    from concurrent.futures import ThreadPoolExecutor
    ex._process_batch_sync(mk())
    with ThreadPoolExecutor(max_workers=a.threads) as pool:
        t0 = time.perf_counter()
        futs = [pool.submit(ex._process_batch_sync, mk()) for _ in range(a.batches)]
        items = [f.result() for f in futs][-1]
        el = time.perf_counter() - t0
else:
    t0 = time.perf_counter()
    for _ in range(a.batches):
        t1 = time.perf_counter()
        items = ex._process_batch_sync(mk())
        per.append(time.perf_counter() - t1)
    el = time.perf_counter() - t0
if a.per_batch:
    print("   per batch (ms):", " ".join(f"{x * 1e3:.1f}" for x in per))
print(f"rows={a.rows} dim={a.dim} k={a.k} rerank={a.rerank} dtype={a.dtype} threads={a.threads} encoder_cus={a.encoder_cus}"
      f"{' one-pass' if a.one_pass else ''}: {el / a.batches * 1e3:.2f} ms/batch  {32 * a.batches / el:.0f} queries/s")
snap = stage_timers.snapshot()
acc = 0.0
for s, v in snap.items():
    print(f"   {s:28s} {v['seconds'] / a.batches * 1e3:8.2f} ms/batch")
    acc += v["seconds"]
print(f"   {'(rest: rerank + objects)':28s} {(el - acc) / a.batches * 1e3:8.2f} ms/batch")
assert len(items) == 32 and len(items[0].docs) == a.k
if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); ex._process_batch_sync(mk()); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
