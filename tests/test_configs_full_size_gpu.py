"""BASELINE.json configs[1] and configs[2] at their FULL sizes, on the GPU, against the oracle.

  B  1 x MI355X: all-MiniLM-L6-v2 architecture encoder in front of a 1M x 384 brute-force scan, batch 32, k = 10
  C  B + cross-encoder ms-marco-MiniLM-L-6-v2 architecture, top-100 -> top-10: 32 x 100 = 3200 pairs per batch
     through Reranker.rerank_batch (reference: components/reranker.py:274-308)

No trained checkpoint exists offline, so both models are seeded random weights of the named architecture
(`synthetic:` names, model_source.py) with the deterministic hashing tokenizer; throughput and parity do not
depend on weight values.  The 1M x 384 corpus is small enough for the oracle to search WHOLE (1.5 GB of
host rows, a few hundred ms per batch), so the scan is held bit-exact at full size, not through properties.
"""
import numpy as np
import pytest

from oracle import bert as obert
from oracle import flat as oracle

pytestmark = pytest.mark.gpu

N, D, B = 1_000_000, 384, 32
WORDS = ("retrieval augmented generation pipeline vector index query document embedding transformer attention "
         "gpu memory bandwidth kernel matrix latency throughput batch scheduler cache shard merge score").split()


def _queries():
    rng = np.random.default_rng(4321)
    return [" ".join(rng.choice(WORDS, size=int(rng.integers(6, 16)))) + "?" for _ in range(B)]


def _doc_text(doc_id: int) -> str:
    # ~25 words per synthetic document, as the reference's generator writes them (scripts/create_test_docs.py:47)
    rng = np.random.default_rng(1_000_003 * (doc_id + 1))
    return " ".join(rng.choice(WORDS, size=int(rng.integers(20, 31))))


@pytest.fixture(scope="module")
def corpus_and_index(gpu_required):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    X = oracle.synth_rows(1234, 0, N, D)
    idx = FlatIndex(D)
    idx.add_synthetic(N, 1234)
    yield X, idx
    idx.close()


@pytest.fixture(scope="module")
def encoded_queries(gpu_required):
    """32 text queries through the HIP encoder (EmbeddingGenerator.encode, reference embedding.py:100-175) and
    through the oracle encoder on the same token ids."""
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    from rag_inference_pipeline_amd.config import PipelineSettings
    from rag_inference_pipeline_amd.model_source import resolve_model
    name = "synthetic:all-MiniLM-L6-v2:11"
    gen = EmbeddingGenerator(PipelineSettings(embedding_model_name=name, DISABLE_CACHE_FOR_PROFILING="true", faiss_dim=D))
    gen.load()
    queries = _queries()
    emb = gen.encode(queries)
    gen.unload()
    cfg, w, tok, max_len = resolve_model(name, "embedding")
    ids, types = tok.encode_batch(queries, max_len)
    return queries, emb, obert.embed(cfg, w, ids, types)


def test_config_b_encoder_then_1m_x_384_scan_batch_32(corpus_and_index, encoded_queries):
    """Config B end to end at full size: embeddings within 1e-5 of the oracle encoder (unit-norm rows), then
    ids AND score bits of the 1M-row scan identical to the oracle's — one-pass and two-stage, k = 10 and the
    k = 100 that config C retrieves."""
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, SCREEN_OFF
    X, idx = corpus_and_index
    _, emb, emb_oracle = encoded_queries
    assert emb.shape == (B, D) and emb.dtype == np.float32
    assert np.abs(emb - emb_oracle).max() < 1e-5                    # fp32 tolerance of the encoder (DESIGN.md "Bars")
    assert np.abs(np.linalg.norm(emb.astype(np.float64), axis=1) - 1).max() < 1e-6
    for k in (10, 100):
        Do, Io = oracle.search(X, emb, k)                           # the WHOLE corpus through the oracle
        for mode in (SCREEN_OFF, SCREEN_FP16):
            idx.set_screening(mode)
            Dg, Ig = idx.search(emb, k)
            np.testing.assert_array_equal(Ig, Io)
            np.testing.assert_array_equal(Dg.view(np.uint32), Do.view(np.uint32))
        # the oracle-encoded queries rank the same rows away from near-ties at the k boundary (north star:
        # "doc ids identical to the CPU embedder + FAISS path")
        Dq, Iq = oracle.search(X, emb_oracle, k + 1)
        for b in range(B):
            if Dq[b, k - 1] - Dq[b, k] > 1e-4:                      # gap above the 1e-5 embedding tolerance x norms
                assert set(Iq[b, :k].tolist()) == set(Io[b].tolist())
    assert idx.screen_stats()["fallbacks"] == 0
    # planted queries: a corpus row is its own nearest neighbour
    planted = X[[7, 500_000, N - 1]]
    Dp, Ip = idx.search(planted, 10)
    assert Ip[:, 0].tolist() == [7, 500_000, N - 1]


def test_config_c_rerank_top100_to_top10_3200_pairs(corpus_and_index, encoded_queries):
    """Config C at full size: the 32 x 100 documents retrieved above go through ONE Reranker.rerank_batch call
    (3200 pairs, ~180k tokens, the big-batch split-bf16 GEMM path).  Contract of the reference's rerank
    (reranker.py:237-272): stable descending sort by sigmoid score, `top_n` slices it.  Scores of a sample of
    queries are held to oracle/bert.py within the fp32 tolerance, order likewise away from near-equal scores."""
    from rag_inference_pipeline_amd.components.reranker import Reranker
    from rag_inference_pipeline_amd.components.schemas import Document
    from rag_inference_pipeline_amd.config import PipelineSettings
    from rag_inference_pipeline_amd.model_source import resolve_model
    X, idx = corpus_and_index
    queries, emb, _ = encoded_queries
    _, I = idx.search(emb, 100)
    docs_batch = [[Document(doc_id=int(i), title=f"Document {int(i)}", content=_doc_text(int(i)), category="general")
                   for i in row] for row in I]
    name = "synthetic:ms-marco-MiniLM-L-6-v2:12"
    rr = Reranker(PipelineSettings(reranker_model_name=name))
    rr.load()
    full = rr.rerank_batch(queries, docs_batch)                      # top_n=None: all 100, as the retrieval node calls it
    top10 = rr.rerank_batch(queries, docs_batch, top_n=10)
    rr.unload()
    assert len(full) == B and all(len(r) == 100 for r in full) and all(len(r) == 10 for r in top10)
    for b in range(B):
        scores = [d.score for d in full[b]]
        assert scores == sorted(scores, reverse=True) and all(0.0 < s < 1.0 for s in scores)
        assert sorted(d.doc_id for d in full[b]) == sorted(I[b].tolist())            # a permutation of the input
        assert [(d.doc_id, d.score) for d in top10[b]] == [(d.doc_id, d.score) for d in full[b][:10]]
        pos = {int(i): j for j, i in enumerate(I[b])}
        for a, c in zip(full[b], full[b][1:]):                                        # stable: ties keep retrieval order
            if a.score == c.score:
                assert pos[a.doc_id] < pos[c.doc_id]
    cfg, w, tok, max_len = resolve_model(name, "reranker")
    for b in (0, 13, 31):                                            # 300 of the 3200 pairs through the oracle
        pids, ptypes = tok.encode_pairs([queries[b]] * 100, [d.content for d in docs_batch[b]], min(512, max_len))
        want = obert.classify(cfg, w, pids, ptypes)[:, 0]
        got = {d.doc_id: d.score for d in full[b]}
        diffs = [abs(got[int(i)] - float(s)) for i, s in zip(I[b], want)]
        assert max(diffs) < 2e-5, max(diffs)                         # sigmoid scores, fp32 tolerance (test_components_gpu bar)
        order = sorted(range(100), key=lambda j: float(want[j]), reverse=True)
        ws = [float(want[j]) for j in order]
        for r, j in enumerate(order):                                # same rank unless the oracle's neighbours are within tolerance
            if (r == 0 or ws[r - 1] - ws[r] > 1e-4) and (r == 99 or ws[r] - ws[r + 1] > 1e-4):
                assert full[b][r].doc_id == int(I[b][j])


def test_config_c_in_the_references_gpu_precision_3200_pairs(corpus_and_index, encoded_queries):
    """Config C with `RAG_AMD_RERANKER_DTYPE=f16` — the precision the reference runs its reranker at on a GPU
    (reranker.py:91-93) — at full size: 3200 pairs through fp16 activations stored in MFMA-fragment order.  The rerank
    contract is the same (stable descending sort, `top_n` slices it); scores stay within half precision's reach of the
    fp32 oracle (5e-3 on sigmoid scores, the bar of test_cross_encoder_fp16_gemm_mode_tracks_fp32_oracle), and a
    document the oracle ranks clearly first (by more than twice that) is ranked first."""
    from rag_inference_pipeline_amd.components.reranker import Reranker
    from rag_inference_pipeline_amd.components.schemas import Document
    from rag_inference_pipeline_amd.config import PipelineSettings
    from rag_inference_pipeline_amd.model_source import resolve_model
    X, idx = corpus_and_index
    queries, emb, _ = encoded_queries
    _, I = idx.search(emb, 100)
    docs_batch = [[Document(doc_id=int(i), title=f"Document {int(i)}", content=_doc_text(int(i)), category="general")
                   for i in row] for row in I]
    name = "synthetic:ms-marco-MiniLM-L-6-v2:12"
    rr = Reranker(PipelineSettings(reranker_model_name=name, RAG_AMD_RERANKER_DTYPE="f16"))
    rr.load()
    assert rr.model.cfg.gemm_dtype == "f16"
    full = rr.rerank_batch(queries, docs_batch)
    top10 = rr.rerank_batch(queries, docs_batch, top_n=10)
    rr.unload()
    assert len(full) == B and all(len(r) == 100 for r in full) and all(len(r) == 10 for r in top10)
    for b in range(B):
        scores = [d.score for d in full[b]]
        assert scores == sorted(scores, reverse=True) and all(0.0 <= s <= 1.0 for s in scores)
        assert sorted(d.doc_id for d in full[b]) == sorted(I[b].tolist())
        assert [(d.doc_id, d.score) for d in top10[b]] == [(d.doc_id, d.score) for d in full[b][:10]]
    cfg, w, tok, max_len = resolve_model(name, "reranker")
    for b in (0, 13, 31):
        pids, ptypes = tok.encode_pairs([queries[b]] * 100, [d.content for d in docs_batch[b]], min(512, max_len))
        want = obert.classify(cfg, w, pids, ptypes)[:, 0]
        got = {d.doc_id: d.score for d in full[b]}
        diffs = [abs(got[int(i)] - float(s)) for i, s in zip(I[b], want)]
        assert max(diffs) < 5e-3, max(diffs)
        order = sorted(range(100), key=lambda j: float(want[j]), reverse=True)
        if float(want[order[0]]) - float(want[order[1]]) > 1e-2:
            assert full[b][0].doc_id == int(I[b][order[0]])


def test_config_b_size_on_the_surveys_gaussian_corpus(gpu_required):
    """SURVEY 8(d)'s synthetic inputs, literally: corpus rows from `np.random.default_rng(1234 + chunk)
    .standard_normal` in chunks, L2-normalised; queries from `default_rng(4321)`; a second query set of perturbed
    corpus rows (x + 0.05 noise, renormalised) whose true top-1 is known a priori.  1M x 384, batch 32, k = 10,
    one-pass and two-stage, against the oracle over the whole corpus.  (bench.py uses the integer-hash generator
    instead because the CPU must regenerate single rows of a 10M-row corpus bit-exactly; this test holds the same
    kernels to the oracle on the distribution the survey names.)"""
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, SCREEN_OFF, FlatIndex
    chunk = 250_000
    idx = FlatIndex(D)
    parts = []
    for c in range(N // chunk):
        x = np.random.default_rng(1234 + c).standard_normal((chunk, D), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        parts.append(x.astype(np.float32))
        idx.add(parts[-1])
    X = np.concatenate(parts)
    q = np.random.default_rng(4321).standard_normal((B, D), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    rows = np.random.default_rng(99).integers(0, N, size=B)
    p = X[rows] + 0.05 * np.random.default_rng(5).standard_normal((B, D), dtype=np.float32)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    for Q in (q.astype(np.float32), p.astype(np.float32)):
        Do, Io = oracle.search(X, Q, 10)
        for mode in (SCREEN_OFF, SCREEN_FP16):
            idx.set_screening(mode)
            Dg, Ig = idx.search(Q, 10)
            np.testing.assert_array_equal(Ig, Io)
            np.testing.assert_array_equal(Dg.view(np.uint32), Do.view(np.uint32))
    assert Io[:, 0].tolist() == rows.tolist()      # the perturbed rows come back as their own nearest neighbour
    assert idx.screen_stats()["fallbacks"] == 0
    idx.close()
