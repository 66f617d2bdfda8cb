"""GPU: the drop-in components behind the reference's plugin surface."""
import asyncio

import os

import numpy as np
import pytest

from oracle import flat as oracle
from rag_inference_pipeline_amd import index_io, runtime_factory
from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.retrieval_executor import RetrievalExecutor
from rag_inference_pipeline_amd.schemas import RetrievalRequestItem

pytestmark = pytest.mark.gpu


@pytest.fixture()
def corpus(tmp_path):
    X = oracle.synth_rows(1234, 0, 10_000, 384)  # BASELINE config A: 10k x 384
    path = tmp_path / "faiss_index.bin"
    index_io.write_flat_index(path, X, 0)
    return X, path


def test_faiss_store_contract_and_parity(gpu_required, corpus):
    X, path = corpus
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(path), faiss_dim=384))
    with pytest.raises(RuntimeError, match="not loaded"):
        store.search(np.zeros((1, 384), np.float32), 10)
    store.load()
    store.load()  # idempotent
    assert store.is_loaded and store.index_size == 10_000
    Q = oracle.synth_rows(4321, 0, 1, 384)
    D, I = store.search(Q.astype(np.float64), 10)  # non-fp32 input is converted, as the reference does
    Do, Io = oracle.search(X, Q, 10)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (1, 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert isinstance(I[0].tolist()[0], int)
    with pytest.raises(ValueError, match="2D array"):
        store.search(np.zeros(384, np.float32), 10)
    with pytest.raises(ValueError, match="dimension mismatch"):
        store.search(np.zeros((1, 100), np.float32), 10)
    store.unload()
    assert not store.is_loaded and store.index_size == 0


def test_faiss_only_profile_end_to_end(gpu_required, corpus, tmp_path):
    """configs/retrieval_faiss_only.yaml shape: index only, embeddings come with the request."""
    X, path = corpus
    prof = tmp_path / "retrieval_faiss_only.yaml"
    prof.write_text("---\nname: retrieval_faiss_only\ncomponents:\n  - name: faiss_store\n    type: faiss\n"
                    "routes:\n  - prefix: /retrieve\n    target: retrieval\n")
    settings = PipelineSettings(FAISS_INDEX_PATH=str(path), ROLE_PROFILE_OVERRIDE_PATH=str(prof), faiss_dim=384,
                                retrieval_batch_size=8, retrieval_max_batch_delay_ms=20)
    registry, profile, _ = runtime_factory.build_registry_from_profile(settings)
    assert registry.get("faiss_store").is_loaded

    async def run():
        ex = RetrievalExecutor(registry, settings)
        await ex.start()
        Q = oracle.synth_rows(99, 0, 11, 384)
        outs = await asyncio.gather(*[
            ex.process_request(RetrievalRequestItem(request_id=f"r{i}", query="", embedding=Q[i].tolist()))
            for i in range(11)])
        await ex.stop()
        return Q, outs

    Q, outs = asyncio.run(run())
    Do, Io = oracle.search(X, Q, 10)
    for i, item in enumerate(outs):
        assert item.request_id == f"r{i}"
        assert [d.doc_id for d in item.docs] == Io[i].tolist()
        np.testing.assert_array_equal(np.array([d.score for d in item.docs], np.float32), Do[i])
    registry.unload_all()
    assert not registry.get("faiss_store").is_loaded


def test_binary_embedding_field_through_the_gpu_index(gpu_required, corpus):
    """the optional `embedding_f32` field (little-endian fp32 bytes; base64 text in JSON) carries the
    same query as the reference's list of floats — through the real index on the GPU both forms, mixed in one
    batch, return the oracle's ids and score bits, and the JSON round trip of the request item keeps the bytes."""
    import base64
    X, path = corpus
    settings = PipelineSettings(FAISS_INDEX_PATH=str(path), faiss_dim=384, retrieval_batch_size=16,
                                retrieval_max_batch_delay_ms=20, DISABLE_CACHE_FOR_PROFILING="true")
    store = FAISSStore(settings)
    store.load()
    from rag_inference_pipeline_amd.component_registry import ComponentRegistry
    registry = ComponentRegistry()
    registry.register("faiss_store", store)
    Q = oracle.synth_rows(77, 0, 9, 384)

    async def run():
        ex = RetrievalExecutor(registry, settings)
        await ex.start()
        items = []
        for i in range(9):
            if i % 2 == 0:   # binary form, as it arrives from JSON: base64 text -> bytes by the validator
                raw = RetrievalRequestItem.model_validate_json(
                    '{"request_id": "r%d", "query": "", "embedding_f32": "%s"}' % (i, base64.b64encode(Q[i].tobytes()).decode()))
                assert raw.embedding_f32 == Q[i].tobytes()
                items.append(raw)
            else:
                items.append(RetrievalRequestItem(request_id=f"r{i}", query="", embedding=Q[i].tolist()))
        outs = await asyncio.gather(*[ex.process_request(it) for it in items])
        await ex.stop()
        return outs

    outs = asyncio.run(run())
    Do, Io = oracle.search(X, Q, 10)
    for i, item in enumerate(outs):
        assert item.request_id == f"r{i}" and [d.doc_id for d in item.docs] == Io[i].tolist()
        np.testing.assert_array_equal(np.array([d.score for d in item.docs], np.float32).view(np.uint32), Do[i].view(np.uint32))
    store.unload()


# ---- embedder / reranker / whole retrieval node --------------------------------------------------------

def _hf_checkpoint(tmp_path, head):
    """A tiny BERT checkpoint + WordPiece vocabulary written with `transformers` itself."""
    import torch
    import transformers
    from test_components_cpu import VOCAB, make_tokenizer_dir

    torch.manual_seed(0)
    cfg = transformers.BertConfig(vocab_size=len(VOCAB), hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                                  intermediate_size=128, max_position_embeddings=64, num_labels=1)
    model = (transformers.BertForSequenceClassification(cfg) if head else transformers.BertModel(cfg, add_pooling_layer=False)).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    d = tmp_path / ("ce" if head else "st")
    d.mkdir()
    model.save_pretrained(str(d))
    make_tokenizer_dir(d)
    return str(d), model, transformers.AutoTokenizer.from_pretrained(str(d), local_files_only=True)


def test_embedding_generator_from_local_checkpoint_matches_transformers(gpu_required, tmp_path):
    """The reference's embed path restated with `transformers` on the CPU: tokenizer -> BertModel ->
    mean pooling -> normalize (what SentenceTransformer.encode(normalize_embeddings=True) does)."""
    import torch
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    path, model, tok = _hf_checkpoint(tmp_path, head=False)
    gen = EmbeddingGenerator(PipelineSettings(embedding_model_name=path, DISABLE_CACHE_FOR_PROFILING="true"))
    gen.load()
    texts = ["What is the capital of France?", "paris is a city in france.", "the river is long", "?"]
    emb = gen.encode(texts)
    enc = tok(texts, padding=True, truncation=True, return_tensors="pt")
    with torch.no_grad():
        h = model(**enc).last_hidden_state
    m = enc["attention_mask"][..., None].float()
    want = torch.nn.functional.normalize((h * m).sum(1) / m.sum(1), dim=1).numpy()
    assert emb.shape == (4, 64) and emb.dtype == np.float32
    np.testing.assert_allclose(emb, want, atol=1e-5)
    # cache path returns the same rows and hits on the second call
    gen2 = EmbeddingGenerator(PipelineSettings(embedding_model_name=path, DISABLE_CACHE_FOR_PROFILING="false"))
    gen2.load()
    a, b = gen2.encode(texts), gen2.encode(texts[::-1])
    np.testing.assert_array_equal(a, b[::-1])
    assert gen2.cache.hits >= 4
    gen.unload(); gen2.unload()
    assert not gen.is_loaded


def test_embedder_hands_its_batch_to_the_index_in_device_memory(gpu_required, corpus):
    """encode_device() -> FAISSStore.search: the embeddings stay in HBM, the search runs on the encoder's stream,
    ids and score bits equal encode() -> search() (and so the oracle); concurrent batches on pool threads — the
    scheduler's situation — do not mix their embeddings up."""
    import threading
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    from rag_inference_pipeline_amd.device_embeddings import DeviceEmbeddings
    X, path = corpus
    settings = PipelineSettings(FAISS_INDEX_PATH=str(path), faiss_dim=384, embedding_model_name="synthetic:all-MiniLM-L6-v2",
                                DISABLE_CACHE_FOR_PROFILING="true")
    store, gen = FAISSStore(settings), EmbeddingGenerator(settings)
    store.load(); gen.load()
    texts = [f"query number {i} about topic {i % 7} and some more words" for i in range(32)]
    host = gen.encode(texts)
    dev = gen.encode_device(texts)
    assert isinstance(dev, DeviceEmbeddings) and dev.shape == (32, 384) and dev.ndim == 2
    np.testing.assert_array_equal(dev.numpy().view(np.uint32), host.view(np.uint32))   # the same forward pass
    D1, I1 = store.search(host, 10)
    D2, I2 = store.search(gen.encode_device(texts), 10)
    Do, Io = oracle.search(X, host, 10)
    for D, I in ((D1, I1), (D2, I2)):
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    with pytest.raises(ValueError, match="dimension mismatch"):
        store.search(DeviceEmbeddings(dev._tensor[:, :100].contiguous(), dev.stream, dev.device), 10)
    errors = []

    def client(t):
        try:
            mine = [f"thread {t} text {i} {'word ' * (i % 5)}" for i in range(8 + t)]
            want = oracle.search(X, gen.encode(mine), 10)
            for _ in range(5):
                D, I = store.search(gen.encode_device(mine), 10)
                if not (np.array_equal(I, want[1]) and np.array_equal(D.view(np.uint32), want[0].view(np.uint32))):
                    errors.append(f"thread {t}: results differ")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {t}: {type(exc).__name__}: {exc}")

    pool = [threading.Thread(target=client, args=(t,)) for t in range(4)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    assert not errors, errors
    gen.unload(); store.unload()


def test_reranker_from_local_checkpoint_matches_transformers(gpu_required, tmp_path):
    import torch
    from rag_inference_pipeline_amd.components.reranker import Reranker
    from rag_inference_pipeline_amd.components.schemas import Document
    path, model, tok = _hf_checkpoint(tmp_path, head=True)
    rr = Reranker(PipelineSettings(reranker_model_name=path, truncate_length=32))
    rr.load()
    docs = [Document(doc_id=i, title=f"t{i}", content=c) for i, c in enumerate(
        ["paris is the capital of france.", "the river is long. " * 12, "a city", "what?", "france"])]
    query = "what is the capital of france?"
    got = rr.rerank(query, docs)
    enc = tok([[query, d.content] for d in docs], padding=True, truncation=True, return_tensors="pt", max_length=32)
    with torch.no_grad():
        want = torch.sigmoid(model(**enc).logits.view(-1).float()).numpy()
    order = sorted(range(len(docs)), key=lambda i: want[i], reverse=True)
    assert [d.doc_id for d in got] == order
    np.testing.assert_allclose([d.score for d in got], want[order], atol=1e-5)
    assert [d.doc_id for d in rr.rerank(query, docs, top_n=2)] == order[:2]
    both = rr.rerank_batch([query, "a city in france"], [docs, docs[:3]])
    assert [d.doc_id for d in both[0]] == order and len(both[1]) == 3
    rr.unload()


def test_retriever_with_rerank_profile_end_to_end(gpu_required, tmp_path):
    """configs/retriever_with_rerank.yaml shape on synthetic models: text in, reranked documents out;
    every stage checked against the oracle restatement of the same stage."""
    import sqlite3
    from oracle import bert as obert
    from rag_inference_pipeline_amd.model_source import resolve_model

    n_docs, d = 2000, 384
    words = ["alpha", "beta", "gamma", "delta", "river", "city", "france", "paris", "long", "capital", "model", "index"]
    rng = np.random.default_rng(0)
    contents = [" ".join(rng.choice(words, size=int(rng.integers(8, 25)))) for _ in range(n_docs)]
    emb_name, rr_name = "synthetic:all-MiniLM-L6-v2:7", "synthetic:ms-marco-MiniLM-L-6-v2:8"
    ecfg, ew, etok, emax = resolve_model(emb_name, "embedding")
    # corpus embeddings come from the ORACLE encoder (index building is not the path under test)
    ids, types = etok.encode_batch(contents, emax)
    X = np.concatenate([obert.embed(ecfg, ew, ids[i:i + 250], types[i:i + 250]) for i in range(0, n_docs, 250)])
    index_path = tmp_path / "faiss_index.bin"
    index_io.write_flat_index(index_path, X, 0)
    docs_dir = tmp_path / "documents"
    docs_dir.mkdir()
    con = sqlite3.connect(docs_dir / "documents.db")
    con.execute("CREATE TABLE documents (doc_id INTEGER PRIMARY KEY, title TEXT, content TEXT, category TEXT)")
    con.executemany("INSERT INTO documents VALUES (?,?,?,?)", [(i, f"Doc {i}", c, "cat") for i, c in enumerate(contents)])
    con.commit(); con.close()
    prof = tmp_path / "retriever_with_rerank.yaml"
    prof.write_text("---\nname: retrieval_with_rerank\ncomponents:\n  - name: embedding_generator\n    type: embedding\n"
                    "  - name: faiss_store\n    type: faiss\n  - name: document_store\n    type: document_store\n"
                    "  - name: reranker\n    type: reranker\nroutes:\n  - prefix: /retrieve\n    target: retrieval\n")
    settings = PipelineSettings(FAISS_INDEX_PATH=str(index_path), DOCUMENTS_DIR=str(docs_dir), faiss_dim=d,
                                ROLE_PROFILE_OVERRIDE_PATH=str(prof), DOCUMENTS_PAYLOAD_MODE="full",
                                DISABLE_CACHE_FOR_PROFILING="true", embedding_model_name=emb_name,
                                reranker_model_name=rr_name, retrieval_k=10, retrieval_batch_size=8,
                                retrieval_max_batch_delay_ms=20)
    registry, _, _ = runtime_factory.build_registry_from_profile(settings)
    queries = ["capital of france", "long river city", "model index alpha beta", "paris", "gamma delta gamma"]

    async def run():
        ex = RetrievalExecutor(registry, settings)
        await ex.start()
        outs = await asyncio.gather(*[ex.process_request(RetrievalRequestItem(request_id=f"r{i}", query=q))
                                      for i, q in enumerate(queries)])
        await ex.stop()
        return outs

    outs = asyncio.run(run())
    # oracle pipeline: embed -> search -> rerank
    qids, qtypes = etok.encode_batch(queries, emax)
    Qo = obert.embed(ecfg, ew, qids, qtypes)
    Do, Io = oracle.search(X, Qo, 10)
    rcfg, rw, rtok, rmax = resolve_model(rr_name, "reranker")
    for qi, item in enumerate(outs):
        got_ids = [d.doc_id for d in item.docs]
        s64 = X.astype(np.float64) @ Qo[qi].astype(np.float64)
        gap = np.sort(s64)[::-1][9] - np.sort(s64)[::-1][10]
        if gap > 1e-5:  # away from a near-tie at the k boundary the retrieved SET must be identical
            assert sorted(got_ids) == sorted(Io[qi].tolist())
        pids, ptypes = rtok.encode_pairs([queries[qi]] * 10, [contents[i] for i in Io[qi]], min(512, rmax))
        want = obert.classify(rcfg, rw, pids, ptypes)[:, 0]
        by_id = {int(i): float(s) for i, s in zip(Io[qi], want)}
        for ddoc in item.docs:
            if ddoc.doc_id in by_id:
                assert ddoc.score == pytest.approx(by_id[ddoc.doc_id], abs=2e-5)
            assert ddoc.content == contents[ddoc.doc_id][:512] and ddoc.title == f"Doc {ddoc.doc_id}"
        scores = [dd.score for dd in item.docs]
        assert scores == sorted(scores, reverse=True) and len(item.docs) == 10
    asyncio.run(registry.stop_all())
    registry.unload_all()


def test_build_index_tool_embeds_documents_and_round_trips(gpu_required, tmp_path):
    """tools/build_index: documents -> encoder -> raw fp32 + sidecar -> FAISSStore; a document's own
    text retrieves it first."""
    import sqlite3
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    from rag_inference_pipeline_amd.tools.build_index import build_index
    docs_dir = tmp_path / "documents"
    docs_dir.mkdir()
    texts = [f"document number {i} talks about topic {i % 7} and item {i * 13 % 101}" for i in range(300)]
    con = sqlite3.connect(docs_dir / "documents.db")
    con.execute("CREATE TABLE documents (doc_id INTEGER PRIMARY KEY, title TEXT, content TEXT, category TEXT)")
    con.executemany("INSERT INTO documents VALUES (?,?,?,?)", [(i, f"t{i}", t, "c") for i, t in enumerate(texts)])
    con.commit(); con.close()
    out = tmp_path / "faiss_index.f32"
    model = "synthetic:all-MiniLM-L6-v2:3"
    n, d = build_index(str(docs_dir), model, str(out), batch_docs=128)
    assert (n, d) == (300, 384) and out.stat().st_size == 300 * 384 * 4
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(out), faiss_dim=384))
    store.load()
    gen = EmbeddingGenerator(PipelineSettings(embedding_model_name=model, DISABLE_CACHE_FOR_PROFILING="true"))
    gen.load()
    probe = [5, 77, 299]
    D, I = store.search(gen.encode([texts[i] for i in probe]), 3)
    assert I[:, 0].tolist() == probe and np.allclose(D[:, 0], 1.0, atol=1e-5)
    store.unload(); gen.unload()
    # the same build as three processes sharing the GPU (one per "rank"): the same file
    import subprocess
    import sys
    out3 = tmp_path / "faiss_index_3ranks.f32"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # leftovers of a crashed earlier build of the same file: a marker of another build and a stale sidecar
    (tmp_path / "faiss_index_3ranks.f32.part1.done").write_text('{"build_id": "crashed-build", "rank": 1, "rows": 100}')
    (tmp_path / "faiss_index_3ranks.f32.json").write_text('{"d": 384, "ntotal": 1, "metric": "ip"}')
    procs = [subprocess.Popen([sys.executable, "-m", "rag_inference_pipeline_amd.tools.build_index", "--documents-dir",
                               str(docs_dir), "--model", model, "--out", str(out3), "--batch-docs", "64",
                               "--rank", str(r), "--world", "3", "--device", "0", "--build-id", "test-build-7"], cwd=root)
             for r in (2, 1, 0)]   # rank 0 last: the others may well finish before it has started
    try:
        assert all(p.wait(timeout=300) == 0 for p in procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a3, a1 = np.fromfile(out3, dtype=np.float32), np.fromfile(out, dtype=np.float32)
    assert a3.shape == a1.shape and np.abs(a3 - a1).max() < 2e-6   # other batch shapes, other GEMM paths: fp32 rounding
    import json
    side = json.loads((tmp_path / "faiss_index_3ranks.f32.json").read_text())
    assert side["ntotal"] == 300 and side["d"] == 384 and side["ranks"] == 3 and side["build_id"] == "test-build-7"
    with pytest.raises(ValueError, match="build id"):
        build_index(str(docs_dir), model, str(tmp_path / "x.f32"), rank=1, world=2)
    assert not list(tmp_path.glob("*.done"))


def test_a_share_of_the_chip_for_each_stage_changes_no_result(gpu_required):
    """rag_stream_create_masked + rag_*_set_cu_budget (include/rag_amd.h): the scan on 224 CUs and the encoder on 32
    return exactly what they return on the whole device."""
    import torch
    from oracle import flat as oracle
    from rag_inference_pipeline_amd.bert import BertConfig, BertModel, pack_sequences, random_weights
    from rag_inference_pipeline_amd import _native
    from rag_inference_pipeline_amd.flat_index import FlatIndex, create_masked_stream, destroy_stream

    total = torch.cuda.get_device_properties(0).multi_processor_count
    if total < 64:
        pytest.skip("needs a device with at least 64 compute units")
    d, nq, k = 384, 32, 10
    idx = FlatIndex(d); idx.add_synthetic(300_000, 7)
    Q = torch.from_numpy(oracle.synth_rows(99, 0, nq, d)).cuda()
    s0 = torch.empty((nq, k), dtype=torch.float32, device="cuda"); i0 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    s1, i1 = torch.empty_like(s0), torch.empty_like(i0)
    idx.search_device(Q.data_ptr(), nq, k, s0.data_ptr(), i0.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    scan_st = create_masked_stream(0, 32, total - 32)
    enc_st = create_masked_stream(0, 0, 32)
    try:
        idx.set_cu_budget(total - 32)
        idx.search_device(Q.data_ptr(), nq, k, s1.data_ptr(), i1.data_ptr(), scan_st)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
        idx.set_cu_budget(0)

        cfg = BertConfig.minilm_l6(); cfg.vocab_size = 2000; cfg.n_layers = 2
        model = BertModel(cfg, random_weights(cfg, 3))
        rng = np.random.default_rng(3)
        seqs = [rng.integers(3, cfg.vocab_size, size=int(n)).tolist() for n in rng.integers(8, 21, size=32)]
        ids, _, cu = pack_sequences(seqs)
        ids_t, cu_t = torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda()
        e0 = torch.empty((32, cfg.hidden), dtype=torch.float32, device="cuda"); e1 = torch.empty_like(e0)
        args = (ids_t.data_ptr(), 0, cu_t.data_ptr(), 32, int(cu[-1]), int(max(map(len, seqs))), _native.BERT_OUT_MEAN, True)
        model.forward_device(*args, e0.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        model.set_cu_budget(32)
        model.forward_device(*args, e1.data_ptr(), enc_st)
        torch.cuda.synchronize()
        # (the CU budget may change a split-K count, i.e. a summation order: equal to fp32 rounding, not bit for bit)
        np.testing.assert_allclose(e1.cpu().numpy(), e0.cpu().numpy(), atol=2e-6)
        model.close()
    finally:
        destroy_stream(0, scan_st); destroy_stream(0, enc_st)
        idx.close()
    with pytest.raises(RuntimeError):
        create_masked_stream(0, total - 8, 16)      # outside the device


def test_encoder_share_of_the_chip_in_the_product_path(gpu_required, tmp_path):
    """settings.encoder_cus (RAG_AMD_ENCODER_CUS): the embedder's passes run on a stream that owns 32 CUs, searches that
    follow them on the device on a stream that owns the rest; two batches in flight on pool threads.  Same documents,
    scores to fp32 rounding (the CU budget may change a split-K choice in the encoder), as the default deployment."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from rag_inference_pipeline_amd.batch_scheduler import Batch
    from rag_inference_pipeline_amd.schemas import PendingRequest
    from rag_inference_pipeline_amd.component_registry import ComponentRegistry
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16

    if torch.cuda.get_device_properties(0).multi_processor_count < 64:
        pytest.skip("needs a device with at least 64 compute units")

    def build(share):
        settings = PipelineSettings(DISABLE_CACHE_FOR_PROFILING="true", faiss_dim=384, retrieval_k=10,
                                    embedding_model_name="synthetic:all-MiniLM-L6-v2:3", RAG_AMD_ENCODER_CUS=share)
        store = FAISSStore(settings)
        idx = FlatIndex(384); idx.add_synthetic(200_000, 11); idx.set_screening(SCREEN_FP16)
        store._index, store._ntotal, store._is_loaded = idx, 200_000, True
        reg = ComponentRegistry()
        emb = EmbeddingGenerator(settings); reg.register("embedding_generator", emb, emb.load)
        reg.register("faiss_store", store)
        return RetrievalExecutor(reg, settings), emb, store

    rng = np.random.default_rng(5)
    words = "vector index query document embedding transformer attention memory kernel latency shard merge score".split()
    batches = [[" ".join(rng.choice(words, size=int(rng.integers(5, 14)))) for _ in range(32)] for _ in range(6)]
    mk = lambda qs, b: Batch(b, [PendingRequest(request_id=f"b{b}r{i}", query=q, timestamp=0.0) for i, q in enumerate(qs)])
    ex0, emb0, store0 = build(0)
    want = [ex0._process_batch_sync(mk(qs, b)) for b, qs in enumerate(batches)]
    ex1, emb1, store1 = build(32)
    with ThreadPoolExecutor(max_workers=2) as pool:
        got = list(pool.map(lambda a: ex1._process_batch_sync(mk(a[1], a[0])), enumerate(batches)))
    assert emb1._share_stream and store1._share_stream          # both streams exist: the partition was really used
    for wb, gb in zip(want, got):
        for w, g in zip(wb, gb):
            assert [d.doc_id for d in g.docs] == [d.doc_id for d in w.docs]
            np.testing.assert_allclose([d.score for d in g.docs], [d.score for d in w.docs], atol=2e-6)
    for e, s in ((emb0, store0), (emb1, store1)):
        e.unload(); s.unload()
