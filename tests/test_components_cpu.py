"""CPU: component contracts that need no GPU (error messages pinned by the reference's tests),
tokenisation, model-name resolution."""
import numpy as np
import pytest

from rag_inference_pipeline_amd.cache import LRUCache
from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
from rag_inference_pipeline_amd.components.reranker import Reranker
from rag_inference_pipeline_amd.components.schemas import Document
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.model_source import HashTokenizer, HFTokenizer, resolve_model

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "what", "is", "the", "capital", "of", "france", "paris",
         "a", "city", "in", ".", "?", "##s", "river", "long"]


def make_tokenizer_dir(path):
    (path / "vocab.txt").write_text("\n".join(VOCAB) + "\n")
    (path / "tokenizer_config.json").write_text('{"tokenizer_class": "BertTokenizer", "do_lower_case": true}')
    return str(path)


def test_embedding_generator_contract_before_load():
    g = EmbeddingGenerator(PipelineSettings(embedding_model_name="synthetic:all-MiniLM-L6-v2"))
    assert g.is_loaded is False and "not loaded" in repr(g)
    with pytest.raises(RuntimeError, match="not loaded"):
        g.encode(["x"])
    g._is_loaded, g._model = True, object()
    with pytest.raises(ValueError, match="empty"):
        g.encode([])
    g._is_loaded, g._model = False, None
    g.unload()
    g.clear_cache()


def test_reranker_contract_before_load_and_length_check():
    r = Reranker(PipelineSettings(reranker_model_name="synthetic:ms-marco-MiniLM-L-6-v2"))
    assert r.is_loaded is False
    with pytest.raises(RuntimeError, match="not loaded"):
        r.rerank("q", [Document(doc_id=1, title="t", content="c")])
    with pytest.raises(RuntimeError, match="not loaded"):
        r.rerank_batch(["q"], [[]])
    r._loaded, r.model, r.tokenizer = True, object(), object()
    assert r.rerank("q", []) == []
    with pytest.raises(ValueError, match="must have same length"):
        r.rerank_batch(["a", "b"], [[]])
    assert r.rerank_batch(["a"], [[]]) == [[]]


def test_ranked_is_stable_and_top_n():
    docs = [Document(doc_id=i, title="", content=str(i)) for i in range(4)]
    out = Reranker._ranked(docs, [0.2, 0.9, 0.2, 0.5], None)
    assert [d.doc_id for d in out] == [1, 3, 0, 2] and out[0].score == pytest.approx(0.9)
    assert [d.doc_id for d in Reranker._ranked(docs, [0.2, 0.9, 0.2, 0.5], 2)] == [1, 3]


def test_resolve_model_never_downloads_and_knows_presets():
    with pytest.raises(RuntimeError, match="never downloads"):
        resolve_model("BAAI/definitely-not-cached-model", "embedding")
    with pytest.raises(ValueError, match="unknown synthetic architecture"):
        resolve_model("synthetic:nope", "embedding")
    with pytest.raises(ValueError, match="no classifier head"):
        resolve_model("synthetic:all-MiniLM-L6-v2", "reranker")
    cfg, w, tok, max_len = resolve_model("synthetic:bge-reranker-base:3", "reranker")
    assert (cfg.hidden, cfg.pos_offset, cfg.head, cfg.vocab_size) == (768, 2, "roberta", 250002)
    assert w["word_emb"].shape == (250002, 768) and max_len == 512 and tok.roberta


def test_hash_tokenizer_framing_and_truncation():
    t = HashTokenizer(30522)
    ids, types = t.encode_batch(["Hello, world!", ""], 16)
    assert ids[0][0] == 101 and ids[0][-1] == 102 and len(ids[0]) == 6 and ids[1] == [101, 102]
    assert t.encode_batch(["hello"], 8)[0] == t.encode_batch(["HELLO"], 8)[0]
    ids, types = t.encode_pairs(["what is x"], ["x is a thing . " * 10], 16)
    assert len(ids[0]) == 16 and types[0] == [0] * 5 + [1] * 11 and ids[0].count(102) == 2
    r = HashTokenizer(250002, roberta=True)
    ids, types = r.encode_pairs(["a b"], ["c"], 32)
    assert ids[0][0] == 0 and ids[0][3:5] == [2, 2] and ids[0][-1] == 2 and set(types[0]) == {0}


def test_hf_tokenizer_from_local_files_matches_reference_call(tmp_path):
    transformers = pytest.importorskip("transformers")
    path = make_tokenizer_dir(tmp_path)
    tok = HFTokenizer(path)
    ref = transformers.AutoTokenizer.from_pretrained(path, local_files_only=True)
    texts = ["What is the capital of France?", "Paris."]
    ids, types = tok.encode_batch(texts, 16)
    assert ids == ref(texts, truncation=True, max_length=16)["input_ids"]
    assert ids[0][0] == 2 and ids[0][-1] == 3
    q, d = ["what is the capital of france ?"] * 2, ["paris is a city in france .", "the river is long . " * 20]
    pids, ptypes = tok.encode_pairs(q, d, 24)
    enc = ref([[a, b] for a, b in zip(q, d)], padding=True, truncation=True, max_length=24)  # reranker.py:240-246
    for i in range(2):
        n = sum(enc["attention_mask"][i])
        assert pids[i] == enc["input_ids"][i][:n] and ptypes[i] == enc["token_type_ids"][i][:n]
    assert len(pids[1]) == 24


def test_lru_cache_ttl_and_eviction():
    c = LRUCache(capacity=2, ttl=None)
    c.put("a", 1); c.put("b", 2); assert c.get("a") == 1
    c.put("c", 3)
    assert c.get("b") is None and c.get("a") == 1 and c.get("c") == 3 and len(c) == 2
    c.clear(); assert len(c) == 0
    c = LRUCache(capacity=2, ttl=0.0)
    c.put("x", np.ones(2))
    import time; time.sleep(0.01)
    assert c.get("x") is None


def test_one_device_rule_for_every_component(monkeypatch):
    """index, encoder and cross-encoder must agree on the GPU.  Alone: RAG_AMD_DEVICE.  One
    process per GPU (initialised group, world > 1): LOCAL_RANK, else torch's current device."""
    import torch
    import torch.distributed as dist

    from rag_inference_pipeline_amd.config import PipelineSettings, resolve_gpu_device
    from rag_inference_pipeline_amd.components.embedding import EmbeddingGenerator
    from rag_inference_pipeline_amd.components.reranker import Reranker

    s = PipelineSettings(RAG_AMD_DEVICE=3)
    assert resolve_gpu_device(s) == 3 and resolve_gpu_device(None) == 0
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 8)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 5)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert resolve_gpu_device(s) == 5
    monkeypatch.setenv("LOCAL_RANK", "6")
    assert resolve_gpu_device(s) == 6
    assert EmbeddingGenerator(s).device == "cuda:6" and Reranker(s).device == "cuda:6"
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 1)   # a group of one is the plain deployment
    assert resolve_gpu_device(s) == 3


def test_document_store_fetches_a_batch_in_one_pass(tmp_path):
    """fetch_documents_batch (reference document_store.py:278-302): per query in the requested order, unknown
    ids silently dropped (:276), title/content cut to truncate_length characters (:59-84); ids shared by
    several queries are read once."""
    import sqlite3

    from rag_inference_pipeline_amd.components.document_store import DocumentStore
    from rag_inference_pipeline_amd.config import PipelineSettings
    d = tmp_path / "documents"
    d.mkdir()
    con = sqlite3.connect(d / "documents.db")
    con.execute("CREATE TABLE documents (doc_id INTEGER PRIMARY KEY, title TEXT, content TEXT, category TEXT)")
    con.executemany("INSERT INTO documents VALUES (?,?,?,?)",
                    [(i, f"title {i}", f"content of document {i} " * 4, None if i % 2 else "cat") for i in range(2500)])
    con.commit(); con.close()
    store = DocumentStore(PipelineSettings(DOCUMENTS_DIR=str(d)))
    batch = [[5, 3, 99999, 3, 2499], [], [3, 7], list(range(2000, 100, -1))]     # > 900 distinct ids: several IN (...) passes
    got = store.fetch_documents_batch(batch, truncate_length=12)
    assert [[x.doc_id for x in docs] for docs in got] == [[5, 3, 3, 2499], [], [3, 7], list(range(2000, 100, -1))]
    assert got[0][0].title == "title 5" and got[0][0].content == "content of d" and got[0][0].category is None
    assert got[3][0].category == "cat" and got[2][0].category is None and len(got[3][0].content) == 12
    full = store.fetch_documents([7, 123456, 1])
    assert [x.doc_id for x in full] == [7, 1] and full[0].content == "content of document 7 " * 4
    assert store.fetch_documents([]) == [] and store.fetch_documents_batch([]) == []
    store.close_all()


def test_fast_constructor_builds_what_validation_builds():
    """components/schemas.fast_constructor: the objects a rerank batch returns are ordinary model instances —
    equal to validated ones, dumpable, serialisable, accepted by a validating parent model."""
    from rag_inference_pipeline_amd.components.schemas import RerankedDocument, fast_constructor
    from rag_inference_pipeline_amd.schemas import RetrievalDocument, RetrievalResponseItem
    for cls in (RerankedDocument, RetrievalDocument):
        make = fast_constructor(cls)
        a = make(doc_id=7, title="t", content="c", category="", score=0.25)
        b = cls(doc_id=7, title="t", content="c", score=0.25)
        assert a == b and a.model_dump() == b.model_dump() and a.model_dump_json() == b.model_dump_json()
        assert a.model_fields_set == {"doc_id", "title", "content", "category", "score"} and a.model_copy() == b
    item = RetrievalResponseItem(request_id="r", docs=[fast_constructor(RetrievalDocument)(
        doc_id=1, title="", content="", category="", score=1.0)], compressed_docs=None)
    assert item.docs[0].doc_id == 1 and '"score":1.0' in item.model_dump_json()


def test_build_index_does_not_take_torchruns_default_run_id_as_a_build_id(monkeypatch, tmp_path):
    """torchrun's static rendezvous exports TORCHELASTIC_RUN_ID=none for EVERY run: accepted as a build id it
    would make the .done markers of a crashed earlier build indistinguishable from this build's."""
    from rag_inference_pipeline_amd.tools.build_index import build_index

    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    with pytest.raises(ValueError, match="build id"):
        build_index(str(tmp_path), "synthetic:all-MiniLM-L6-v2", str(tmp_path / "x.f32"), rank=1, world=2)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-4711")   # a real id is used: the call gets past the check
    with pytest.raises(FileNotFoundError, match="Document database"):
        build_index(str(tmp_path), "synthetic:all-MiniLM-L6-v2", str(tmp_path / "x.f32"), rank=1, world=2)


def test_native_pair_encoder_equals_the_python_stand_in_tokenizer():
    """rag_hash_encode_pairs (csrc/rag_text.cpp) against HashTokenizer.encode_pairs + pack_sequences: same ids, token
    types and sequence boundaries for ASCII text — punctuation, control characters, every whitespace Python's `\\s`
    knows, empty sides, truncation down to nothing, BERT and RoBERTa framing; non-ASCII text is declined (None)."""
    from rag_inference_pipeline_amd.bert import pack_sequences

    rng = np.random.default_rng(3)
    alphabet = [chr(c) for c in range(1, 128)]
    words = ["Retrieval", "augmented_generation", "x1", "don't", "3.14", "(a-b)", "GPU", "\x1c", "\x0b", "tab\there", "_"]

    def text():
        parts = []
        for _ in range(int(rng.integers(0, 40))):
            parts.append(str(rng.choice(words)) if rng.random() < 0.7 else "".join(rng.choice(alphabet, size=int(rng.integers(1, 6)))))
        return str(rng.choice([" ", "  ", "\n", "\t "])).join(parts)

    for roberta, vocab in ((False, 30522), (True, 250002), (False, 500)):
        tok = HashTokenizer(vocab, roberta=roberta)
        first, second = [text() for _ in range(120)], [text() for _ in range(120)]
        first[3], second[4], second[3] = "", "", ""
        for max_len in (512, 40, 7, 4, 3, 0):
            ids, types = tok.encode_pairs(first, second, max_len)
            want = pack_sequences(ids, types)
            got = tok.encode_pairs_packed(first, second, max_len, True)
            assert got is not None
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w)
            assert tok.encode_pairs_packed(first, second, max_len, False)[1] is None
    assert HashTokenizer(30522).encode_pairs_packed(["café"], ["x"], 16) is None


def test_search_share_of_the_chip_is_set_up_once_and_only_on_one_gpu(monkeypatch):
    """settings.encoder_cus (RAG_AMD_ENCODER_CUS): FAISSStore creates its search stream over the CUs the encoder does not
    own at the first device hand-off, once, tells the index how many CUs to plan for, and releases the stream at unload;
    nothing happens when the setting is off, when it is not smaller than the device, or in a sharded deployment."""
    from rag_inference_pipeline_amd import flat_index
    from rag_inference_pipeline_amd.components.faiss_store import FAISSStore

    made, destroyed = [], []
    monkeypatch.setattr(flat_index, "create_masked_stream", lambda dev, first, n: made.append((dev, first, n)) or 4242)
    monkeypatch.setattr(flat_index, "destroy_stream", lambda dev, st: destroyed.append((dev, st)))
    monkeypatch.setattr(flat_index, "device_cu_count", lambda dev: 256)

    class Index:
        device = 3
        budget = None
        closed = False
        def set_cu_budget(self, n): self.budget = n
        def close(self): self.closed = True

    def store(**env):
        s = FAISSStore(PipelineSettings(**env))
        s._index, s._ntotal, s._is_loaded = Index(), 10, True
        return s

    s = store(RAG_AMD_ENCODER_CUS=64)
    assert s._search_share() == 4242 and s._search_share() == 4242
    assert made == [(3, 64, 192)] and s._index.budget == 192          # once; the CUs after the encoder's 64
    idx = s._index
    s.unload()
    assert destroyed == [(3, 4242)] and idx.closed and s._share_stream == 0
    for env in ({}, {"RAG_AMD_ENCODER_CUS": 0}, {"RAG_AMD_ENCODER_CUS": 256}):
        s = store(**env)
        assert s._search_share() == 0 and s._index.budget is None
    s = store(RAG_AMD_ENCODER_CUS=64)
    s._sharded = object()                                              # one process per GPU: the ranks do not partition
    assert s._search_share() == 0
    assert made == [(3, 64, 192)]
