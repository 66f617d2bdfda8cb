"""CPU: plugin surface, profiles, scheduler and executor semantics (host logic, no GPU)."""
import asyncio
import glob
import os
import textwrap
import time

import importlib.util

import numpy as np
import pytest

from rag_inference_pipeline_amd import component_factory, runtime_factory
from rag_inference_pipeline_amd.batch_scheduler import AdaptiveBatchPolicy, Batch, BatchScheduler
from rag_inference_pipeline_amd.component_registry import ComponentRegistry
from rag_inference_pipeline_amd.config import PipelineSettings
from rag_inference_pipeline_amd.enums import ComponentType
from rag_inference_pipeline_amd.profile_schema import ProfileFile, load_profile_file
from rag_inference_pipeline_amd.retrieval_executor import RetrievalExecutor
from rag_inference_pipeline_amd.schemas import PendingRequest, RetrievalRequestItem

REF_CONFIGS = "/root/reference/configs"


# ---- settings -----------------------------------------------------------------------------------

def test_settings_alias_fields_read_uppercase_env_and_plain_fields_do_not():
    s = PipelineSettings.from_env({"FAISS_INDEX_PATH": "/x/idx.bin", "ONLY_CPU": "false",
                                   "RETRIEVAL_BATCH_SIZE": "8", "retrieval_k": "5"})
    assert s.faiss_index_path == "/x/idx.bin" and s.only_cpu is False
    assert s.retrieval_batch_size == 32  # un-aliased: the upper-case variable is ignored, as in the reference
    assert s.retrieval_k == 5
    assert s.faiss_dim == 768 and s.retrieval_max_batch_delay_ms == 50 and s.truncate_length == 512


# ---- registry -------------------------------------------------------------------------------------

class _Comp:
    def __init__(self, fail=False):
        self.events, self.fail = [], fail

    def load(self):
        if self.fail:
            raise RuntimeError("boom")
        self.events.append("load")

    def start(self):
        self.events.append("start")

    async def stop(self):
        self.events.append("stop")

    def unload(self):
        self.events.append("unload")


def test_registry_lifecycle_order_aliases_and_failures():
    reg = ComponentRegistry()
    a, b = _Comp(), _Comp()
    reg.register("a", a, a.load, a.start, a.stop, a.unload)
    reg.register("b", b, b.load, b.start, b.stop, b.unload)
    assert a.events == ["load"]  # load ran inside register
    reg.register_alias("faiss_store", "a")
    assert reg.get("faiss_store") is a and reg.get("nope") is None
    with pytest.raises(ValueError, match="already registered"):
        reg.register("a", _Comp())
    with pytest.raises(ValueError, match="conflicts"):
        reg.register_alias("b", "a")
    with pytest.raises(ValueError, match="already registered to"):
        reg.register_alias("faiss_store", "b")
    bad = _Comp(fail=True)
    with pytest.raises(RuntimeError, match="boom"):
        reg.register("bad", bad, bad.load)
    assert "bad" not in reg.components
    order = []
    a.start = lambda: order.append("a")
    b.start = lambda: order.append("b")
    reg._lifecycle_hooks["a"]["start"], reg._lifecycle_hooks["b"]["start"] = a.start, b.start
    asyncio.run(reg.start_all())
    assert order == ["a", "b"]
    asyncio.run(reg.stop_all())
    reg.unload_all()
    assert a.events[-2:] == ["stop", "unload"] and b.events[-2:] == ["stop", "unload"]
    reg.unregister("a")
    assert reg.get("faiss_store") is None


# ---- factory --------------------------------------------------------------------------------------

def test_factory_resolution_rules():
    s = PipelineSettings()
    made = []
    component_factory.register_factory("unit_test_type", lambda st, cfg: made.append((st, cfg)) or "X")
    assert component_factory.create_component("unit_test_type", s, {"a": 1}) == "X" and made[0][1] == {"a": 1}
    with pytest.raises(ValueError, match="Unknown component type: nope"):
        component_factory.create_component("nope", s)
    for key in (ComponentType.FAISS, "faiss", "faiss_store"):
        store = component_factory.create_component(key, s)
        assert type(store).__name__ == "FAISSStore" and store.is_loaded is False
    with pytest.raises(ValueError, match="Unknown component type"):
        component_factory.create_component("llm", s)  # outside the accelerated path unless registered


# ---- profiles ---------------------------------------------------------------------------------------

PROFILE = textwrap.dedent("""\
    ---
    name: t_retrieval
    description: test profile
    components:
      - name: my_index
        type: faiss
        aliases: [vectors]
      - name: reranker
        type: reranker
    routes:
      - prefix: /retrieve
        target: retrieval
        component_aliases: {the_index: my_index}
    """)


def test_profile_schema_validation(tmp_path):
    p = tmp_path / "p.yaml"
    p.write_text(PROFILE)
    prof = load_profile_file(p)
    assert prof.name == "t_retrieval" and prof.components[0].aliases == ["vectors"] and prof.batch_size is None
    with pytest.raises(ValueError, match="Duplicate prefixes"):
        ProfileFile(name="x", routes=[{"target": "retrieval", "prefix": "/a"}, {"target": "gateway", "prefix": "/a"}])
    with pytest.raises(ValueError, match="unknown component"):
        ProfileFile(name="x", components=[{"name": "a", "type": "faiss"}],
                    routes=[{"target": "retrieval", "component_aliases": {"z": "missing"}}])
    with pytest.raises(ValueError, match="Invalid role profile file"):
        bad = tmp_path / "bad.yaml"
        bad.write_text("name: [unclosed")
        load_profile_file(bad)


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference checkout not present")
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(REF_CONFIGS, "*.yaml"))) or ["<none>"])
def test_every_reference_profile_parses_unchanged(path):
    prof = load_profile_file(path)
    assert prof.name and all(c.name and c.type for c in prof.components)


def test_registry_from_profile_registers_default_aliases(tmp_path, monkeypatch):
    made = {}

    class Fake:
        def __init__(self, kind):
            self.kind, self.loaded = kind, False

        def load(self):
            self.loaded = True

        is_loaded = property(lambda self: self.loaded)

    monkeypatch.setitem(component_factory.COMPONENT_FACTORIES, ComponentType.FAISS,
                        lambda s, c: made.setdefault("faiss", Fake("faiss")))
    monkeypatch.setitem(component_factory.COMPONENT_FACTORIES, ComponentType.RERANKER,
                        lambda s, c: made.setdefault("rr", Fake("rr")))
    p = tmp_path / "p.yaml"
    p.write_text(PROFILE)
    s = PipelineSettings(ROLE_PROFILE_OVERRIDE_PATH=str(p))
    reg, prof, aliases = runtime_factory.build_registry_from_profile(s)
    assert reg.get("my_index") is made["faiss"] and made["faiss"].loaded
    assert reg.get("faiss_store") is made["faiss"]      # default alias for the type
    assert reg.get("vectors") is made["faiss"] and reg.get("the_index") is made["faiss"]
    assert reg.get("reranker") is made["rr"] and "reranker" not in aliases  # name == default alias
    with pytest.raises(ValueError, match="No valid role profile"):
        runtime_factory.load_role_profile(PipelineSettings(PIPELINE_ROLE_PROFILE="does_not_exist"), tmp_path)


# ---- scheduler --------------------------------------------------------------------------------------

def _req(i, emb=None):
    return PendingRequest(request_id=f"r{i}", query=f"q{i}", embedding=emb, timestamp=time.time())


def test_adaptive_policy_bounds_and_trend():
    pol = AdaptiveBatchPolicy(max_batch_size=8, min_delay_sec=0.01, max_delay_sec=0.2)
    lo = [pol.update(0) for _ in range(20)][-1]
    hi = [pol.update(16) for _ in range(40)][-1]
    assert lo == pytest.approx(0.01) and hi == pytest.approx(0.2, rel=1e-3)
    d1 = AdaptiveBatchPolicy(8, 0.01, 0.2).update(4)  # one EWMA step from min toward the midpoint
    assert d1 == pytest.approx(0.7 * 0.01 + 0.3 * (0.01 + 0.5 * 0.19))


def test_scheduler_flushes_on_size_and_timeout_and_shutdown():
    async def run():
        seen = []

        async def fn(batch: Batch):
            seen.append(len(batch))
            return [r.request_id for r in batch.requests]

        sch = BatchScheduler(batch_size=4, max_batch_delay_ms=30, process_batch_fn=fn)
        with pytest.raises(RuntimeError, match="not running"):
            await sch.enqueue(_req(0))
        await sch.start()
        out = await asyncio.gather(*[sch.enqueue(_req(i)) for i in range(4)])
        assert out == ["r0", "r1", "r2", "r3"] and seen == [4]
        t0 = time.perf_counter()
        out = await asyncio.gather(sch.enqueue(_req(10)), sch.enqueue(_req(11)))
        assert out == ["r10", "r11"] and seen == [4, 2] and time.perf_counter() - t0 >= 0.025
        pending = asyncio.create_task(sch.enqueue(_req(20)))
        await asyncio.sleep(0)
        await sch.stop()
        assert await pending == "r20" and seen == [4, 2, 1]

    asyncio.run(run())


def test_scheduler_error_conventions():
    async def run():
        async def boom(batch):
            raise KeyError("x")

        sch = BatchScheduler(2, 10, boom)
        await sch.start()
        res = await asyncio.gather(sch.enqueue(_req(0)), sch.enqueue(_req(1)), return_exceptions=True)
        assert all(isinstance(r, RuntimeError) and str(r) == "Batch processing failed" for r in res)

        async def short(batch):
            return ["only one"]

        sch = BatchScheduler(2, 10, short)
        await sch.start()
        res = await asyncio.gather(sch.enqueue(_req(0)), sch.enqueue(_req(1)), return_exceptions=True)
        assert all(isinstance(r, ValueError) and "Result count mismatch" in str(r) for r in res)

    asyncio.run(run())


# ---- executor ---------------------------------------------------------------------------------------

class _Index:
    is_loaded = True

    def __init__(self):
        self.calls = []

    def search(self, emb, k):
        self.calls.append((emb.copy(), k))
        n = emb.shape[0]
        ids = np.arange(n * k, dtype=np.int64).reshape(n, k)
        return (np.linspace(1, 0, n * k, dtype=np.float32).reshape(n, k), ids)


class _Embedder:
    is_loaded = True

    def encode(self, texts):
        return np.ones((len(texts), 4), dtype=np.float32) * np.arange(len(texts))[:, None]


class _Reranker:
    def rerank_batch(self, queries, docs_batch, top_n=None):
        from rag_inference_pipeline_amd.components.schemas import RerankedDocument
        return [[RerankedDocument(doc_id=d.doc_id, title=d.title, content=d.content, category=d.category,
                                  score=1.0 / (1 + d.doc_id)) for d in reversed(docs)] for docs in docs_batch]


def _executor(components, **kw):
    reg = ComponentRegistry()
    for name, comp in components.items():
        reg.register(name, comp)
    kw.setdefault("DISABLE_CACHE_FOR_PROFILING", "true")
    return RetrievalExecutor(reg, PipelineSettings(retrieval_k=3, **kw)), reg


def test_executor_precomputed_path_order_and_missing_embedding():
    index = _Index()
    ex, _ = _executor({"faiss_store": index})
    batch = Batch(1, [_req(0, [0.0] * 4), _req(1, [1.0] * 4)])
    items = ex._process_batch_sync(batch)
    assert [it.request_id for it in items] == ["r0", "r1"]
    assert [d.doc_id for d in items[1].docs] == [3, 4, 5] and items[0].docs[0].title == ""
    assert index.calls[0][0].dtype == np.float32 and index.calls[0][1] == 3
    with pytest.raises(ValueError, match="Missing embedding in batch"):
        ex._process_batch_sync(Batch(2, [_req(0, [0.0] * 4), _req(1, None)]))
    with pytest.raises(RuntimeError, match="Embedding generator not available"):
        ex._process_batch_sync(Batch(3, [_req(0, None)]))
    ex2, _ = _executor({})
    with pytest.raises(RuntimeError, match="FAISS store not available"):
        ex2._process_batch_sync(Batch(4, [_req(0, [0.0] * 4)]))


def test_executor_encode_path_and_rerank_replace_scores():
    ex, _ = _executor({"faiss_store": _Index(), "embedding_generator": _Embedder(), "reranker": _Reranker()})
    items = ex._process_batch_sync(Batch(1, [_req(0), _req(1)]))
    assert [d.doc_id for d in items[0].docs] == [2, 1, 0]
    assert items[0].docs[0].score == pytest.approx(1 / 3)


def test_executor_accepts_binary_embeddings():
    """Optional wire form of a precomputed embedding: little-endian fp32 bytes (base64 in JSON)."""
    import base64

    from rag_inference_pipeline_amd.schemas import RetrievalRequestItem
    index = _Index()
    ex, _ = _executor({"faiss_store": index})
    e0, e1 = np.arange(4, dtype=np.float32), np.arange(4, dtype=np.float32) * 0.5 + 1
    item = RetrievalRequestItem(request_id="r0", query="q", embedding_f32=base64.b64encode(e0.tobytes()).decode())
    assert item.embedding_f32 == e0.tobytes()                      # base64 text is decoded on validation
    reqs = [PendingRequest(request_id="r0", query="q", embedding_f32=e0.tobytes(), timestamp=0.0),
            PendingRequest(request_id="r1", query="q", embedding=e1.tolist(), timestamp=0.0)]   # forms may mix
    ex._process_batch_sync(Batch(1, reqs))
    np.testing.assert_array_equal(index.calls[0][0], np.stack([e0, e1]))


def test_executor_compressed_payload_mode_is_an_lz4_frame_of_compact_json():
    """api.py:516-523: docs leave as lz4.frame(msgspec JSON) and `docs` is empty; the frame must be what
    generation/service.py:429-431 decodes (standard LZ4 frame, list of doc dicts in field order)."""
    import json

    from rag_inference_pipeline_amd import lz4frame
    ex, _ = _executor({"faiss_store": _Index(), "embedding_generator": _Embedder(), "reranker": _Reranker()},
                      DOCUMENTS_PAYLOAD_MODE="compressed")
    items = ex._process_batch_sync(Batch(1, [_req(0), _req(1)]))
    assert all(it.docs == [] and isinstance(it.compressed_docs, bytes) for it in items)
    frame = items[0].compressed_docs
    assert frame[:4] == bytes([0x04, 0x22, 0x4D, 0x18]) and frame[-4:] == b"\0\0\0\0"   # magic ... end mark
    raw = lz4frame.decompress(frame)
    assert raw.startswith(b'[{"doc_id":2,"title":"","content":"","category":"","score":')       # compact, field order
    docs = json.loads(raw)
    assert [d["doc_id"] for d in docs] == [2, 1, 0] and docs[0]["score"] == pytest.approx(1 / 3)


def _liblz4_decompress(frame: bytes) -> bytes:
    """Independent reader for the `compressed` payload: the system's liblz4 (the library behind python-lz4,
    which the generation node decodes with: generation/service.py:429-431), through ctypes LZ4F_decompress.
    Tests only — the product never links it."""
    import ctypes as C
    import ctypes.util

    path = ctypes.util.find_library("lz4") or "liblz4.so.1"
    try:
        lib = C.CDLL(path)
    except OSError:
        pytest.skip("liblz4 is not installed on this machine")
    lib.LZ4F_getVersion.restype = C.c_uint
    lib.LZ4F_createDecompressionContext.restype = C.c_size_t
    lib.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lib.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    lib.LZ4F_isError.restype = C.c_uint
    lib.LZ4F_isError.argtypes = [C.c_size_t]
    lib.LZ4F_getErrorName.restype = C.c_char_p
    lib.LZ4F_getErrorName.argtypes = [C.c_size_t]
    lib.LZ4F_decompress.restype = C.c_size_t
    lib.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    ctx = C.c_void_p()
    rc = lib.LZ4F_createDecompressionContext(C.byref(ctx), lib.LZ4F_getVersion())
    assert not lib.LZ4F_isError(rc), lib.LZ4F_getErrorName(rc)
    out = bytearray()
    try:
        src = C.create_string_buffer(frame, len(frame))
        dst = C.create_string_buffer(1 << 20)
        pos, hint = 0, 1
        while pos < len(frame):
            dn, sn = C.c_size_t(len(dst)), C.c_size_t(len(frame) - pos)
            hint = lib.LZ4F_decompress(ctx, dst, C.byref(dn), C.byref(src, pos), C.byref(sn), None)
            assert not lib.LZ4F_isError(hint), lib.LZ4F_getErrorName(hint)
            out += dst.raw[:dn.value]
            pos += sn.value
            assert sn.value or dn.value, "liblz4 made no progress"
        assert hint == 0, "liblz4: frame not complete at the end of the input"
    finally:
        lib.LZ4F_freeDecompressionContext(ctx)
    return bytes(out)


def test_compressed_payload_decodes_through_liblz4():
    """frames from lz4frame.compress (csrc/rag_lz4.cpp block compressor + frame header)
    were only ever read back by this repo's own decoder.  Here the system's liblz4 — an implementation
    that shares no code with this repo — decodes the retrieval node's `compressed` payload, the edge
    cases of the frame test and the whole fuzz corpus of test_lz4_frame_fuzz_round_trip."""
    import json
    import os

    from rag_inference_pipeline_amd import lz4frame
    ex, _ = _executor({"faiss_store": _Index(), "embedding_generator": _Embedder(), "reranker": _Reranker()},
                      DOCUMENTS_PAYLOAD_MODE="compressed")
    for it in ex._process_batch_sync(Batch(1, [_req(0), _req(1)])):
        raw = _liblz4_decompress(it.compressed_docs)
        assert raw == lz4frame.decompress(it.compressed_docs) and len(json.loads(raw)) == 3
    rng = np.random.default_rng(0)
    words = [b"retrieval", b"augmented", b"generation", b"pipeline", b"vector", b"index", b" ", b"\xc3\xa9"]
    for data in [b"", b"a", b"abc" * 5, bytes(70_000), os.urandom(3000), b"".join(rng.choice(words, size=20_000)),
                 bytes(range(256)) * 40, b"x" * ((4 << 20) + 17)]:
        assert _liblz4_decompress(lz4frame.compress(data)) == data
    for trial, data in enumerate(_lz4_fuzz_corpus()):
        assert _liblz4_decompress(lz4frame.compress(data)) == data, trial
    if importlib.util.find_spec("lz4") is not None:      # python-lz4 itself, when a machine has it
        import lz4.frame
        for data in list(_lz4_fuzz_corpus())[:50]:
            assert lz4.frame.decompress(lz4frame.compress(data)) == data


def test_lz4_frame_round_trips_and_header_checksum():
    import os

    from rag_inference_pipeline_amd import lz4frame
    rng = np.random.default_rng(0)
    words = [b"retrieval", b"augmented", b"generation", b"pipeline", b"vector", b"index", b" ", b"\xc3\xa9"]
    cases = [b"", b"a", b"abc" * 5, bytes(70_000), os.urandom(3000), b"".join(rng.choice(words, size=20_000)),
             bytes(range(256)) * 40, b"x" * ((4 << 20) + 17)]       # the last one spans two blocks
    for data in cases:
        frame = lz4frame.compress(data)
        assert frame[4:7] == bytes([0x60, 0x70, 0x73])      # FLG, BD, (xxh32(FLG BD) >> 8) & 0xFF
        assert lz4frame.decompress(frame) == data
    text = cases[5]
    assert len(lz4frame.compress(text)) < len(text) // 2    # it does compress


def _lz4_fuzz_corpus():
    """300 seeded buffers mixing literals, short and long matches, overlapping copies and block
    boundaries of the token format (lengths around 15 / 270, offsets up to 65535), plus a match beyond the window."""
    rng = np.random.default_rng(7)
    for _ in range(300):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                parts.append(rng.integers(0, 256, size=int(rng.integers(0, 400)), dtype=np.uint8).tobytes())
            elif kind == 1:
                parts.append(bytes([int(rng.integers(0, 256))]) * int(rng.choice([1, 4, 14, 15, 16, 19, 269, 270, 271, 5000])))
            elif kind == 2:
                unit = rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
                parts.append(unit * int(rng.integers(1, 60)))
            elif parts:
                parts.append(parts[int(rng.integers(0, len(parts)))])          # a far back-reference
        yield b"".join(parts)
    far = np.random.default_rng(8).integers(0, 256, size=300, dtype=np.uint8).tobytes()
    yield far + bytes(70_000) + far                        # the second copy is beyond the 64 KiB window


def test_lz4_frame_fuzz_round_trip():
    from rag_inference_pipeline_amd import lz4frame
    for trial, data in enumerate(_lz4_fuzz_corpus()):
        assert lz4frame.decompress(lz4frame.compress(data)) == data, trial


def test_executor_result_cache_hits_skip_the_index():
    index = _Index()
    ex, _ = _executor({"faiss_store": index}, DISABLE_CACHE_FOR_PROFILING="false")
    b1 = Batch(1, [_req(0, [0.5] * 4), _req(1, [1.5] * 4)])
    first = ex._process_batch_sync(b1)
    assert len(index.calls) == 1 and index.calls[0][0].shape == (2, 4)
    b2 = Batch(2, [_req(2, [1.5] * 4), _req(3, [2.5] * 4), _req(4, [0.5] * 4)])
    second = ex._process_batch_sync(b2)
    assert len(index.calls) == 2 and index.calls[1][0].shape == (1, 4)  # only the unseen embedding is searched
    assert [d.doc_id for d in second[0].docs] == [d.doc_id for d in first[1].docs]
    assert [d.doc_id for d in second[2].docs] == [d.doc_id for d in first[0].docs]
    ex.clear_cache()
    ex._process_batch_sync(Batch(3, [_req(5, [0.5] * 4)]))
    assert len(index.calls) == 3
    ex_off, _ = _executor({"faiss_store": index}, DISABLE_CACHE_FOR_PROFILING="true")
    ex_off._process_batch_sync(b1); ex_off._process_batch_sync(b1)
    assert len(index.calls) == 5


def test_executor_end_to_end_through_scheduler():
    async def run():
        ex, _ = _executor({"faiss_store": _Index()}, retrieval_batch_size=2, retrieval_max_batch_delay_ms=20)
        await ex.start()
        outs = await asyncio.gather(*[
            ex.process_request(RetrievalRequestItem(request_id=f"r{i}", query="q", embedding=[float(i)] * 4))
            for i in range(3)])
        await ex.stop()
        assert [o.request_id for o in outs] == ["r0", "r1", "r2"]
        assert all(len(o.docs) == 3 for o in outs)

    asyncio.run(run())


def test_executor_freezes_the_collector_once_after_its_first_batch():
    """settings.gc_freeze_after_warmup: what is alive after the first served batch leaves the cyclic collector's sight
    (a full collection over torch + transformers + pydantic costs 40-60 ms: one batch in ten of a rerank profile)."""
    import gc
    gc.unfreeze()
    try:
        ex, _ = _executor({"faiss_store": _Index()})
        assert gc.get_freeze_count() == 0
        ex._process_batch_sync(Batch(1, [_req(0, [0.5] * 4)]))
        frozen = gc.get_freeze_count()
        assert frozen > 0
        ex._process_batch_sync(Batch(2, [_req(1, [1.5] * 4)]))
        assert gc.get_freeze_count() == frozen          # once, not per batch
        gc.unfreeze()
        off, _ = _executor({"faiss_store": _Index()}, RAG_AMD_GC_FREEZE="false")
        off._process_batch_sync(Batch(3, [_req(2, [2.5] * 4)]))
        assert gc.get_freeze_count() == 0
    finally:
        gc.unfreeze()
