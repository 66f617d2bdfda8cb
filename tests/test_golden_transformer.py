"""Committed golden vectors of the transformer side (tests/golden/*.npz, scheduler_traces.json; written by
tests/golden/make_golden_transformer.py with `transformers` as the source of the expected values — the upstream
dependency behind the reference's embedder and reranker, embedding.py:127-133, reranker.py:237-272).

CPU: oracle/bert.py and the host-side scheduler against the fixtures (no `transformers` import: the GPU box and a
plain checkout need only the data).  GPU (`-m gpu`): the HIP kernels against the same fixtures."""
import asyncio
import json
import os
import time

import numpy as np
import pytest

from oracle import bert as obert
from rag_inference_pipeline_amd.batch_scheduler import AdaptiveBatchPolicy, Batch, BatchScheduler
from rag_inference_pipeline_amd.bert import BertConfig
from rag_inference_pipeline_amd.schemas import PendingRequest
from rag_inference_pipeline_amd.telemetry import counters

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = BertConfig(**json.loads(str(z["config"])))
    w = {k[2:]: (z[k].astype(np.uint32) << 16).view(np.float32) for k in z.files if k.startswith("w.")}
    cu = np.concatenate([[0], np.cumsum(z["lens"])])
    seqs = [z["ids"][cu[i]:cu[i + 1]].tolist() for i in range(len(z["lens"]))]
    types = [z["types"][cu[i]:cu[i + 1]].tolist() for i in range(len(z["lens"]))]
    return cfg, w, seqs, types, z


ENCODERS = [("encoder_minilm_tiny.npz", "mean"), ("encoder_bge_cls_tiny.npz", "cls")]


@pytest.mark.parametrize("name,pooling", ENCODERS)
def test_oracle_encoder_matches_the_transformers_fixture(name, pooling):
    cfg, w, seqs, types, z = _load(name)
    assert cfg.pooling == pooling
    hs = np.concatenate(obert.hidden_states(cfg, w, seqs, types))
    np.testing.assert_allclose(hs, z["hidden"], atol=2e-5, rtol=1e-5)
    emb = obert.embed(cfg, w, seqs, types)
    np.testing.assert_allclose(emb, z["embedding"], atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-6)


def test_oracle_reranker_matches_the_transformers_fixture():
    cfg, w, seqs, types, z = _load("rerank_order.npz")
    logits = obert.classify(cfg, w, seqs, types, sigmoid=False)[:, 0]
    scores = obert.classify(cfg, w, seqs, types)[:, 0]
    np.testing.assert_allclose(logits, z["logits"], atol=2e-5)
    np.testing.assert_allclose(scores, z["scores"], atol=1e-5)
    assert scores[2] == scores[7]                       # the same pair twice: exactly equal scores ...
    order = obert.rerank_order(scores.tolist())
    assert order == z["order"].tolist()                 # ... kept in input order by the stable sort (reranker.py:270)
    assert obert.rerank_order(scores.tolist(), top_n=int(z["top_n"])) == z["order_top_n"].tolist()
    assert obert.rerank_order(z["scores"].tolist()) == z["order"].tolist()


def test_ranked_component_logic_reproduces_the_fixture_order():
    """The product's host-side ranking (components/reranker.py: sort + top_n over the scores the kernels return)
    on the fixture's scores, ties included."""
    from rag_inference_pipeline_amd.components.reranker import Reranker
    from rag_inference_pipeline_amd.components.schemas import Document

    _, _, seqs, _, z = _load("rerank_order.npz")
    docs = [Document(doc_id=100 + i, title="", content=f"d{i}") for i in range(len(seqs))]
    ranked = Reranker._ranked(docs, z["scores"].tolist(), None)
    assert [d.doc_id - 100 for d in ranked] == z["order"].tolist()
    top = Reranker._ranked(docs, z["scores"].tolist(), int(z["top_n"]))
    assert [d.doc_id - 100 for d in top] == z["order_top_n"].tolist()
    np.testing.assert_array_equal(np.array([d.score for d in ranked], np.float32), z["scores"][z["order"]])


def _traces():
    with open(os.path.join(GOLDEN, "scheduler_traces.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("trace", _traces()["scheduler"], ids=lambda t: t["name"])
def test_scheduler_follows_the_scripted_traces(trace):
    """Flush sizes, reasons and membership for scripted arrivals (reference rules batch_scheduler.py:160-236; the
    reference's own timing tests, tests/test_batch_scheduler.py:313-369, are the first two scripts)."""
    service = "trace_" + trace["name"]

    async def run():
        seen = []

        async def fn(batch: Batch):
            seen.append([r.request_id for r in batch.requests])
            return [r.request_id for r in batch.requests]

        sch = BatchScheduler(trace["batch_size"], trace["max_batch_delay_ms"], fn, service_name=service)
        await sch.start()
        tasks = []
        for delay_ms, rid in trace["arrivals"]:
            if delay_ms:
                await asyncio.sleep(delay_ms / 1000.0)
            tasks.append(asyncio.create_task(sch.enqueue(PendingRequest(request_id=rid, query="q", timestamp=time.time()))))
            await asyncio.sleep(0)   # let the request reach the queue before the next one is scripted
        await asyncio.sleep(trace["stop_after_ms"] / 1000.0)
        await sch.stop()
        out = await asyncio.gather(*tasks)
        return seen, out

    before = {r: counters.get("pipeline_batch_flush_total", service=service, reason=r) for r in ("full", "timeout", "shutdown")}
    seen, out = asyncio.run(run())
    assert out == [rid for _, rid in trace["arrivals"]]            # every request got its own result back, in order
    assert seen == [f["ids"] for f in trace["flushes"]]
    assert [len(b) for b in seen] == [f["size"] for f in trace["flushes"]]
    for reason in ("full", "timeout", "shutdown"):
        want = sum(1 for f in trace["flushes"] if f["reason"] == reason)
        assert counters.get("pipeline_batch_flush_total", service=service, reason=reason) - before[reason] == want, reason


@pytest.mark.parametrize("trace", _traces()["adaptive_policy"], ids=lambda t: f"bs{t['max_batch_size']}")
def test_adaptive_policy_follows_the_scripted_depths(trace):
    pol = AdaptiveBatchPolicy(trace["max_batch_size"], trace["min_delay_sec"], trace["max_delay_sec"])
    got = [pol.update(dp) for dp in trace["depths"]]
    np.testing.assert_allclose(got, trace["delays"], rtol=1e-12)


# ---- the HIP kernels against the same fixtures ---------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name,pooling", ENCODERS)
def test_hip_encoder_matches_the_transformers_fixture(gpu_required, name, pooling):
    from rag_inference_pipeline_amd.bert import BertModel
    cfg, w, seqs, types, z = _load(name)
    model = BertModel(cfg, w)
    np.testing.assert_allclose(model.hidden_states(seqs, types), z["hidden"], atol=5e-5, rtol=1e-5)
    emb = model.embed(seqs, types)
    np.testing.assert_allclose(emb, z["embedding"], atol=1e-5)
    dev = model.embed_to_device(seqs, types)            # the device hand-off returns the same values
    np.testing.assert_array_equal(dev.numpy().view(np.uint32), emb.view(np.uint32))
    model.close()


@pytest.mark.gpu
def test_hip_reranker_matches_the_transformers_fixture(gpu_required):
    from rag_inference_pipeline_amd.bert import BertModel
    cfg, w, seqs, types, z = _load("rerank_order.npz")
    model = BertModel(cfg, w)
    logits = model.classify(seqs, types, sigmoid=False)[:, 0]
    scores = model.classify(seqs, types)[:, 0]
    np.testing.assert_allclose(logits, z["logits"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(scores, z["scores"], atol=1e-5)
    assert scores[2] == scores[7]                       # identical pairs score identically whatever their batch position
    assert obert.rerank_order(scores.tolist()) == z["order"].tolist()
    assert obert.rerank_order(scores.tolist(), top_n=int(z["top_n"])) == z["order_top_n"].tolist()
    model.close()
