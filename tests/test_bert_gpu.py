"""GPU parity: HIP transformer (rag_bert_* through the C ABI) vs the torch-CPU oracle.
Tolerances (fp32 everywhere; accumulation order and exp/erf/tanh/rsqrt implementations differ
between the GPU and libm): hidden states 5e-5 abs on O(1) activations, unit-norm embeddings 1e-5,
sigmoid scores 1e-5."""
import numpy as np
import pytest

from oracle import bert as obert
from rag_inference_pipeline_amd.bert import BertConfig, BertModel, random_weights

pytestmark = pytest.mark.gpu


def _seqs(rng, lengths, vocab):
    return [rng.integers(3, vocab, size=int(n)).tolist() for n in lengths]


def _small(cfg: BertConfig, vocab=2000) -> BertConfig:
    cfg.vocab_size = vocab  # the embedding table is the only thing that scales with the vocabulary
    return cfg


def test_minilm_hidden_states_and_mean_pooling(gpu_required):
    cfg = _small(BertConfig.minilm_l6())
    w = random_weights(cfg, 0)
    model = BertModel(cfg, w)
    rng = np.random.default_rng(0)
    seqs = _seqs(rng, [1, 2, 8, 13, 20, 63, 64, 65, 130], cfg.vocab_size)
    types = [[0] * len(s) for s in seqs]
    got = model.hidden_states(seqs, types)
    want = np.concatenate(obert.hidden_states(cfg, w, seqs, types))
    np.testing.assert_allclose(got, want, atol=5e-5, rtol=1e-4)
    emb = model.embed(seqs, types)
    np.testing.assert_allclose(emb, obert.embed(cfg, w, seqs, types), atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    raw = model.embed(seqs, types, normalize=False)
    np.testing.assert_allclose(raw, obert.embed(cfg, w, seqs, types, normalize=False), atol=5e-5, rtol=1e-4)
    model.close()


def test_bge_base_cls_pooling_batch_of_32_queries(gpu_required):
    cfg = _small(BertConfig.bge_base())
    w = random_weights(cfg, 1)
    model = BertModel(cfg, w)
    rng = np.random.default_rng(1)
    seqs = _seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size)  # query-like lengths
    emb = model.embed(seqs)
    assert emb.shape == (32, 768) and emb.dtype == np.float32
    np.testing.assert_allclose(emb, obert.embed(cfg, w, seqs), atol=1e-5)
    model.close()


def test_cross_encoder_bert_head_scores_and_order(gpu_required):
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    w = random_weights(cfg, 2)
    w["head_out_w"] = (w["head_out_w"] * 20).astype(np.float32)  # spread the logits so sigmoid is not flat
    model = BertModel(cfg, w)
    rng = np.random.default_rng(2)
    seqs = _seqs(rng, rng.integers(24, 65, size=100), cfg.vocab_size)  # 100 (query, doc) pairs
    types = [[0] * 10 + [1] * (len(s) - 10) for s in seqs]
    probs = model.classify(seqs, types)
    want = obert.classify(cfg, w, seqs, types)
    np.testing.assert_allclose(probs, want, atol=1e-5)
    logits = model.classify(seqs, types, sigmoid=False)
    np.testing.assert_allclose(logits, obert.classify(cfg, w, seqs, types, sigmoid=False), atol=1e-4, rtol=1e-4)
    # the order the reranker returns: identical unless two oracle scores are closer than the tolerance
    o_gpu, o_cpu = obert.rerank_order(probs[:, 0].tolist()), obert.rerank_order(want[:, 0].tolist())
    gaps = np.abs(np.diff(np.sort(want[:, 0])))
    if gaps.min() > 2e-5:
        assert o_gpu == o_cpu
    model.close()


@pytest.mark.parametrize("nseq,lo,hi", [(24, 8, 21), (40, 30, 61)])   # T ~ 350 (fp32 MFMA everywhere) and ~ 1800 (split-bf16 for T rows)
def test_first_token_last_layer_matches_the_full_pass(gpu_required, nseq, lo, hi):
    """CLS pooling and the classifier head run the last layer on first tokens only (rag_bert.hip: last_first_only).
    Those nseq rows go through GEMM kernels / split-K counts picked for nseq rows, the full pass (hidden_states)
    through the ones picked for T rows — different summation orders on either side of the 1024-row switch, the
    same values to fp32 rounding."""
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    cfg.pooling = "cls"
    w = random_weights(cfg, 5)
    model = BertModel(cfg, w)
    rng = np.random.default_rng(5)
    lens = rng.integers(lo, hi, size=nseq)
    seqs = _seqs(rng, lens, cfg.vocab_size)
    types = [[0] * 4 + [1] * (len(s) - 4) for s in seqs]
    T = int(lens.sum())
    assert (T > 1024) == (lo >= 30)
    first = np.concatenate([[0], np.cumsum(lens)[:-1]])
    full = model.hidden_states(seqs, types)[first]                     # the full pass, then first-token rows
    cls = model.embed(seqs, types, normalize=False, pooling="cls")     # the compact path
    np.testing.assert_allclose(cls, full, atol=2e-5, rtol=1e-5)
    unit = model.embed(seqs, types, normalize=True, pooling="cls")
    np.testing.assert_allclose(unit, full / np.linalg.norm(full, axis=1, keepdims=True), atol=1e-5)
    logits = model.classify(seqs, types, sigmoid=False)[:, 0]          # compact path + head
    pooled = np.tanh(full.astype(np.float64) @ w["head_dense_w"].astype(np.float64).T + w["head_dense_b"])
    want = pooled @ w["head_out_w"].astype(np.float64).T + w["head_out_b"]
    np.testing.assert_allclose(logits, want[:, 0], atol=1e-5, rtol=1e-5)
    model.close()


def test_xlm_roberta_head_and_position_offset_long_sequences(gpu_required):
    cfg = BertConfig(vocab_size=3000, hidden=768, n_layers=2, n_heads=12, intermediate=3072, max_positions=514,
                     type_vocab=1, pos_offset=2, head="roberta", n_labels=1, ln_eps=1e-5)
    w = random_weights(cfg, 3)
    model = BertModel(cfg, w)
    rng = np.random.default_rng(3)
    seqs = _seqs(rng, [512, 300, 7], cfg.vocab_size)  # 512 = truncate_length, the reference maximum
    np.testing.assert_allclose(model.classify(seqs, sigmoid=False), obert.classify(cfg, w, seqs, sigmoid=False),
                               atol=2e-4, rtol=1e-4)
    model.close()


def test_bert_error_contract(gpu_required):
    from rag_inference_pipeline_amd import _native
    cfg = _small(BertConfig.minilm_l6(), vocab=100)
    w = random_weights(cfg, 4)
    model = BertModel(cfg, w)
    with pytest.raises(_native.RagAmdError, match="classifier head"):
        model.classify([[1, 2, 3]])
    with pytest.raises(_native.RagAmdError, match="positions"):
        model.embed([list(range(3, 90)) * 7])  # 609 tokens > 512 positions
    with pytest.raises(_native.RagAmdError, match="empty"):
        model.embed([[1, 2], []])
    bad = dict(w)
    del bad["layer0.qkv_w"]
    with pytest.raises(KeyError):
        BertModel(cfg, bad)
    with pytest.raises(_native.RagAmdError, match="head dim"):
        BertModel(BertConfig(vocab_size=50, hidden=96, n_layers=1, n_heads=2, intermediate=64, max_positions=16),
                  random_weights(BertConfig(vocab_size=50, hidden=96, n_layers=1, n_heads=2, intermediate=64,
                                            max_positions=16), 0))
    model.close()


def test_cross_encoder_fp16_gemm_mode_tracks_fp32_oracle(gpu_required):
    """Opt-in fp16-input GEMMs (the reference's GPU precision, reranker.py:91-93): scores stay within
    fp16 rounding of the fp32 oracle and the fp32 GPU path; the default path is untouched."""
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    w = random_weights(cfg, 5)
    w["head_out_w"] = (w["head_out_w"] * 20).astype(np.float32)
    rng = np.random.default_rng(5)
    seqs = _seqs(rng, rng.integers(24, 65, size=64), cfg.vocab_size)  # 64 pairs > 1024 tokens: the big-M path
    types = [[0] * 10 + [1] * (len(s) - 10) for s in seqs]
    want = obert.classify(cfg, w, seqs, types, sigmoid=False)
    f32 = BertModel(cfg, w)
    got32 = f32.classify(seqs, types, sigmoid=False)
    f32.close()
    cfg16 = _small(BertConfig.ms_marco_minilm_l6())
    cfg16.gemm_dtype = "f16"
    f16 = BertModel(cfg16, w)
    got16 = f16.classify(seqs, types, sigmoid=False)
    probs16 = f16.classify(seqs, types)
    f16.close()
    np.testing.assert_allclose(got32, want, atol=1e-4, rtol=1e-4)
    assert np.abs(got16 - want).max() < 3e-2 * max(1.0, np.abs(want).max())  # fp16 inputs: ~1e-3 relative per GEMM
    assert np.abs(got16 - want).max() > 0  # and it really is a different arithmetic
    np.testing.assert_allclose(probs16, obert.classify(cfg, w, seqs, types), atol=5e-3)


def test_fp16_mode_big_batches_run_on_fragment_ordered_fp16_activations(gpu_required, monkeypatch):
    """RAG_GEMM_F16 above 1024 tokens: activations are fp16 in MFMA-fragment order from the embeddings to the last
    layer (csrc/gemm_wt.hip.h, bert_tiled.hip.h) — the arithmetic of a .half() model (reference reranker.py:91-93).
    Every output kind is held against the fp32 oracle within fp16's rounding (hidden states are O(1): 3e-2 is ~15
    half-precision ulps after six layers) and against the same mode on row-major fp32 activations
    (RAG_AMD_ROW_MAJOR=1, the round-2 data path); a ragged token count, sequences that straddle 32-token row
    blocks, mean pooling, first-token pooling, both DH."""
    for cfg_fn, seed in ((BertConfig.ms_marco_minilm_l6, 21), (BertConfig.bge_base, 22)):
        cfg = _small(cfg_fn())
        cfg.n_layers = min(cfg.n_layers, 3)
        cfg.gemm_dtype = "f16"
        w = random_weights(cfg, seed)
        if "head_out_w" in w:
            w["head_out_w"] = (w["head_out_w"] * 20).astype(np.float32)
        rng = np.random.default_rng(seed)
        seqs = _seqs(rng, rng.integers(5, 71, size=61), cfg.vocab_size)   # ~2300 tokens
        if sum(map(len, seqs)) % 32 == 0:                                 # ... and never whole row blocks
            seqs += _seqs(rng, [7], cfg.vocab_size)
        assert sum(map(len, seqs)) > 1024 and sum(map(len, seqs)) % 32 != 0
        types = [[0] * 3 + [1] * (len(s) - 3) for s in seqs]
        tiled = BertModel(cfg, w)
        monkeypatch.setenv("RAG_AMD_ROW_MAJOR", "1")
        rowm = BertModel(cfg, w)
        monkeypatch.delenv("RAG_AMD_ROW_MAJOR")
        monkeypatch.setenv("RAG_AMD_TILED_ATTENTION_F32", "1")     # same layout, attention products on the fp32 MFMA
        attn32 = BertModel(cfg, w)
        monkeypatch.delenv("RAG_AMD_TILED_ATTENTION_F32")
        want_h = np.concatenate(obert.hidden_states(cfg, w, seqs, types))
        got_h, ref_h = tiled.hidden_states(seqs, types), rowm.hidden_states(seqs, types)
        scale = max(1.0, np.abs(want_h).max())
        assert np.abs(got_h - want_h).max() < 3e-2 * scale, np.abs(got_h - want_h).max()
        assert np.abs(got_h - ref_h).max() < 3e-2 * scale
        assert np.abs(got_h - ref_h).max() > 0          # and it really is another data path
        a32_h = attn32.hidden_states(seqs, types)
        assert np.abs(a32_h - want_h).max() < 3e-2 * scale
        assert 0 < np.abs(got_h - a32_h).max() < 2e-2 * scale   # fp16-MFMA attention: P rounded to fp16, nothing else
        attn32.close()
        for normalize in (True, False):
            want_e = obert.embed(cfg, w, seqs, types, normalize=normalize)
            got_e = tiled.embed(seqs, types, normalize=normalize)
            assert np.abs(got_e - want_e).max() < 2e-2 * max(1.0, np.abs(want_e).max())
        if cfg.head != "none":
            want_l = obert.classify(cfg, w, seqs, types, sigmoid=False)
            got_l = tiled.classify(seqs, types, sigmoid=False)
            assert np.abs(got_l - want_l).max() < 3e-2 * max(1.0, np.abs(want_l).max())
            np.testing.assert_allclose(tiled.classify(seqs, types), obert.classify(cfg, w, seqs, types), atol=5e-3)
        tiled.close()
        rowm.close()


def test_split_bf16_gemms_keep_fp32_accuracy(gpu_required):
    """Default big-batch GEMMs run on the bf16 matrix cores (three-way exact split, six products):
    hidden states must be as close to a float64 evaluation as the fp32-MFMA path is."""
    import torch
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    w = random_weights(cfg, 9)
    rng = np.random.default_rng(9)
    seqs = _seqs(rng, rng.integers(24, 65, size=80), cfg.vocab_size)     # ~3500 tokens: big-M path
    types = [[0] * 8 + [1] * (len(s) - 8) for s in seqs]
    split = BertModel(cfg, w)
    got_split = split.hidden_states(seqs, types)
    split.close()
    cfg_s = _small(BertConfig.ms_marco_minilm_l6())
    cfg_s.gemm_dtype = "f32_strict"
    strict = BertModel(cfg_s, w)
    got_strict = strict.hidden_states(seqs, types)
    strict.close()
    want64 = np.concatenate(obert.hidden_states(cfg, w, seqs, types, dtype=torch.float64), axis=0)
    e_split = np.abs(got_split - want64).max()
    e_strict = np.abs(got_strict - want64).max()
    assert e_strict < 2e-5 and e_split < 2e-5, (e_split, e_strict)
    assert e_split < 3 * e_strict + 1e-6, (e_split, e_strict)            # same accuracy class as the fp32 MFMA
    assert np.abs(got_split - got_strict).max() > 0                      # and it really is another arithmetic


def test_values_outside_fp16_range_take_the_three_plane_path(gpu_required):
    """Default big-batch GEMMs write every fp32 operand as two fp16 numbers (include/rag_amd.h RAG_GEMM_F32): a value
    beyond fp16's range cannot go that way.  An activation of 1e5 (a feed-forward bias pushed there) raises the range
    flag and the pass is repeated on the three-plane bf16 split; a WEIGHT of 1e5 keeps the model on that path from the
    start.  Either way the result is the fp32-MFMA path's, to fp32 rounding."""
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    rng = np.random.default_rng(21)
    seqs = _seqs(rng, rng.integers(24, 65, size=40), cfg.vocab_size)     # ~1800 tokens: the big-batch path
    types = [[0] * 8 + [1] * (len(s) - 8) for s in seqs]

    def run(w, dtype):
        c = _small(BertConfig.ms_marco_minilm_l6())
        c.gemm_dtype = dtype
        m = BertModel(c, w)
        out = m.hidden_states(seqs, types)
        ev = m.range_events()[0]
        out2 = m.hidden_states(seqs, types)      # a second pass: the images built on first need are reused
        ev2 = m.range_events()[0]
        m.close()
        np.testing.assert_array_equal(out, out2)
        return out, ev, ev2

    w = random_weights(cfg, 21)
    base, ev, _ = run(w, "f32")
    assert ev == 0                                                   # ordinary values: nothing repeated
    w_act = dict(w)
    b = w["layer1.ffn_in_b"].copy()
    b[:7] = 1.0e5                                                    # GELU(1e5) = 1e5 feeds the ffn_out GEMM
    w_act["layer1.ffn_in_b"] = b
    got, ev, ev2 = run(w_act, "f32")
    want, _, _ = run(w_act, "f32_strict")
    assert ev == 1 and ev2 == 2                                      # every such pass is repeated, none is wrong
    assert np.isfinite(got).all()
    # (sums of terms of magnitude 1e4 in front of a LayerNorm: the two paths' fp32 summation orders differ by more
    # than they do on ordinary values)
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-5)
    w_big = dict(w)
    ww = w["layer2.attn_out_w"].copy()
    ww[3, 5] = 1.0e5
    w_big["layer2.attn_out_w"] = ww
    got, ev, _ = run(w_big, "f32")
    want, _, _ = run(w_big, "f32_strict")
    assert ev == 0                                                   # known at creation: no pass is ever repeated
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-5)
    assert np.abs(base - got).max() > 0


def test_small_batches_outside_fp16_range_are_flagged_and_repeated(gpu_required):
    """The query encoder's batch (<= 1024 tokens) runs the same two-plane GEMMs on 64-row tiles.  Host path: a pass
    that meets |x| >= 65504 is repeated on the fp32 MFMA.  Device hand-off: the pass cannot repeat itself, its pinned
    flag word says the embeddings are void (DeviceEmbeddings.valid()) and numpy() goes through `recompute`."""
    cfg = _small(BertConfig.minilm_l6())
    w = random_weights(cfg, 31)
    b = w["layer0.ffn_in_b"].copy()
    b[:5] = 1.0e5
    w_act = dict(w, **{"layer0.ffn_in_b": b})
    rng = np.random.default_rng(31)
    seqs = _seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size)
    strict_cfg = _small(BertConfig.minilm_l6())
    strict_cfg.gemm_dtype = "f32_strict"
    for weights, flagged in ((w, False), (w_act, True)):
        strict = BertModel(strict_cfg, weights)
        want = strict.embed(seqs)
        strict.close()
        model = BertModel(cfg, weights)
        got = model.embed(seqs)                                   # host path: repeats by itself
        np.testing.assert_allclose(got, want, atol=2e-5)
        assert model.range_events()[0] == (1 if flagged else 0)
        dev = model.embed_to_device(seqs)
        dev.recompute = lambda: model.embed(seqs)
        arr = dev.numpy()
        assert dev.valid() == (not flagged)
        np.testing.assert_allclose(arr, want, atol=2e-5)          # through recompute when the pass was void
        if not flagged:
            np.testing.assert_array_equal(arr.view(np.uint32), got.view(np.uint32))
        model.close()


def test_background_mode_gives_the_same_embeddings(gpu_required):
    """rag_bert_set_background: the small-batch GEMMs in their 32-KiB-LDS form (one K-step per stage, four waves) — what
    bench.py's pipelined leg runs under the corpus scan — against the oracle and the default form."""
    cfg = _small(BertConfig.bge_base())
    cfg.n_layers = 3
    w = random_weights(cfg, 41)
    rng = np.random.default_rng(41)
    seqs = _seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size)
    model = BertModel(cfg, w)
    base = model.embed(seqs)
    model.set_background(True)
    bg = model.embed(seqs)
    model.set_background(False)
    again = model.embed(seqs)
    model.close()
    want = obert.embed(cfg, w, seqs)
    np.testing.assert_allclose(bg, want, atol=1e-5)
    np.testing.assert_allclose(base, want, atol=1e-5)
    np.testing.assert_array_equal(again.view(np.uint32), base.view(np.uint32))   # switching back restores the default form


def test_query_encoder_replays_a_graph_per_padded_shape(gpu_required, monkeypatch):
    """rag_bert_forward_to_device on small batches (the query encoder): one hipGraph replay per call — shapes padded to
    buckets with dummy sequences, destinations passed through device cells — returns what the eager launches return
    (to fp32 rounding: the padded token count can change a split-K choice), for new shapes, cached shapes, shapes beyond
    the cache's capacity, mean and first-token pooling, with and without normalisation."""
    cfg = _small(BertConfig.minilm_l6())
    cfg.n_layers = 2
    w = random_weights(cfg, 31)
    graphed = BertModel(cfg, w)
    monkeypatch.setenv("RAG_AMD_ENCODER_GRAPH", "0")
    eager = BertModel(cfg, w)
    monkeypatch.delenv("RAG_AMD_ENCODER_GRAPH")
    rng = np.random.default_rng(31)
    shapes = [(n, lo, hi) for n in (1, 3, 11, 19, 32, 40) for lo, hi in ((2, 9), (8, 21), (10, 26))]   # 18+ padded shapes
    shapes += [(64, 8, 15), (7, 100, 140)]          # 64 sequences; 1000+ tokens: beyond the graph's range -> eager inside
    for rep in range(2):
        for n, lo, hi in shapes:
            seqs = _seqs(rng, rng.integers(lo, hi, size=n), cfg.vocab_size)
            types = [[0] * len(q) for q in seqs]
            for pooling, normalize in (("mean", True), ("cls", False)):
                a = graphed.embed_to_device(seqs, types, normalize=normalize, pooling=pooling)
                b = eager.embed_to_device(seqs, types, normalize=normalize, pooling=pooling)
                got, ref = a.numpy(), b.numpy()
                assert a.valid() and b.valid()
                np.testing.assert_allclose(got, ref, atol=3e-6, rtol=1e-5)
                if rep == 0 and n in (3, 32):
                    np.testing.assert_allclose(got, obert.embed(cfg, w, seqs, types, normalize=normalize, pooling=pooling),
                                               atol=1e-5 if normalize else 5e-5, rtol=1e-4)
    # back-to-back calls without reading the first result first: the second waits for the first's buffers itself
    s1 = _seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size)
    s2 = _seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size)
    a1 = graphed.embed_to_device(s1, [[0] * len(q) for q in s1])
    a2 = graphed.embed_to_device(s2, [[0] * len(q) for q in s2])
    np.testing.assert_allclose(a2.numpy(), eager.embed(s2, [[0] * len(q) for q in s2]), atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(a1.numpy(), eager.embed(s1, [[0] * len(q) for q in s1]), atol=3e-6, rtol=1e-5)
    graphed.close()
    eager.close()


@pytest.mark.parametrize("arch", ["minilm", "bge", "cross"])
def test_layernorms_folded_into_their_consumers_match_the_separate_launches(gpu_required, monkeypatch, arch):
    """RAG_AMD_ENCODER_LN_FOLD=1: the query-encoder path (<= 1024 tokens) without LayerNorm launches: the GEMM that produces a layer's pre-LayerNorm
    sum finishes its own split-K reduction (last split of a tile to arrive) and leaves block statistics, the GEMMs that
    read it apply the LayerNorm in their epilogue (gamma folded into the weight image) or recompute the residual from
    it.  Same values as the separate-launch form (the default) and as the oracle, on token counts around
    the 64-row tile boundaries, one-token batches, the graph replay path, and after several back-to-back passes (the
    arrival counters reset themselves)."""
    if arch == "minilm":
        cfg = _small(BertConfig.minilm_l6())
    elif arch == "bge":
        cfg = _small(BertConfig.bge_base())
        cfg.n_layers = 3
    else:
        cfg = _small(BertConfig.ms_marco_minilm_l6())
    w = random_weights(cfg, 91)
    for key in list(w):   # LayerNorm parameters away from (1, 0), so that folding them is actually exercised
        if key.endswith("_g"):
            w[key] = (w[key] * np.linspace(0.5, 1.5, w[key].size, dtype=np.float32)).astype(np.float32)
        elif key.endswith("ln1_b") or key.endswith("ln2_b") or key == "emb_ln_b":
            w[key] = (w[key] + np.linspace(-0.3, 0.3, w[key].size, dtype=np.float32)).astype(np.float32)
    monkeypatch.setenv("RAG_AMD_ENCODER_LN_FOLD", "1")   # (opt-in: measured slower than the separate launches)
    folded = BertModel(cfg, w)
    monkeypatch.delenv("RAG_AMD_ENCODER_LN_FOLD")
    plain = BertModel(cfg, w)
    rng = np.random.default_rng(91)
    cases = [[1], [2, 1], [63], [64], [65], [8] * 16, rng.integers(8, 21, size=32).tolist(), [100, 28], [120] * 8 + [57],
             rng.integers(8, 21, size=32).tolist()]
    for lens in cases:
        seqs = _seqs(rng, lens, cfg.vocab_size)
        types = [[0] * (len(q) // 2) + [1] * (len(q) - len(q) // 2) for q in seqs]
        if arch == "cross":
            got, ref = folded.classify(seqs, types, sigmoid=False), plain.classify(seqs, types, sigmoid=False)
            np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)
            np.testing.assert_allclose(got, obert.classify(cfg, w, seqs, types, sigmoid=False), atol=1e-4, rtol=1e-4)
            continue
        hid, hid_ref = folded.hidden_states(seqs, types), plain.hidden_states(seqs, types)
        np.testing.assert_allclose(hid, hid_ref, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(hid, np.concatenate(obert.hidden_states(cfg, w, seqs, types)), atol=5e-5, rtol=1e-4)
        emb = folded.embed(seqs, types)
        np.testing.assert_allclose(emb, plain.embed(seqs, types), atol=3e-6, rtol=1e-5)
        np.testing.assert_allclose(emb, obert.embed(cfg, w, seqs, types), atol=1e-5)
        np.testing.assert_allclose(folded.embed_to_device(seqs, types).numpy(), emb, atol=3e-6, rtol=1e-5)   # graph replay
    folded.close()
    plain.close()


@pytest.mark.parametrize("arch,w6", [("cross", "0"), ("minilm", "0"), ("bge", "0"), ("cross", "1"), ("bge", "1")])
def test_big_batch_layernorms_are_applied_by_their_consumers(gpu_required, monkeypatch, arch, w6):
    """Default mode above 1024 tokens: the GEMM that writes a layer's pre-LayerNorm sum leaves per-row statistics of each
    128-column tile in its epilogue, the GEMMs that read it apply the LayerNorm there (gamma folded into the image) or
    recompute the residual from it — no LayerNorm pass over the activations except the encoder's last.  Same values as
    the separate passes (RAG_AMD_LN_FOLD=0) and as the oracle: logits through the first-token tail, all hidden states,
    mean and first-token pooling; a ragged token count; LayerNorm parameters away from (1, 0).  w6 = "1": the same on the
    opt-in 128 x 192-tile GEMM (RAG_AMD_GEMM_W6=1: statistics per 64-column block), folded and plain."""
    monkeypatch.setenv("RAG_AMD_GEMM_W6", w6)
    if arch == "cross":
        cfg = _small(BertConfig.ms_marco_minilm_l6())
    elif arch == "minilm":
        cfg = _small(BertConfig.minilm_l6())
        cfg.n_layers = 3
    else:
        cfg = _small(BertConfig.bge_base())
        cfg.n_layers = 2
    w = random_weights(cfg, 93)
    for key in list(w):
        if key.endswith("_g"):
            w[key] = (w[key] * np.linspace(0.5, 1.5, w[key].size, dtype=np.float32)).astype(np.float32)
        elif key.endswith("ln1_b") or key.endswith("ln2_b") or key == "emb_ln_b":
            w[key] = (w[key] + np.linspace(-0.3, 0.3, w[key].size, dtype=np.float32)).astype(np.float32)
    folded = BertModel(cfg, w)
    monkeypatch.setenv("RAG_AMD_LN_FOLD", "0")
    plain = BertModel(cfg, w)
    monkeypatch.delenv("RAG_AMD_LN_FOLD")
    rng = np.random.default_rng(93)
    seqs = _seqs(rng, rng.integers(24, 65, size=47), cfg.vocab_size)     # ~2000 tokens, not a multiple of 128
    types = [[0] * 10 + [1] * (len(q) - 10) for q in seqs]
    if arch == "cross":
        got, ref = folded.classify(seqs, types, sigmoid=False), plain.classify(seqs, types, sigmoid=False)
        np.testing.assert_allclose(got, ref, atol=3e-5, rtol=1e-5)
        np.testing.assert_allclose(got, obert.classify(cfg, w, seqs, types, sigmoid=False), atol=1e-4, rtol=1e-4)
    hid, hid_ref = folded.hidden_states(seqs, types), plain.hidden_states(seqs, types)
    np.testing.assert_allclose(hid, hid_ref, atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(hid, np.concatenate(obert.hidden_states(cfg, w, seqs, types)), atol=5e-5, rtol=1e-4)
    for pooling in ("mean", "cls"):
        emb = folded.embed(seqs, types, pooling=pooling)
        np.testing.assert_allclose(emb, plain.embed(seqs, types, pooling=pooling), atol=3e-6, rtol=1e-5)
        np.testing.assert_allclose(emb, obert.embed(cfg, w, seqs, types, pooling=pooling), atol=1e-5)
    folded.close()
    plain.close()


def test_fused_feed_forward_kernel_matches_the_two_gemms(gpu_required, monkeypatch):
    """RAG_AMD_FFN_FUSED=1 (fp16 mode, hidden 384): FFN-in -> GELU -> FFN-out + residual as one kernel whose 1536-wide
    intermediate stays in registers (the H1 tile's accumulator registers are the second product's operand; W2's image is
    packed in that column order).  Same values as the two-GEMM form to fp16 rounding — the intermediate is rounded to
    fp16 at the same point in both, only the fp32 accumulation order differs — and inside the fp16 mode's bound against
    the oracle; ragged token counts (partial row blocks) included."""
    cfg = _small(BertConfig.ms_marco_minilm_l6())
    cfg.gemm_dtype = "f16"
    w = random_weights(cfg, 55)
    w["head_out_w"] = (w["head_out_w"] * 20).astype(np.float32)
    monkeypatch.setenv("RAG_AMD_FFN_FUSED", "1")
    fused = BertModel(cfg, w)
    monkeypatch.delenv("RAG_AMD_FFN_FUSED")
    plain = BertModel(cfg, w)
    rng = np.random.default_rng(55)
    for n_pairs in (40, 97):   # ~1800 and ~4400 tokens, neither a multiple of 32 or 128
        seqs = _seqs(rng, rng.integers(24, 65, size=n_pairs), cfg.vocab_size)
        types = [[0] * 10 + [1] * (len(q) - 10) for q in seqs]
        a, b = fused.classify(seqs, types, sigmoid=False), plain.classify(seqs, types, sigmoid=False)
        want = obert.classify(cfg, w, seqs, types, sigmoid=False)
        assert np.abs(a - b).max() < 0.06, np.abs(a - b).max()   # two fp16 paths, logits spread over +-20: fp16 noise, as against the oracle
        assert np.abs(a - want).max() < 0.08 and np.abs(b - want).max() < 0.08
        ha, hb = fused.hidden_states(seqs, types), plain.hidden_states(seqs, types)
        np.testing.assert_allclose(ha, hb, atol=2e-2, rtol=2e-2)
    fused.close()
    plain.close()


def test_cached_encoder_graphs_survive_a_workspace_reallocation(gpu_required, monkeypatch):
    """A cached graph holds the activation workspace's addresses.  A later pass that needs a bigger workspace — an eager
    embed() of more tokens than the graph path ever takes — frees and reallocates those buffers; the next replay of the
    small shape must not run into the freed ones (round 3's graphs did: advisor finding).  Sequence: capture a small
    shape, grow the workspace well past the graph path's 1024 tokens, replay the small shape several times with other
    models allocating in between, compare with the eager model; then three batches back to back without reading any
    (two staging blocks take turns, the third call waits for the first's upload only)."""
    cfg = _small(BertConfig.minilm_l6())
    cfg.n_layers = 2
    w = random_weights(cfg, 77)
    graphed = BertModel(cfg, w)
    monkeypatch.setenv("RAG_AMD_ENCODER_GRAPH", "0")
    eager = BertModel(cfg, w)
    monkeypatch.delenv("RAG_AMD_ENCODER_GRAPH")
    rng = np.random.default_rng(77)
    small = _seqs(rng, rng.integers(4, 12, size=3), cfg.vocab_size)
    ref_small = eager.embed(small)
    np.testing.assert_allclose(graphed.embed_to_device(small).numpy(), ref_small, atol=3e-6, rtol=1e-5)   # captured
    big = _seqs(rng, rng.integers(60, 100, size=40), cfg.vocab_size)          # ~3200 tokens: the workspace grows
    np.testing.assert_allclose(graphed.embed(big), eager.embed(big), atol=2e-5, rtol=1e-4)
    other = BertModel(cfg, random_weights(cfg, 78))                            # something else takes the freed ranges
    other.embed(big)
    for _ in range(3):
        np.testing.assert_allclose(graphed.embed_to_device(small).numpy(), ref_small, atol=3e-6, rtol=1e-5)
    bigger = _seqs(rng, rng.integers(60, 100, size=90), cfg.vocab_size)       # ~7200 tokens: grows again
    graphed.embed(bigger)
    np.testing.assert_allclose(graphed.embed_to_device(small).numpy(), ref_small, atol=3e-6, rtol=1e-5)
    batches = [_seqs(rng, rng.integers(8, 21, size=32), cfg.vocab_size) for _ in range(3)]
    handles = [graphed.embed_to_device(b) for b in batches]
    for h, b in zip(reversed(handles), reversed(batches)):
        np.testing.assert_allclose(h.numpy(), eager.embed(b), atol=3e-6, rtol=1e-5)
    for m in (graphed, eager, other):
        m.close()


@pytest.mark.parametrize("hidden,heads,inter,head,seed", [
    (256, 8, 672, "none", 41),        # DH 32, N = 672 = 5.25 tiles of 128, K = 256
    (512, 8, 1568, "bert", 42),       # DH 64, intermediate 1568 (not a multiple of 64): K >= 1536 on the way back
    (1024, 16, 2080, "roberta", 43),  # the widest row the kernels take; RoBERTa framing (pos_offset 2, one token type)
    (384, 6, 96, "none", 44),         # DH 64 with a tiny feed-forward (N = 96 < one tile)
])
def test_fp16_fragment_order_path_on_odd_shapes(gpu_required, hidden, heads, inter, head, seed):
    """The fragment-ordered fp16 data path away from the two model shapes it is tuned on: widths that are not whole
    tiles, both head sizes, K on both sides of the two-step kernel's threshold, sequences of several key tiles and a
    token count that is not a whole row block — hidden states, pooled embeddings and (with a head) logits against
    the fp32 oracle within half precision's reach."""
    cfg = BertConfig(hidden=hidden, n_layers=2, n_heads=heads, intermediate=inter, vocab_size=1500, head=head,
                     n_labels=1 if head != "none" else 1, pooling="cls" if seed % 2 else "mean")
    if head == "roberta":
        cfg.max_positions, cfg.type_vocab, cfg.pos_offset, cfg.ln_eps = 514, 1, 2, 1e-5
    cfg.gemm_dtype = "f16"
    w = random_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    seqs = _seqs(rng, [300, 1, 129, 97, 33, 2, 200, 64, 250, 31, 5, 160], cfg.vocab_size)   # 1272 tokens
    assert sum(map(len, seqs)) > 1024 and sum(map(len, seqs)) % 32 != 0
    types = [[0] * len(s) for s in seqs] if cfg.type_vocab == 1 else [[0] * (len(s) // 2) + [1] * (len(s) - len(s) // 2) for s in seqs]
    model = BertModel(cfg, w)
    want_h = np.concatenate(obert.hidden_states(cfg, w, seqs, types))
    got_h = model.hidden_states(seqs, types)
    scale = max(1.0, np.abs(want_h).max())
    assert np.isfinite(got_h).all()
    assert np.abs(got_h - want_h).max() < 3e-2 * scale, np.abs(got_h - want_h).max()
    want_e = obert.embed(cfg, w, seqs, types)
    assert np.abs(model.embed(seqs, types) - want_e).max() < 2e-2
    if head != "none":
        want_l = obert.classify(cfg, w, seqs, types, sigmoid=False)
        got_l = model.classify(seqs, types, sigmoid=False)
        assert np.abs(got_l - want_l).max() < 3e-2 * max(1.0, np.abs(want_l).max())
    model.close()
