"""CPU: the transformer oracle (oracle/bert.py) against `transformers`' own modules built offline
from a config with seeded random weights — pins the architecture semantics the reference relies on
(BertModel behind SentenceTransformer, *ForSequenceClassification behind the reranker)."""
import numpy as np
import pytest
import torch

from oracle import bert as obert
from rag_inference_pipeline_amd.bert import BertConfig, weights_from_hf_state_dict, pack_sequences, iter_token_budget

transformers = pytest.importorskip("transformers")


def _seqs(rng, n, lo, hi, vocab):
    return [rng.integers(5, vocab, size=int(rng.integers(lo, hi + 1))).tolist() for _ in range(n)]


def _pad(seqs, pad_id=0):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad_id, dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(s)
        mask[i, : len(s)] = 1
    return ids, mask


def _perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():  # move LayerNorm gains/biases and biases off their trivial init
            p.add_(0.05 * torch.randn(p.shape, generator=g))


def test_encoder_matches_hf_bert_model():
    torch.manual_seed(0)
    hf_cfg = transformers.BertConfig(vocab_size=1000, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                                     intermediate_size=128, max_position_embeddings=64, type_vocab_size=2)
    model = transformers.BertModel(hf_cfg, add_pooling_layer=False).eval()
    _perturb(model, 1)
    cfg = BertConfig.from_hf(hf_cfg.to_dict(), head="none")
    assert (cfg.hidden, cfg.n_heads, cfg.pos_offset, cfg.type_vocab) == (64, 2, 0, 2)
    w = weights_from_hf_state_dict(cfg, model.state_dict())
    rng = np.random.default_rng(0)
    seqs = _seqs(rng, 5, 3, 20, 1000)
    types = [[0] * (len(s) // 2) + [1] * (len(s) - len(s) // 2) for s in seqs]
    ids, mask = _pad(seqs)
    tt = torch.zeros_like(ids)
    for i, t in enumerate(types):
        tt[i, : len(t)] = torch.tensor(t)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).last_hidden_state
    got = obert.hidden_states(cfg, w, seqs, types)
    for i, s in enumerate(seqs):
        np.testing.assert_allclose(got[i], ref[i, : len(s)].numpy(), atol=2e-5, rtol=1e-5)
    # sentence-transformers pooling + normalize on top of it
    emb = obert.embed(cfg, w, seqs, types, pooling="mean")
    m = mask[..., None].float()
    st = torch.nn.functional.normalize((ref * m).sum(1) / m.sum(1), dim=1).numpy()
    np.testing.assert_allclose(emb, st, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-6)
    cls = obert.embed(cfg, w, seqs, types, pooling="cls")
    np.testing.assert_allclose(cls, torch.nn.functional.normalize(ref[:, 0], dim=1).numpy(), atol=2e-6)


def test_cross_encoder_matches_hf_bert_for_sequence_classification():
    torch.manual_seed(1)
    hf_cfg = transformers.BertConfig(vocab_size=500, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                                     intermediate_size=128, max_position_embeddings=64, num_labels=1)
    model = transformers.BertForSequenceClassification(hf_cfg).eval()
    _perturb(model, 2)
    cfg = BertConfig.from_hf(hf_cfg.to_dict(), head="bert")
    w = weights_from_hf_state_dict(cfg, model.state_dict())
    rng = np.random.default_rng(1)
    seqs = _seqs(rng, 6, 4, 30, 500)
    types = [[0] * 3 + [1] * (len(s) - 3) for s in seqs]
    ids, mask = _pad(seqs)
    tt = torch.zeros_like(ids)
    for i, t in enumerate(types):
        tt[i, : len(t)] = torch.tensor(t)
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits.view(-1).float()
    np.testing.assert_allclose(obert.classify(cfg, w, seqs, types, sigmoid=False)[:, 0], logits.numpy(), atol=2e-5)
    np.testing.assert_allclose(obert.classify(cfg, w, seqs, types)[:, 0], torch.sigmoid(logits).numpy(), atol=1e-5)


def test_cross_encoder_matches_hf_xlm_roberta_position_offset():
    torch.manual_seed(2)
    hf_cfg = transformers.XLMRobertaConfig(vocab_size=600, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                                           intermediate_size=128, max_position_embeddings=70, type_vocab_size=1,
                                           pad_token_id=1, num_labels=1, layer_norm_eps=1e-5)
    model = transformers.XLMRobertaForSequenceClassification(hf_cfg).eval()
    _perturb(model, 3)
    cfg = BertConfig.from_hf(hf_cfg.to_dict(), head="roberta")
    assert cfg.pos_offset == 2 and cfg.type_vocab == 1
    w = weights_from_hf_state_dict(cfg, model.state_dict())
    rng = np.random.default_rng(2)
    seqs = _seqs(rng, 4, 5, 40, 600)
    ids, mask = _pad(seqs, pad_id=1)
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=mask).logits.view(-1)
    np.testing.assert_allclose(obert.classify(cfg, w, seqs, sigmoid=False)[:, 0], logits.numpy(), atol=2e-5)


def test_rerank_order_is_stable_descending():
    assert obert.rerank_order([0.2, 0.9, 0.2, 0.5]) == [1, 3, 0, 2]
    assert obert.rerank_order([0.2, 0.9, 0.2, 0.5], top_n=2) == [1, 3]
    assert obert.rerank_order([]) == []


def test_pack_and_token_budget_helpers():
    ids, types, cu = pack_sequences([[1, 2, 3], [4], [5, 6]], [[0, 0, 1], [0], [1, 1]])
    assert ids.tolist() == [1, 2, 3, 4, 5, 6] and types.tolist() == [0, 0, 1, 0, 1, 1] and cu.tolist() == [0, 3, 4, 6]
    assert list(iter_token_budget([5, 5, 5, 20, 1], 10)) == [(0, 2), (2, 3), (3, 4), (4, 5)]
    assert list(iter_token_budget([], 10)) == []
