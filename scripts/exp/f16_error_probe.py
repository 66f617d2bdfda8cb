"""Error of the RAG_GEMM_F16 variants against the fp32 oracle on the cross-encoder test's inputs (several seeds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import bert as obert
from rag_inference_pipeline_amd.bert import BertConfig, BertModel, random_weights

def model(cfg, w, env):
    for k, v in env.items(): os.environ[k] = v
    m = BertModel(cfg, w)
    for k in env: del os.environ[k]
    return m

for seed in (5, 6, 7, 8):
    cfg = BertConfig.ms_marco_minilm_l6(); cfg.vocab_size = 2000
    w = random_weights(cfg, seed)
    w["head_out_w"] = (w["head_out_w"] * 20).astype(np.float32)
    rng = np.random.default_rng(seed)
    seqs = [rng.integers(3, cfg.vocab_size, size=int(n)).tolist() for n in rng.integers(24, 65, size=64)]
    types = [[0] * 10 + [1] * (len(s) - 10) for s in seqs]
    want = obert.classify(cfg, w, seqs, types, sigmoid=False)
    wantp = obert.classify(cfg, w, seqs, types)
    cfg.gemm_dtype = "f16"
    for name, env in (("row-major fp32 activations", {"RAG_AMD_ROW_MAJOR": "1"}), ("tiled, fp32-MFMA attention", {"RAG_AMD_TILED_ATTENTION_F32": "1"}),
                      ("tiled, fp16-MFMA attention", {})):
        m = model(cfg, w, env)
        got = m.classify(seqs, types, sigmoid=False); gotp = m.classify(seqs, types)
        m.close()
        print(f"seed {seed} {name:30s} logits max|err| {np.abs(got - want).max():.4f} mean {np.abs(got - want).mean():.4f} (max|logit| {np.abs(want).max():.2f})"
              f"  probs max|err| {np.abs(gotp - wantp).max():.5f}", flush=True)
