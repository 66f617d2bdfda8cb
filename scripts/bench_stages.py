"""Stage timings of the transformer side (encoder / cross-encoder) with GEMM flop accounting."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.bert import BertConfig, BertModel, random_weights, pack_sequences

def gemm_flops(cfg, T, nseq, head):
    H, I = cfg.hidden, cfg.intermediate
    per_tok = 2 * (3 * H * H + H * H + 2 * H * I) * cfg.n_layers
    return per_tok * T + (2 * H * H * nseq if head else 0)

def attn_flops(cfg, lens):
    return sum(4 * L * L * cfg.hidden for L in lens) * cfg.n_layers

def run(name, cfg, lens, out_kind, reps=10):
    cfg.vocab_size = min(cfg.vocab_size, 30522)
    model = BertModel(cfg, random_weights(cfg, 0))
    rng = np.random.default_rng(0)
    seqs = [rng.integers(5, cfg.vocab_size, size=int(n)).tolist() for n in lens]
    ids, types, cu = pack_sequences(seqs, [[0] * len(s) for s in seqs])
    dev = torch.device("cuda")
    ids_t, types_t, cu_t = (torch.from_numpy(a).to(dev) for a in (ids, types, cu))
    nseq, T, L = len(seqs), int(cu[-1]), int(max(lens))
    width = cfg.hidden if out_kind in (_native.BERT_OUT_MEAN, _native.BERT_OUT_CLS) else cfg.n_labels
    out = torch.empty((nseq, width), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def go():
        model.forward_device(ids_t.data_ptr(), types_t.data_ptr() if cfg.type_vocab > 1 else 0, cu_t.data_ptr(), nseq, T, L,
                             out_kind, True, out.data_ptr(), st)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = gemm_flops(cfg, T, nseq, cfg.head != "none")
    print(f"{name}: nseq={nseq} tokens={T} maxlen={L}: {ms:.3f} ms/batch  GEMM {gf/1e9:.1f} GFLOP -> {gf/ms/1e9:.1f} TF/s "
          f"(attn {attn_flops(cfg, lens)/1e9:.2f} GFLOP)  {nseq/ms*1e3:.0f} seq/s", flush=True)
    model.close()

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    q = rng.integers(8, 21, size=32)
    run("encoder MiniLM-L6 (32 queries)", BertConfig.minilm_l6(), q, _native.BERT_OUT_MEAN)
    run("encoder bge-base (32 queries)", BertConfig.bge_base(), q, _native.BERT_OUT_CLS)
    pairs = rng.integers(24 + 12, 64 + 12, size=3200)
    run("rerank ms-marco-MiniLM (3200 pairs)", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=5)
    run("rerank ms-marco-MiniLM (320 pairs)", BertConfig.ms_marco_minilm_l6(), pairs[:320], _native.BERT_OUT_PROBS, reps=5)
    run("rerank bge-reranker-base arch (3200 pairs)", BertConfig.bge_reranker_base(), pairs, _native.BERT_OUT_PROBS, reps=3)
