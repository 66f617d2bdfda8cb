import os
"""Stage timings of the transformer side (query encoder / cross-encoder) with GEMM flop accounting
and the torch-CPU oracle timed beside each stage on a bounded sample.

    python scripts/bench_stages.py [--json profiles/rNN_stages.json] [--no-cpu]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.bert import BertConfig, BertModel, random_weights, pack_sequences

FP32_MFMA_PEAK_TF = 157.3
BF16_MFMA_PEAK_TF = 2500.0   # dense (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)
BIG_M = 1024                 # tokens above which the GEMMs take the big-batch kernels (rag_bert.hip launch_gemm)


def mode_peak(gemm_dtype, tokens):
    """(peak TF/s in fp32-equivalent GEMM flops, what bounds it) of the GEMM path a stage actually runs on:
    small batches stay on the fp32 MFMA whatever the mode; big batches run three fp16 MFMAs per product in the
    default mode (two-plane split, 2.5 PF / 3), one fp16 MFMA in f16 mode, the fp32 MFMA in strict mode."""
    if tokens <= BIG_M or gemm_dtype == "f32_strict":
        return FP32_MFMA_PEAK_TF, "fp32 MFMA (157.3 TF/s)"
    if gemm_dtype == "f16":
        return BF16_MFMA_PEAK_TF, "fp16 MFMA (2.5 PF/s dense)"
    return BF16_MFMA_PEAK_TF / 3.0, "fp16 MFMA, three products per fp32 product (2.5 PF/s / 3 = 833 TF/s fp32-equivalent)"


def gemm_flops(cfg, T, nseq, head, first_only=False):
    """GEMM flops the forward pass EXECUTES.  With an output that reads first tokens only (CLS pooling, classifier
    head) the last layer's output projection and feed-forward run on nseq rows, not T (rag_bert.hip)."""
    H, I = cfg.hidden, cfg.intermediate
    per_token = 2 * (3 * H * H + H * H + 2 * H * I)
    total = per_token * cfg.n_layers * T + (2 * H * H * nseq if head else 0)
    if first_only and nseq < T:
        total -= 2 * (H * H + 2 * H * I) * (T - nseq)
    return total


def attn_flops(cfg, lens):
    return sum(4 * int(L) * int(L) * cfg.hidden for L in lens) * cfg.n_layers


def cpu_leg(cfg, w, seqs, out_kind, sample, threads):
    from oracle import bert as obert
    torch.set_num_threads(threads)
    sub = seqs[:sample]
    fn = (lambda: obert.classify(cfg, w, sub)) if out_kind == _native.BERT_OUT_PROBS else (lambda: obert.embed(cfg, w, sub))
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn(); reps += 1
        el = time.perf_counter() - t0
        if el > 5.0 or reps >= 20:
            break
    return {"value": len(sub) * reps / el, "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"oracle/bert.py (torch-CPU fp32, padded batch) on the first {len(sub)} of {len(seqs)} sequences, "
                      f"{reps} reps at {el / reps * 1e3:.1f} ms"}


def run(name, cfg, lens, out_kind, reps=10, cpu_sample=0, threads=16, gemm_dtype="f32"):
    cfg.vocab_size = min(cfg.vocab_size, 30522)
    cfg.gemm_dtype = gemm_dtype
    w = random_weights(cfg, 0)
    model = BertModel(cfg, w)
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
    for _ in range(reps):
        go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = gemm_flops(cfg, T, nseq, cfg.head != "none", first_only=out_kind != _native.BERT_OUT_MEAN and out_kind != _native.BERT_OUT_HIDDEN)
    res = {"stage": name, "nseq": nseq, "tokens": T, "max_len": L, "ms_per_batch": ms, "gemm_gflop": gf / 1e9,
           "attention_gflop": attn_flops(cfg, lens) / 1e9, "gemm_tflops_end_to_end": gf / ms / 1e9,
           "mfma_peak_tflops_of_this_mode": mode_peak(gemm_dtype, T)[0], "mfma_peak_basis": mode_peak(gemm_dtype, T)[1],
           "frac_of_mode_mfma_peak": gf / ms / 1e9 / mode_peak(gemm_dtype, T)[0], "sequences_per_s": nseq / ms * 1e3,
           "dtype": {"f32": "f32 (big-batch GEMMs: two fp16 planes per operand, 3 products, on the fp16 matrix cores)",
                     "f32_strict": "f32 on the fp32 MFMA throughout",
                     "f16": "fp16 activations in MFMA-fragment order above 1024 tokens (GEMMs and attention on the fp16 matrix cores), "
                            "f32 accumulation, softmax and LayerNorm statistics"}[gemm_dtype]}
    print(f"{name}: nseq={nseq} tokens={T} maxlen={L}: {ms:.3f} ms/batch  GEMM {gf/1e9:.1f} GFLOP -> {gf/ms/1e9:.1f} TF/s "
          f"(attn {res['attention_gflop']:.2f} GFLOP)  {nseq/ms*1e3:.0f} seq/s", flush=True)
    model.close()
    if cpu_sample:
        res["cpu_baseline"] = cpu_leg(cfg, w, seqs, out_kind, cpu_sample, threads)
        print(f"    cpu: {res['cpu_baseline']['value']:.0f} seq/s on {threads} threads", flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(16, avail))
    c = 0 if a.no_cpu else 1
    rng = np.random.default_rng(1)
    q = rng.integers(8, 21, size=32)
    pairs = rng.integers(24 + 12, 64 + 12, size=3200)
    out = [
        run("query encoder, all-MiniLM-L6-v2 arch, 32 queries", BertConfig.minilm_l6(), q, _native.BERT_OUT_MEAN, cpu_sample=32 * c, threads=threads),
        run("query encoder, bge-base-en-v1.5 arch, 32 queries", BertConfig.bge_base(), q, _native.BERT_OUT_CLS, cpu_sample=32 * c, threads=threads),
        run("cross-encoder, ms-marco-MiniLM-L-6-v2 arch, 32 queries x 100 docs", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=5, cpu_sample=200 * c, threads=threads),
        run("cross-encoder, ms-marco-MiniLM-L-6-v2 arch, 32 queries x 10 docs", BertConfig.ms_marco_minilm_l6(), pairs[:320], _native.BERT_OUT_PROBS, reps=5),
        run("cross-encoder, bge-reranker-base arch, 32 queries x 100 docs", BertConfig.bge_reranker_base(), pairs, _native.BERT_OUT_PROBS, reps=3, cpu_sample=100 * c, threads=threads),
        run("cross-encoder (fp32 MFMA only), ms-marco-MiniLM-L-6-v2 arch, 32 x 100 docs", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=5, gemm_dtype="f32_strict"),
        run("cross-encoder (fp32 MFMA only), bge-reranker-base arch, 32 x 100 docs", BertConfig.bge_reranker_base(), pairs, _native.BERT_OUT_PROBS, reps=3, gemm_dtype="f32_strict"),
        run("cross-encoder (fp16 GEMM mode), ms-marco-MiniLM-L-6-v2 arch, 32 x 100 docs", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=5, gemm_dtype="f16"),
        run("cross-encoder (fp16 GEMM mode), bge-reranker-base arch, 32 x 100 docs", BertConfig.bge_reranker_base(), pairs, _native.BERT_OUT_PROBS, reps=3, gemm_dtype="f16"),
    ]
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TF, "bf16_mfma_dense_peak_tflops": BF16_MFMA_PEAK_TF,
                       "note": "frac_of_mode_mfma_peak = end-to-end GEMM TF/s (attention, LayerNorm, launches included in the time) "
                               "over the peak of the matrix-core path that mode runs on; never above 1", "stages": out}, fh, indent=1)
