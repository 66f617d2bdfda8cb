"""The cross-encoder pass in RAG_GEMM_F16 mode (fp16 activations in fragment order), for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d … -- python3 scripts/prof_rerank_f16.py [base]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scripts.bench_stages as bs
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.bert import BertConfig
rng = np.random.default_rng(1)
rng.integers(8, 21, size=32)
pairs = rng.integers(24 + 12, 64 + 12, size=3200)
if len(sys.argv) > 1 and sys.argv[1] == "base":
    bs.run("rerank bge-reranker-base, f16 mode (3200 pairs)", BertConfig.bge_reranker_base(), pairs, _native.BERT_OUT_PROBS, reps=3, gemm_dtype="f16")
else:
    bs.run("rerank ms-marco-MiniLM, f16 mode (3200 pairs)", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=3, gemm_dtype="f16")
