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
bs.run("rerank ms-marco-MiniLM (3200 pairs)", BertConfig.ms_marco_minilm_l6(), pairs, _native.BERT_OUT_PROBS, reps=3)
