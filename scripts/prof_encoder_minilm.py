import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x"]
import numpy as np
import scripts.bench_stages as bs
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.bert import BertConfig
rng = np.random.default_rng(1)
q = rng.integers(8, 21, size=32)
bs.run("encoder all-MiniLM-L6 (32 queries)", BertConfig.minilm_l6(), q, _native.BERT_OUT_MEAN, reps=20)
