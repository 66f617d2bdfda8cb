"""Time of the fused feed-forward kernel alone (fp16 mode, MiniLM cross-encoder shapes) under its ablation switches.
usage: [RAG_AMD_FFN_ABL=n] python scripts/exp/ffn_fused_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.bert import BertConfig, BertModel, random_weights, pack_sequences
rng = np.random.default_rng(1)
pairs = rng.integers(36, 76, size=3200)
cfg = BertConfig.ms_marco_minilm_l6(); cfg.gemm_dtype = "f16"
m = BertModel(cfg, random_weights(cfg, 0))
seqs = [rng.integers(1000, 30000, size=int(n)).tolist() for n in pairs]
ids, types, cu = pack_sequences(seqs, [[0] * 10 + [1] * (len(q) - 10) for q in seqs])
ids_t, ty_t, cu_t = (torch.from_numpy(a).cuda() for a in (ids, types, cu))
out = torch.empty((len(seqs), 1), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
f = lambda: m.forward_device(ids_t.data_ptr(), ty_t.data_ptr(), cu_t.data_ptr(), len(seqs), int(cu[-1]), int(pairs.max()), _native.BERT_OUT_PROBS, False, out.data_ptr(), st)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
print(f"abl={os.environ.get('RAG_AMD_FFN_ABL', '0')} fused={os.environ.get('RAG_AMD_FFN_FUSED', '1')}: pass {e0.elapsed_time(e1) / 5:.3f} ms")
