import os, sys, numpy as np
sys.path.insert(0, ".")
import torch
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex, merge_topk_device
n, d, nq, k = 4000, 64, 4, 5
X = oracle.synth_rows(1234, 0, n, d); Q = oracle.synth_rows(4321, 0, nq, d)
parts = []
for lo, hi in [(0, 2000), (2000, 4000)]:
    idx = FlatIndex(d); idx.add(X[lo:hi]); idx.set_id_offset(lo)
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    idx.search_device(torch.from_numpy(Q).cuda().data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("part", lo, i[0].tolist(), s[0].tolist())
    parts.append((s, i))
all_s = torch.stack([p[0] for p in parts]).contiguous(); all_i = torch.stack([p[1] for p in parts]).contiguous()
out_s = torch.full((nq, k), -7.0, dtype=torch.float32, device="cuda"); out_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
merge_topk_device(0, 0, 2, nq, k, all_s.data_ptr(), all_i.data_ptr(), out_s.data_ptr(), out_i.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("merged", out_i[0].tolist(), out_s[0].tolist())
D, I = oracle.search(X, Q, k); print("oracle", I[0].tolist(), D[0].tolist())
