"""One shard's step of the 8-GPU series, on one GPU: 1.25M x 768 rows, batch 32, k = 10.

Two figures per search mode, timed like bench.py (device-resident queries, back-to-back steps):
  local    rag_index_search_device alone: scan -> merge (one-pass) or prep -> screen -> resolve -> fallback
           (two-stage).  This is the step "before the collective".
  sharded  the same through ShardedFlatIndex over a real nccl process group of world size 1
           (... -> all_gather_into_tensor -> merge of the gathered lists): what every rank of the N-GPU
           run executes; the N-GPU step is this plus the collective's latency over xGMI.
Prints one JSON object (committed as profiles/r02_shard_step.json)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16
from rag_inference_pipeline_amd.sharded import ShardedFlatIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d, B, k = 768, 32, 10
idx = FlatIndex(d); idx.add_synthetic(rows, 1234)
sh = ShardedFlatIndex(idx, 0, device=0)
Q = torch.from_numpy(oracle.synth_rows(4321, 0, B, d)).cuda()
out_s = torch.empty((B, k), dtype=torch.float32, device="cuda")
out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
sptr = torch.cuda.current_stream().cuda_stream


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    idx.profile_enable(True); idx.profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ms, n = idx.profile(reset=True); idx.profile_enable(False)
    return el * 1e3, ms / n


res = {"rows": rows, "dim": d, "batch": B, "k": k, "steps": steps}
ref = None
for mode in ("one_pass", "two_stage"):
    if mode == "two_stage":
        idx.set_screening(SCREEN_FP16)
    local_ms, scan_ms = timed(lambda: idx.search_device(Q.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), sptr))
    torch.cuda.synchronize()
    ids = out_i.cpu().numpy().copy(); sc = out_s.cpu().numpy().copy()
    sharded_ms, _ = timed(lambda: sh.search_tensors(Q, k))
    s2, i2 = sh.search_tensors(Q, k); torch.cuda.synchronize()
    same = bool(np.array_equal(i2.cpu().numpy(), ids) and np.array_equal(s2.cpu().numpy().view(np.uint32), sc.view(np.uint32)))
    if ref is None:
        ref = (ids, sc)
    res[mode] = {"local_step_ms": round(local_ms, 4), "scan_kernel_ms": round(scan_ms, 4),
                 "sharded_world1_step_ms": round(sharded_ms, 4),
                 "scan_GBps": round((4.0 if mode == "one_pass" else 2.0) * rows * d / (scan_ms * 1e-3) / 1e9, 1),
                 "sharded_equals_local": same,
                 "identical_to_one_pass": bool(np.array_equal(ids, ref[0]) and np.array_equal(sc.view(np.uint32), ref[1].view(np.uint32)))}
if idx.screening == SCREEN_FP16:
    res["two_stage"]["fallbacks"] = idx.screen_stats()["fallbacks"]
print(json.dumps(res), flush=True)
dist.destroy_process_group()
