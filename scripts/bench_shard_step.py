"""One shard's step of the 8-GPU series, on one GPU: 1.25M x 768 rows through ShardedFlatIndex over a
real nccl process group of world size 1 (scan -> merge -> pack -> all_gather -> merge), timed like
bench.py.  The N-GPU step is this plus the collective's latency over xGMI."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16
from rag_inference_pipeline_amd.sharded import ShardedFlatIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
steps = 200
idx = FlatIndex(768); idx.add_synthetic(rows, 1234)
sh = ShardedFlatIndex(idx, 0, device=0)
Q = torch.from_numpy(oracle.synth_rows(4321, 0, 32, 768)).cuda()
for mode in ("one-pass", "two-stage"):
    if mode == "two-stage":
        idx.set_screening(SCREEN_FP16)
    for _ in range(10):
        sh.search_tensors(Q, 10)
    torch.cuda.synchronize()
    idx.profile_enable(True); idx.profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sh.search_tensors(Q, 10)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ms, n = idx.profile(reset=True); idx.profile_enable(False)
    print(f"{mode}: rows={rows} step {el*1e3:.3f} ms, scan kernel {ms/n:.3f} ms", flush=True)
dist.destroy_process_group()
