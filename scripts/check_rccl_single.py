"""RCCL smoke on ONE rank: the exact collective calls the sharded path makes (uint8 all_gather_into_tensor,
float64 all_reduce MAX, barrier) through a real nccl process group, then a sharded search with world=1
geometry forced through ShardedFlatIndex to exercise the packed merge on the NCCL code path."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex
from rag_inference_pipeline_amd.sharded import ShardedFlatIndex
X = oracle.synth_rows(1234, 0, 20000, 384); Q = oracle.synth_rows(4321, 0, 32, 384)
idx = FlatIndex(384); idx.add(X)
sh = ShardedFlatIndex(idx, 0, device=0)
assert sh.backend == "nccl" and sh.world == 1
D, I = sh.search(Q, 10)
Do, Io = oracle.search(X, Q, 10)
assert (I == Io).all() and (D == Do).all()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print("rccl single-rank path ok:", sh.backend, float(t.item()))
dist.destroy_process_group()
