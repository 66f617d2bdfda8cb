"""Worker: a sharded IVFFlat index in the nprobe mode over RCCL (backend "nccl", rank r on GPU r) — every rank holds its share
of every list, the shard step is ONE C-ABI call on the library's own communicator (rag_ivf_search_gather_device).

argv: rank world port out_path n d nlist nprobe k metric"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_lists(n, d, nlist, metric, nprobe):
    """The same lists on every rank and in the test (seeded): any partition of the rows is a valid set of lists."""
    from rag_inference_pipeline_amd.index_io import IVFFlatLists
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    cent = np.ascontiguousarray(X[rng.integers(0, n, size=nlist)] + np.float32(0.05) * rng.standard_normal((nlist, d), dtype=np.float32))
    assign = np.sort(rng.integers(0, nlist, size=n))
    offsets = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(np.bincount(assign, minlength=nlist), out=offsets[1:])
    ids = rng.permutation(n).astype(np.int64)
    Q = rng.standard_normal((32, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    return IVFFlatLists(cent, 1, X, ids, offsets, metric, nprobe), Q


def main() -> None:
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    n, d, nlist, nprobe, k, metric = (int(a) for a in sys.argv[5:11])
    import torch
    import torch.distributed as dist

    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    from rag_inference_pipeline_amd.sharded import ShardedFlatIndex

    torch.cuda.set_device(rank % torch.cuda.device_count())
    dev = torch.cuda.current_device()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device("cuda", dev))
    lists, Q = build_lists(n, d, nlist, metric, nprobe)
    local = IVFFlatIndex(lists.shard(rank, world), device=dev, nprobe=nprobe)
    sharded = ShardedFlatIndex(local, metric, device=dev)
    assert sharded.backend == "nccl" and sharded.own_rccl == (os.environ.get("RAG_AMD_OWN_RCCL", "1") != "0")
    own = int(sharded.own_rccl)
    D, I = sharded.search(Q, k)
    Qr = np.ascontiguousarray(Q[::-1])
    qa, qb = torch.from_numpy(Q).to(sharded.device), torch.from_numpy(Qr).to(sharded.device)
    over = ShardedFlatIndex(local, metric, device=dev, overlap_collective=True)   # all-gather + merge on a second stream
    ta, tb = over.submit(qa, k), over.submit(qb, k)
    over.collect(ta)
    D4, I4 = ta.host_s.numpy().copy(), ta.host_i.numpy().copy()
    over.collect(tb)
    D5, I5 = tb.host_s.numpy().copy(), tb.host_i.numpy().copy()
    over.close()
    extra = {}
    if rank == 0:
        D3, I3 = sharded.leader_search(Q[:5], k)
        sharded.shutdown()
        extra.update(D3=D3, I3=I3)
    else:
        extra.update(served=sharded.follower_loop())
    sharded.close()
    np.savez(out_path, D=D, I=I, D4=D4, I4=I4, D5=D5, I5=I5, own=own, two_stage=int(local.two_stage), **extra)
    local.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
