"""Worker: FAISSStore in sharded serving mode (rank 0 serves, the others follow); gloo, shared GPU 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    rank, world, port, index_path, out_path, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
    import torch
    import torch.distributed as dist

    from oracle import flat as oracle
    from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
    from rag_inference_pipeline_amd.config import PipelineSettings

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    extra = {}
    if len(sys.argv) > 7:   # "nprobe:<n>": the IVFFlat nprobe mode, every rank holding its share of every list
        extra = dict(RAG_AMD_IVF_MODE="nprobe", FAISS_NPROBE=int(sys.argv[7].split(":")[1]))
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=index_path, faiss_dim=d, **extra))
    store.load()
    assert store.index_size > 0
    if rank == 0:
        results = {}
        for i, (nq, k) in enumerate([(1, 10), (32, 10), (5, 100)]):
            D, I = store.search(oracle.synth_rows(4321 + i, 0, nq, d), k)
            results[f"D{i}"], results[f"I{i}"] = D, I
        np.savez(out_path, **results)
        store.unload()  # releases the followers
    else:
        served = store.serve_forever()
        np.savez(out_path, served=served)
        store.unload()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
