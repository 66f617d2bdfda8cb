"""Worker for the multi-rank tests (launched once per rank by test_sharded.py).

argv: mode (cpu|gpu) rank world port out_path n d nq k metric
  cpu: the local index is a test double backed by the ORACLE (tests may use the oracle; the product
       path never does) and the merge is oracle.merge — this rehearses sharding, packing and the
       gloo all-gather with no GPU.
  gpu: the real FlatIndex and the device merge kernel; ranks share GPU 0, gather staged over gloo.
  nccl: the product's multi-GPU configuration — backend "nccl" (= RCCL), rank r on GPU r, the packed
       buffer gathered device-to-device by all_gather_into_tensor and merged in place by
       rag_merge_topk_packed_device; also drives the leader path (head + query broadcast on the device).
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    mode, rank, world, port, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    n, d, nq, k, metric = (int(a) for a in sys.argv[6:11])
    import torch
    import torch.distributed as dist

    from oracle import flat as oracle
    from rag_inference_pipeline_amd.sharded import ShardedFlatIndex, shard_range

    if mode == "nccl":
        torch.cuda.set_device(rank % torch.cuda.device_count())
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    X_local = oracle.synth_rows(1234, lo, hi - lo, d)
    Q = oracle.synth_rows(4321, 0, nq, d)

    if mode in ("cpu", "cpuflag"):
        class OracleLocal:
            def search_device(self, q_ptr, nq_, k_, s_ptr, i_ptr, stream=0):
                q = np.ctypeslib.as_array(ctypes.cast(q_ptr, ctypes.POINTER(ctypes.c_float)), (nq_, d))
                D, I = oracle.search(X_local, q, k_, metric, id_offset=lo)
                np.ctypeslib.as_array(ctypes.cast(s_ptr, ctypes.POINTER(ctypes.c_float)), (nq_, k_))[:] = D
                np.ctypeslib.as_array(ctypes.cast(i_ptr, ctypes.POINTER(ctypes.c_int64)), (nq_, k_))[:] = I

        def merge(metric_, all_s, all_i, out_s, out_i):
            D, I = oracle.merge(all_s.numpy(), all_i.numpy(), metric_)
            out_s.copy_(torch.from_numpy(D))
            out_i.copy_(torch.from_numpy(I))

        class FlaggingLocal(OracleLocal):
            """A local index with the two-stage choice (rag_index_search_device_ex's contract): in
            RAG_SEARCH_DEFER_FALLBACK mode the LAST rank's first two searches report "not final" and leave
            garbage in the lists; RAG_SEARCH_EXACT_ONE_PASS returns the truth and clears the word."""
            deferred_calls = 0

            def search_device_ex(self, q_ptr, nq_, k_, s_ptr, i_ptr, mode_, flag_ptr, stream=0):
                flag = np.ctypeslib.as_array(ctypes.cast(flag_ptr, ctypes.POINTER(ctypes.c_uint32)), (1,))
                self.search_device(q_ptr, nq_, k_, s_ptr, i_ptr)
                flag[0] = 0
                if mode_ == 2:  # SEARCH_DEFER_FALLBACK
                    self.deferred_calls += 1
                    if rank == world - 1 and self.deferred_calls <= 2:
                        np.ctypeslib.as_array(ctypes.cast(i_ptr, ctypes.POINTER(ctypes.c_int64)), (nq_, k_))[:] = 7
                        np.ctypeslib.as_array(ctypes.cast(s_ptr, ctypes.POINTER(ctypes.c_float)), (nq_, k_))[:] = 1e30
                        flag[0] = 1

        local = FlaggingLocal() if mode == "cpuflag" else OracleLocal()
        sharded = ShardedFlatIndex(local, metric, device="cpu", merge=merge, dim=d)
    else:
        from rag_inference_pipeline_amd.flat_index import FlatIndex

        dev = torch.cuda.current_device() if mode == "nccl" else 0
        torch.cuda.set_device(dev)
        local = FlatIndex(d, metric, device=dev)
        local.add_synthetic(hi - lo, 1234, row_number_offset=lo)
        local.set_id_offset(lo)
        sharded = ShardedFlatIndex(local, metric, device=dev)
        assert sharded.backend == ("nccl" if mode == "nccl" else "gloo"), sharded.backend

    extra = {}
    if mode == "cpuflag":
        # two searches in flight (submit / collect), both flagged by the last rank: each is repeated through the
        # "fp32 scan" on EVERY rank (the flag is the same word everywhere) and comes out right
        Qa, Qb = torch.from_numpy(Q.copy()), torch.from_numpy(Q[::-1].copy())
        ta, tb = sharded.submit(Qa, k), sharded.submit(Qb, k)
        sa, ia = (t.clone() for t in sharded.collect(ta))
        sb, ib = (t.clone() for t in sharded.collect(tb))
        extra = dict(Da=sa.numpy(), Ia=ia.numpy(), Db=sb.numpy(), Ib=ib.numpy(), repeats_pipelined=sharded.repeats)
    D, I = sharded.search(Q, k)
    D2, I2 = sharded.search(Q[: max(1, nq // 2)], k)  # a second shape re-uses the group
    if mode == "cpuflag":
        extra["repeats_total"] = sharded.repeats
    if mode == "nccl":
        # C1: the step runs on the library's own RCCL communicator unless the environment turns that off
        want_own = os.environ.get("RAG_AMD_OWN_RCCL", "1") != "0"
        assert sharded.own_rccl == want_own, (sharded.own_rccl, want_own)
        # two-stage local searches, two batches in flight, then the same with all-gather + merge on a second stream
        local.set_screening(True)
        Qr = np.ascontiguousarray(Q[::-1])
        qa, qb = torch.from_numpy(Q).to(sharded.device), torch.from_numpy(Qr).to(sharded.device)
        ta, tb = sharded.submit(qa, k), sharded.submit(qb, k)
        sharded.collect(ta)
        D4, I4 = ta.host_s.numpy().copy(), ta.host_i.numpy().copy()
        sharded.collect(tb)
        D5, I5 = tb.host_s.numpy().copy(), tb.host_i.numpy().copy()
        over = ShardedFlatIndex(local, metric, device=dev, overlap_collective=True)
        assert (over._comm_stream is not None) == want_own
        ta, tb = over.submit(qa, k), over.submit(qb, k)
        over.collect(ta)
        D6, I6 = ta.host_s.numpy().copy(), ta.host_i.numpy().copy()
        over.collect(tb)
        D7, I7 = tb.host_s.numpy().copy(), tb.host_i.numpy().copy()
        over.close()
        extra = dict(D4=D4, I4=I4, D5=D5, I5=I5, D6=D6, I6=I6, D7=D7, I7=I7, own=int(sharded.own_rccl))
    if mode == "nccl":  # the serving protocol over RCCL: control words and the batch travel on the device
        if rank == 0:
            D3, I3 = sharded.leader_search(Q, k)
            D8, I8 = sharded.leader_search(Q[:3], k)   # a second request: the sequence number moves on
            sharded.shutdown()
            extra.update(D3=D3, I3=I3, D8=D8, I8=I8)
        else:
            extra.update(served=sharded.follower_loop())
        sharded.close()
    np.savez(out_path, D=D, I=I, D2=D2, I2=I2, lo=lo, hi=hi, **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
