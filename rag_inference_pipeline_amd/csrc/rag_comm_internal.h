// rag_comm_internal.h — what rag_amd.hip needs of the communicator (rag_comm.hip) for the fused shard step.
#pragma once
#include <cstddef>

#include "rag_common.h"

struct rag_comm;
// ncclAllGather of bytes_per_rank bytes per rank on `st` (the caller holds no lock of the communicator: RCCL calls on
// one communicator are issued by one thread at a time — the index's mutex and the serving channel's lock see to that).
int ragc_comm_all_gather(rag_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, hipStream_t st);
int ragc_comm_device(const rag_comm* c);
int ragc_comm_world(const rag_comm* c);
