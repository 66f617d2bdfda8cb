// rag_comm.hip — C1 of SURVEY §8a: the shard step's collectives on an OWN RCCL communicator (rag_comm_* in
// include/rag_amd.h).  One process per GPU; torch.distributed (or any other bootstrap) only carries the 128-byte
// unique id from rank 0 to the others — every collective of the data path is enqueued here, on the caller's stream,
// between the kernels it orders: no second stream, no cross-stream events, no Python per collective.
//
// RCCL is bound at run time (dlopen), not at link time: a PyTorch-ROCm process already holds a copy of librccl (and of
// the HIP runtime it was built against) and a second copy in the same process would bring a second runtime with it.
// Search order of rag_comm_runtime(NULL): $RAG_AMD_RCCL_LIB, a copy already mapped into the process ("librccl.so",
// "librccl.so.1" with RTLD_NOLOAD), then "librccl.so.1" / "/opt/rocm/lib/librccl.so.1" from the loader's path.
#include <dlfcn.h>
#include <sched.h>
#include <time.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "rag_comm_internal.h"

namespace {

// The handful of RCCL types this file needs, restated so that the build does not depend on which rccl.h is installed
// (they are ABI constants of NCCL 2.x: a 128-byte id, an opaque communicator pointer, int-sized enums).
struct NcclUniqueId {
    char internal[RAG_COMM_ID_BYTES];
};
using NcclComm = void*;
constexpr int kNcclSuccess = 0;
constexpr int kNcclUint8 = 1;   // ncclUint8

struct Rccl {
    void* dl = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommAbort)(NcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int version = 0;
    char path[512] = "";
};

Rccl g_rccl;
std::mutex g_rccl_mu;

template <class F>
bool bind(void* dl, const char* name, F* out) {
    *out = reinterpret_cast<F>(dlsym(dl, name));
    return *out != nullptr;
}

bool bind_all(void* dl, Rccl* r) {
    return bind(dl, "ncclGetVersion", &r->GetVersion) && bind(dl, "ncclGetUniqueId", &r->GetUniqueId) &&
           bind(dl, "ncclCommInitRank", &r->CommInitRank) && bind(dl, "ncclCommDestroy", &r->CommDestroy) &&
           bind(dl, "ncclCommAbort", &r->CommAbort) && bind(dl, "ncclAllGather", &r->AllGather) &&
           bind(dl, "ncclBroadcast", &r->Broadcast) && bind(dl, "ncclGetErrorString", &r->GetErrorString);
}

int load_rccl_locked(const char* path) {
    if (g_rccl.dl) {
        if (path && *path && std::strcmp(path, g_rccl.path) != 0)
            return ragc_fail(RAG_ERR_STATE, "RCCL is already bound to %s", g_rccl.path);
        return RAG_OK;
    }
    struct Try {
        const char* name;
        int flags;
    };
    const char* env = getenv("RAG_AMD_RCCL_LIB");
    const Try order[] = {
        {path && *path ? path : nullptr, RTLD_NOW | RTLD_LOCAL},
        {env && *env ? env : nullptr, RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},      // the copy a PyTorch-ROCm process has mapped
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
    };
    char last_err[256] = "no candidate";
    for (const Try& t : order) {
        if (!t.name) continue;
        void* dl = dlopen(t.name, t.flags);
        if (!dl) {
            const char* e = dlerror();
            if (e && !(t.flags & RTLD_NOLOAD)) snprintf(last_err, sizeof last_err, "%s", e);
            if (t.name == path) return ragc_fail(RAG_ERR_STATE, "cannot load RCCL from %s: %s", path, last_err);
            continue;
        }
        Rccl r;
        if (!bind_all(dl, &r)) {
            snprintf(last_err, sizeof last_err, "%s lacks an RCCL entry point", t.name);
            (void)dlclose(dl);
            if (t.name == path) return ragc_fail(RAG_ERR_STATE, "%s", last_err);
            continue;
        }
        r.dl = dl;
        if (r.GetVersion(&r.version) != kNcclSuccess) r.version = 0;
        snprintf(r.path, sizeof r.path, "%s", t.name);
        g_rccl = r;
        return RAG_OK;
    }
    return ragc_fail(RAG_ERR_STATE, "RCCL not found (%s); set RAG_AMD_RCCL_LIB", last_err);
}

int nccl_fail(const char* what, int rc) {
    return ragc_fail(RAG_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

// After the broadcast: the head of the request and its sequence number into pinned host memory (posted stores; the
// sequence word last, behind a system-scope fence), so a follower's host thread sees a request the moment it has
// landed — it polls one word instead of paying a stream synchronisation.
__global__ void post_head_kernel(const unsigned long long* msg, unsigned long long* mirror, int head_words, unsigned long long seq) {
    const int t = threadIdx.x;
    if (t < head_words) mirror[t] = msg[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(&mirror[RAG_COMM_HEAD_WORDS], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

struct rag_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
    std::mutex mu;
};

int ragc_comm_all_gather(rag_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, hipStream_t st) {
    if (!c || !c->comm) return ragc_fail(RAG_ERR_INVALID_ARG, "null communicator");
    const int rc = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, c->comm, st);
    if (rc != kNcclSuccess) return nccl_fail("ncclAllGather", rc);
    return RAG_OK;
}

int ragc_comm_device(const rag_comm* c) { return c ? c->device : -1; }
int ragc_comm_world(const rag_comm* c) { return c ? c->world : 0; }

extern "C" int rag_comm_runtime(const char* librccl_path, int32_t* version_out) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    int rc = load_rccl_locked(librccl_path);
    if (rc) return rc;
    if (version_out) *version_out = g_rccl.version;
    return RAG_OK;
}

extern "C" int rag_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return ragc_fail(RAG_ERR_INVALID_ARG, "id_out is null");
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        int rc = load_rccl_locked(nullptr);
        if (rc) return rc;
    }
    NcclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != kNcclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, RAG_COMM_ID_BYTES);
    return RAG_OK;
}

extern "C" int rag_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, rag_comm** out) {
    if (!out) return ragc_fail(RAG_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!id || world <= 0 || rank < 0 || rank >= world) return ragc_fail(RAG_ERR_INVALID_ARG, "bad rank %d / world %d", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ragc_fail(RAG_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ragc_fail(RAG_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        int rc = load_rccl_locked(nullptr);
        if (rc) return rc;
    }
    RagcDeviceGuard g(device);
    if (!g.ok) return ragc_fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    rag_comm* c = new (std::nothrow) rag_comm();
    if (!c) return ragc_fail(RAG_ERR_OOM, "host allocation failed");
    c->rank = rank;
    c->world = world;
    c->device = device;
    NcclUniqueId uid;
    std::memcpy(uid.internal, id, RAG_COMM_ID_BYTES);
    const int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);   // collective: returns once every rank has joined
    if (rc != kNcclSuccess) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return RAG_OK;
}

extern "C" int rag_comm_destroy(rag_comm* c) {
    if (!c) return RAG_OK;
    {
        RagcDeviceGuard g(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        (void)hipDeviceSynchronize();   // nothing of ours may still be inside a collective
        if (c->comm) (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    delete c;
    return RAG_OK;
}

extern "C" int32_t rag_comm_rank(const rag_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t rag_comm_world(const rag_comm* c) { return c ? c->world : 0; }

extern "C" int rag_comm_all_gather_device(rag_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev || bytes_per_rank <= 0) return ragc_fail(RAG_ERR_INVALID_ARG, "bad arguments");
    RagcDeviceGuard g(c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    return ragc_comm_all_gather(c, send_dev, recv_dev, (size_t)bytes_per_rank, (hipStream_t)stream);
}

extern "C" int rag_comm_broadcast_device(rag_comm* c, void* buf_dev, int64_t bytes, int32_t root, void* stream) {
    if (!c || !c->comm || !buf_dev || bytes <= 0 || root < 0 || root >= c->world) return ragc_fail(RAG_ERR_INVALID_ARG, "bad arguments");
    RagcDeviceGuard g(c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    const int rc = g_rccl.Broadcast(buf_dev, buf_dev, (size_t)bytes, kNcclUint8, root, c->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) return nccl_fail("ncclBroadcast", rc);
    return RAG_OK;
}

extern "C" int rag_comm_request_device(rag_comm* c, const void* msg_host, void* msg_dev, int64_t bytes, int32_t root,
                                       void* head_mirror, uint64_t seq, void* stream) {
    if (!c || !c->comm || !msg_dev || bytes < 8 * RAG_COMM_HEAD_WORDS || bytes % 8 || root < 0 || root >= c->world)
        return ragc_fail(RAG_ERR_INVALID_ARG, "bad arguments (a request is at least %d bytes, a multiple of 8)", 8 * RAG_COMM_HEAD_WORDS);
    if ((reinterpret_cast<uintptr_t>(msg_dev) % 8) || (head_mirror && reinterpret_cast<uintptr_t>(head_mirror) % 8))
        return ragc_fail(RAG_ERR_INVALID_ARG, "request buffers must be 8-byte aligned");
    if (seq == 0 && head_mirror) return ragc_fail(RAG_ERR_INVALID_ARG, "sequence numbers start at 1 (0 is the mirror's idle value)");
    RagcDeviceGuard g(c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t st = (hipStream_t)stream;
    if (c->rank == root && msg_host) RAGC_HIP_TRY(hipMemcpyAsync(msg_dev, msg_host, (size_t)bytes, hipMemcpyHostToDevice, st));
    if (c->world > 1) {
        const int rc = g_rccl.Broadcast(msg_dev, msg_dev, (size_t)bytes, kNcclUint8, root, c->comm, st);
        if (rc != kNcclSuccess) return nccl_fail("ncclBroadcast", rc);
    }
    if (head_mirror) {
        post_head_kernel<<<dim3(1), dim3(64), 0, st>>>(static_cast<const unsigned long long*>(msg_dev),
                                                      static_cast<unsigned long long*>(head_mirror), RAG_COMM_HEAD_WORDS, seq);
        RAGC_HIP_TRY(hipGetLastError());
    }
    return RAG_OK;
}

extern "C" int rag_comm_wait_head(const void* head_mirror, uint64_t seq, int64_t timeout_us, int64_t* head_out) {
    if (!head_mirror || !head_out) return ragc_fail(RAG_ERR_INVALID_ARG, "null mirror or output");
    const volatile unsigned long long* m = static_cast<const volatile unsigned long long*>(head_mirror);
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long long spins = 0;; ++spins) {
        if (__atomic_load_n(&m[RAG_COMM_HEAD_WORDS], __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 1023) == 1023) {
            timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            const long long us = (t.tv_sec - t0.tv_sec) * 1000000LL + (t.tv_nsec - t0.tv_nsec) / 1000;
            if (timeout_us >= 0 && us > timeout_us) return ragc_fail(RAG_ERR_STATE, "no request %llu within %lld us", (unsigned long long)seq, (long long)timeout_us);
            if (us > 2000) {   // an idle service: stop burning the core, look again every 50 us
                timespec nap{0, 50000};
                nanosleep(&nap, nullptr);
            }
        } else {
            __builtin_ia32_pause();
        }
    }
    for (int i = 0; i < RAG_COMM_HEAD_WORDS; ++i) head_out[i] = (int64_t)m[i];
    return RAG_OK;
}
