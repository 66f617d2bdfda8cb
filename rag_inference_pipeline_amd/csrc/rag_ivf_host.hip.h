// rag_ivf_host.hip.h — host side of the IVFFlat `nprobe` mode (rag_ivf_* in include/rag_amd.h); part of rag_amd.hip's
// translation unit (it launches the flat search's merge kernel and uses its helpers).  No CPU compute path.
#pragma once
#include "ivf_kernels.hip.h"

struct rag_ivf {
    int device = 0, d = 0, d8 = 0, metric = 0;
    long long n = 0, nlist = 0;
    rag_index* coarse = nullptr;      // flat index over the nlist centroids (the coarse quantizer)
    float* X = nullptr;               // rows in list order, d8 columns
    float* xnorm = nullptr;           // L2
    uint32_t* ids = nullptr;
    long long* list_off = nullptr;    // nlist + 1
    hipStream_t stream = nullptr;
    std::mutex mu;
    // per-search workspace
    float* q_dev = nullptr; size_t q_cap = 0;
    float* qnorm = nullptr; size_t qn_cap = 0;
    float* c_scores = nullptr; long long* probe = nullptr; size_t probe_cap = 0;
    ragk::u64* partial = nullptr; size_t partial_cap = 0;
    float* out_s = nullptr; long long* out_i = nullptr; size_t out_cap = 0;
};

namespace {
template <typename T>
int ivf_grow(T** p, size_t* cap, size_t want) {
    if (want <= *cap) return RAG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    int rc = dev_alloc(p, want);
    if (rc) return rc;
    *cap = want;
    return RAG_OK;
}
}  // namespace

extern "C" int rag_ivf_create(int32_t d, int32_t metric, int32_t quantizer_metric, int32_t device, rag_ivf** out) {
    if (!out) return fail(RAG_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (d <= 0 || d > 4096) return fail(RAG_ERR_INVALID_ARG, "dimension %d out of range (IVF mode takes d <= 4096)", d);
    if ((metric != RAG_METRIC_INNER_PRODUCT && metric != RAG_METRIC_L2) ||
        (quantizer_metric != RAG_METRIC_INNER_PRODUCT && quantizer_metric != RAG_METRIC_L2))
        return fail(RAG_ERR_INVALID_ARG, "unknown metric");
    rag_index* coarse = nullptr;
    int rc = rag_index_create(d, quantizer_metric, device, &coarse);
    if (rc) return rc;
    rag_ivf* h = new (std::nothrow) rag_ivf();
    if (!h) {
        rag_index_destroy(coarse);
        return fail(RAG_ERR_OOM, "host allocation failed");
    }
    h->device = device;
    h->d = d;
    h->d8 = round_up(d, 8);
    h->metric = metric;
    h->coarse = coarse;
    DeviceGuard g(device);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        rag_index_destroy(coarse);
        delete h;
        return fail(RAG_ERR_HIP, "hipStreamCreate failed");
    }
    *out = h;
    return RAG_OK;
}

extern "C" int rag_ivf_destroy(rag_ivf* h) {
    if (!h) return RAG_OK;
    {
        DeviceGuard g(h->device);
        std::lock_guard<std::mutex> lk(h->mu);
        (void)hipDeviceSynchronize();
        void* ptrs[] = {h->X, h->xnorm, h->ids, h->list_off, h->q_dev, h->qnorm, h->c_scores, h->probe, h->partial, h->out_s, h->out_i};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    rag_index_destroy(h->coarse);
    delete h;
    return RAG_OK;
}

extern "C" int64_t rag_ivf_ntotal(const rag_ivf* h) { return h ? h->n : 0; }
extern "C" int64_t rag_ivf_nlist(const rag_ivf* h) { return h ? h->nlist : 0; }

extern "C" int rag_ivf_set_lists(rag_ivf* h, const float* centroids_host, int64_t nlist, const float* rows_host,
                                 const int64_t* ids_host, const int64_t* list_offsets_host) {
    if (!h || nlist <= 0 || !centroids_host || !list_offsets_host) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (h->n || h->nlist) return fail(RAG_ERR_STATE, "the lists of this index are already set");
    const long long n = list_offsets_host[nlist];
    if (list_offsets_host[0] != 0 || n < 0 || n > 0xFFFFFFFEll) return fail(RAG_ERR_INVALID_ARG, "list offsets must start at 0 and end below 2^32-1");
    for (int64_t l = 0; l < nlist; ++l)
        if (list_offsets_host[l + 1] < list_offsets_host[l]) return fail(RAG_ERR_INVALID_ARG, "list offsets must not decrease");
    if (n > 0 && (!rows_host || !ids_host)) return fail(RAG_ERR_INVALID_ARG, "null rows or ids");
    for (long long i = 0; i < n; ++i)
        if (ids_host[i] < 0 || ids_host[i] > 0xFFFFFFFEll) return fail(RAG_ERR_UNSUPPORTED, "stored id %lld outside [0, 2^32-2]", (long long)ids_host[i]);
    int rc = rag_index_add(h->coarse, centroids_host, nlist);
    if (rc) return rc;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    if ((rc = dev_alloc(&h->list_off, (size_t)nlist + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(h->list_off, list_offsets_host, ((size_t)nlist + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    if (n > 0) {
        if ((rc = dev_alloc(&h->X, (size_t)n * h->d8))) return rc;
        if ((rc = dev_alloc(&h->ids, (size_t)n))) return rc;
        if (h->d8 != h->d) HIP_TRY(hipMemsetAsync(h->X, 0, (size_t)n * h->d8 * sizeof(float), h->stream));
        HIP_TRY(hipMemcpy2DAsync(h->X, (size_t)h->d8 * sizeof(float), rows_host, (size_t)h->d * sizeof(float),
                                 (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyHostToDevice, h->stream));
        std::vector<uint32_t> ids32((size_t)n);
        for (long long i = 0; i < n; ++i) ids32[(size_t)i] = (uint32_t)ids_host[i];
        HIP_TRY(hipMemcpyAsync(h->ids, ids32.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));   // ids32 leaves scope
        if (h->metric == RAG_METRIC_L2) {
            if ((rc = dev_alloc(&h->xnorm, (size_t)n))) return rc;
            ragk::row_sqnorm_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream>>>(h->X, h->d8, h->d, 0, n, h->xnorm);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->n = n;
    h->nlist = nlist;
    return RAG_OK;
}

extern "C" int rag_ivf_search(rag_ivf* h, const float* queries_host, int32_t nq, int32_t k, int32_t nprobe,
                              float* out_scores, int64_t* out_ids) {
    using namespace ragk;
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    if (nq < 0 || k <= 0 || nprobe <= 0) return fail(RAG_ERR_INVALID_ARG, "nq=%d k=%d nprobe=%d out of range", nq, k, nprobe);
    if (nq > 0 && (!queries_host || !out_scores || !out_ids)) return fail(RAG_ERR_INVALID_ARG, "null query or output buffer");
    if (nq == 0) return RAG_OK;
    if (h->nlist == 0) return fail(RAG_ERR_STATE, "rag_ivf_set_lists has not been called");
    if (k > kIvfMaxK) return fail(RAG_ERR_UNSUPPORTED, "the IVF mode returns up to %d results per query (k = %d)", kIvfMaxK, k);
    const int np = (int)std::min<long long>(nprobe, h->nlist);
    if (np > 256 * kMergeMaxOwned) return fail(RAG_ERR_UNSUPPORTED, "nprobe %d exceeds the merge kernel's %d lists", np, 256 * kMergeMaxOwned);
    if (nq > 65535) return fail(RAG_ERR_UNSUPPORTED, "more than 65535 queries per call");
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = h->stream;
    int rc;
    if ((rc = ivf_grow(&h->q_dev, &h->q_cap, (size_t)nq * h->d))) return rc;
    if ((rc = ivf_grow(&h->qnorm, &h->qn_cap, (size_t)nq))) return rc;
    {
        size_t cap2 = h->probe_cap;
        if ((rc = ivf_grow(&h->c_scores, &cap2, (size_t)nq * np))) return rc;
        if ((rc = ivf_grow(&h->probe, &h->probe_cap, (size_t)nq * np))) return rc;
    }
    if ((rc = ivf_grow(&h->partial, &h->partial_cap, (size_t)nq * np * k))) return rc;
    {
        size_t cap2 = h->out_cap;
        if ((rc = ivf_grow(&h->out_s, &cap2, (size_t)nq * k))) return rc;
        if ((rc = ivf_grow(&h->out_i, &h->out_cap, (size_t)nq * k))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(h->q_dev, queries_host, (size_t)nq * h->d * sizeof(float), hipMemcpyHostToDevice, st));
    // step 1: the coarse quantizer — the flat search over the centroids, nprobe nearest per query
    rc = rag_index_search_device(h->coarse, h->q_dev, nq, np, h->c_scores, reinterpret_cast<int64_t*>(h->probe), (void*)st);
    if (rc) return rc;
    if (h->metric == RAG_METRIC_L2) {
        row_sqnorm_kernel<<<dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st>>>(h->q_dev, h->d, h->d, 0, nq, h->qnorm);
        HIP_TRY(hipGetLastError());
    }
    // step 2: every (query, probed list) pair scans its list; step 3: the merge of a query's nprobe lists
    IvfScanParams sp{h->X, h->d8, h->xnorm, h->ids, h->list_off, h->q_dev, h->qnorm, h->probe, h->partial,
                     h->d, h->d8, np, k, h->metric == RAG_METRIC_L2 ? 1 : 0};
    const size_t lds = 512 * 8 + (size_t)h->d8 * sizeof(float);
    ivf_list_scan_kernel<<<dim3((unsigned)np, (unsigned)nq), dim3(256), lds, st>>>(sp);
    HIP_TRY(hipGetLastError());
    KeyListSrc src{h->partial, np, k};
    MergeOut mo{h->out_s, h->out_i, nullptr, k, h->qnorm, 0, h->metric, 0, nullptr, 0};
    launch_merge(src, np, nq, k, merge_look(np, k, k), mo, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_scores, h->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_ids, h->out_i, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RAG_OK;
}
