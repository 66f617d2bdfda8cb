// rag_ivf_host.hip.h — host side of the IVFFlat `nprobe` mode (rag_ivf_* in include/rag_amd.h); part of rag_amd.hip's
// translation unit (it launches the flat search's merge kernel and uses its helpers).  No CPU compute path.
#pragma once
#include "ivf_kernels.hip.h"

struct rag_ivf {
    int device = 0, d = 0, d8 = 0, metric = 0, n_cus = 0;
    long long n = 0, nlist = 0;
    long long rows_padded = 0;        // every list rounded up to whole 32-row tiles
    rag_index* coarse = nullptr;      // flat index over the nlist centroids (the coarse quantizer)
    rag_index* rowsidx = nullptr;     // the rows in padded list order as a flat index: it owns X / xnorm and, for the two-stage
                                      // search, the scaled fp16 copy, its statistics and the certificate state
    float* X = nullptr;               // = rowsidx->X: d8 columns
    float* xnorm = nullptr;           // = rowsidx->xnorm (L2)
    bool two_stage = false;           // the fp16 copy exists and covers the corpus (rag_index_set_screening on rowsidx)
    uint32_t* fb_pin = nullptr;       // pinned: the two-stage search's any_fallback word, read back with the results (host entry points)
    uint32_t* ids = nullptr;          // kIvfPadId on padding rows
    uint32_t* tile_off = nullptr;     // nlist + 1: first tile of each list
    hipStream_t stream = nullptr;     // the host-pointer entry point's stream
    std::mutex mu;
    // the search workspace is shared by every stream searches are issued on: a search that follows one on a different
    // stream first waits for ws_event (as rag_index does)
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    // per-search workspace
    float* q_dev = nullptr; size_t q_cap = 0;
    float* qnorm = nullptr; size_t qn_cap = 0;
    float* c_scores = nullptr; long long* probe = nullptr; size_t probe_cap = 0;
    float* c_acc = nullptr;           // [centroid tiles * 32][kQT] inner products of the coarse step (own selection)
    ragk::IvfTile* tiles = nullptr; size_t tiles_cap = 0;   // a pass's packed (tile, mask) entries
    uint32_t* words = nullptr;        // [0] entries of the pass, [1] ticket, [2] done (the scan kernel leaves [1], [2] at 0), [4 ...] shared thresholds (a line per query)
    ragk::u64* partial = nullptr; size_t partial_cap = 0;
    ragk::u64* round_keys = nullptr;  // 2 x kQT keys (k > max_k rounds)
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
    if (d <= 0 || d > 1024) return fail(RAG_ERR_INVALID_ARG, "dimension %d out of range (the nprobe mode takes d <= 1024)", d);
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
    h->n_cus = coarse->n_cus;
    DeviceGuard g(device);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ws_event, hipEventDisableTiming) != hipSuccess) {
        if (h->stream) (void)hipStreamDestroy(h->stream);
        rag_index_destroy(coarse);
        delete h;
        return fail(RAG_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
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
        void* ptrs[] = {h->ids, h->tile_off, h->q_dev, h->qnorm, h->c_scores, h->probe, h->c_acc, h->tiles, h->words, h->partial,
                        h->round_keys, h->out_s, h->out_i};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        if (h->ws_event) (void)hipEventDestroy(h->ws_event);
        if (h->fb_pin) (void)hipHostFree(h->fb_pin);
    }
    rag_index_destroy(h->coarse);
    rag_index_destroy(h->rowsidx);   // X, xnorm, the fp16 copy
    delete h;
    return RAG_OK;
}

extern "C" int32_t rag_ivf_two_stage(const rag_ivf* h) { return h && h->two_stage ? 1 : 0; }
extern "C" int rag_ivf_screen_stats(rag_ivf* h, int64_t* queries, int64_t* fallbacks, double* max_err_ratio, int32_t reset) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    if (!h->rowsidx) {
        if (queries) *queries = 0;
        if (fallbacks) *fallbacks = 0;
        if (max_err_ratio) *max_err_ratio = 0.0;
        return RAG_OK;
    }
    return rag_index_screen_stats(h->rowsidx, queries, fallbacks, max_err_ratio, reset);
}
extern "C" int64_t rag_ivf_ntotal(const rag_ivf* h) { return h ? h->n : 0; }
extern "C" int64_t rag_ivf_nlist(const rag_ivf* h) { return h ? h->nlist : 0; }

extern "C" int rag_ivf_set_lists(rag_ivf* h, const float* centroids_host, int64_t nlist, const float* rows_host,
                                 const int64_t* ids_host, const int64_t* list_offsets_host) {
    using namespace ragk;
    if (!h || nlist <= 0 || !centroids_host || !list_offsets_host) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (h->n || h->nlist) return fail(RAG_ERR_STATE, "the lists of this index are already set");
    if (nlist > kIvfMaxLists) return fail(RAG_ERR_UNSUPPORTED, "the nprobe mode takes up to %d lists (nlist = %lld)", kIvfMaxLists, (long long)nlist);
    const long long n = list_offsets_host[nlist];
    if (list_offsets_host[0] != 0 || n < 0 || n > 0xFFFFFFFEll) return fail(RAG_ERR_INVALID_ARG, "list offsets must start at 0 and end below 2^32-1");
    for (int64_t l = 0; l < nlist; ++l)
        if (list_offsets_host[l + 1] < list_offsets_host[l]) return fail(RAG_ERR_INVALID_ARG, "list offsets must not decrease");
    if (n > 0 && (!rows_host || !ids_host)) return fail(RAG_ERR_INVALID_ARG, "null rows or ids");
    for (long long i = 0; i < n; ++i)
        if (ids_host[i] < 0 || ids_host[i] > 0xFFFFFFFEll) return fail(RAG_ERR_UNSUPPORTED, "stored id %lld outside [0, 2^32-2]", (long long)ids_host[i]);
    int rc = rag_index_add(h->coarse, centroids_host, nlist);
    if (rc) return rc;
    // padded layout: list l starts at tile tile_off[l]
    std::vector<uint32_t> toff((size_t)nlist + 1);
    long long tiles = 0;
    for (int64_t l = 0; l < nlist; ++l) {
        toff[(size_t)l] = (uint32_t)tiles;
        tiles += (list_offsets_host[l + 1] - list_offsets_host[l] + kTileRows - 1) / kTileRows;
    }
    toff[(size_t)nlist] = (uint32_t)tiles;
    const long long rows_padded = tiles * kTileRows;
    if (rows_padded > 0xFFFFFFFFll) return fail(RAG_ERR_UNSUPPORTED, "too many rows");
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = h->stream;
    if ((rc = dev_alloc(&h->tile_off, (size_t)nlist + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(h->tile_off, toff.data(), ((size_t)nlist + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if ((rc = dev_alloc(&h->words, (size_t)32 + 2 * ragk::kQT * ragk::kIvfThrStride))) return rc;
    HIP_TRY(hipMemsetAsync(h->words, 0, (32 + 2 * ragk::kQT * ragk::kIvfThrStride) * sizeof(uint32_t), st));
    if (rows_padded > 0) {
        if ((rc = dev_alloc(&h->ids, (size_t)rows_padded))) return rc;
        std::vector<uint32_t> ids32((size_t)rows_padded, kIvfPadId);
        for (int64_t l = 0; l < nlist; ++l) {
            const long long r0 = list_offsets_host[l], len = list_offsets_host[l + 1] - r0;
            const long long dst = (long long)toff[(size_t)l] * kTileRows;
            for (long long i = 0; i < len; ++i) ids32[(size_t)(dst + i)] = (uint32_t)ids_host[r0 + i];
        }
        HIP_TRY(hipMemcpyAsync(h->ids, ids32.data(), (size_t)rows_padded * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));   // ids32 leaves scope
    }
    // the rows, list after list with zero rows up to the next tile boundary, as a flat index of their own
    rag_index* R = nullptr;
    if ((rc = rag_index_create(h->d, h->metric, h->device, &R))) return rc;
    h->rowsidx = R;
    if (rows_padded > 0) {
        if ((rc = rag_index_reserve(R, rows_padded))) return rc;
        const std::vector<float> zeros((size_t)(kTileRows - 1) * h->d, 0.f);
        for (int64_t l = 0; l < nlist; ++l) {
            const long long r0 = list_offsets_host[l], len = list_offsets_host[l + 1] - r0;
            if (len == 0) continue;
            if ((rc = rag_index_add(R, rows_host + r0 * h->d, len))) return rc;
            const long long pad = (kTileRows - len % kTileRows) % kTileRows;
            if (pad && (rc = rag_index_add(R, zeros.data(), pad))) return rc;
        }
        if (R->n != rows_padded) return fail(RAG_ERR_STATE, "list layout: %lld rows placed, %lld expected", (long long)R->n, rows_padded);
        h->X = R->X;
        h->xnorm = R->xnorm;
        // the two-stage search (fp16 screening pass + exact second stage): on unless RAG_AMD_IVF_TWO_STAGE=0 or the
        // corpus is outside what it covers; never a reason not to serve
        if (env_int("RAG_AMD_IVF_TWO_STAGE", 1) != 0 && round_up(h->d, 64) <= kScreenMaxD64) {
            if (rag_index_set_screening(R, RAG_SCREEN_FP16) == RAG_OK) h->two_stage = R->screen_on && R->screen_valid;
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    h->n = n;
    h->nlist = nlist;
    h->rows_padded = rows_padded;
    return RAG_OK;
}

namespace {
using IvfScanFn = void (*)(const ragk::IvfBatchParams);
template <int E>
IvfScanFn ivf_scan_fn_ring(int ring) {
    switch (ring) {
        case 8: return (IvfScanFn)ragk::ivf_batch_scan_kernel<E, 8>;
        case 4: return (IvfScanFn)ragk::ivf_batch_scan_kernel<E, 4>;
        default: return (IvfScanFn)ragk::ivf_batch_scan_kernel<E, 1>;
    }
}
IvfScanFn ivf_scan_fn(int cap, int ring) {
    switch (cap) {
        case 64: return ivf_scan_fn_ring<1>(ring);
        case 128: return ivf_scan_fn_ring<2>(ring);
        default: return ivf_scan_fn_ring<4>(ring);
    }
}
// candidate buffer capacity for k (a buffer must hold k and the next arrival), 0 = k needs rounds
int ivf_capacity(int d8, int k) {
    for (int c : {64, 128, 256})
        if (k <= c - 16 && ragk::ivf_scan_lds_bytes(d8, c) <= 160 * 1024) return c;
    return 0;
}
int ivf_max_k(int d8) {
    int best = 0;
    for (int c : {64, 128, 256})
        if (ragk::ivf_scan_lds_bytes(d8, c) <= 160 * 1024) best = c - 16;
    return best;
}

// The exact list scan of one pass (<= 32 queries) over the plan's tiles, k in rounds, and the merge of the workgroups' lists.
// `enable` / `qmask` / `epoch`: as the two-stage search's fallback (no-op launches unless *enable == epoch, then only the
// queries whose qmask word equals epoch are rewritten); null / 0: the search itself.
int ivf_exact_pass(rag_ivf* h, const float* q_pass, const float* qnorm_pass, int nb, int k, float* out_s, long long* out_i,
                   hipStream_t st, const uint32_t* enable, const uint32_t* qmask, uint32_t epoch) {
    using namespace ragk;
    const int kmax = ivf_max_k(h->d8);
    const int grid = std::max(1, h->n_cus);
    const bool l2 = h->metric == RAG_METRIC_L2;
    const bool share_thr = env_int("RAG_AMD_IVF_SHARED_THRESHOLDS", 1) != 0;   // (experiment switch; results do not depend on it)
    const int S = h->d8 / 8;
    const int ring = S % 8 == 0 ? 8 : (S % 4 == 0 ? 4 : 1);
    int rc, done = 0, flip = 0;
    while (done < k) {
        const int kr = std::min(kmax, k - done);
        const int cap = ivf_capacity(h->d8, kr);
        const u64* ceil = done ? h->round_keys + (size_t)flip * kQT : nullptr;
        u64* last = k > kmax ? h->round_keys + (size_t)(flip ^ 1) * kQT : nullptr;
        IvfBatchParams sp{h->X, h->d8, h->xnorm, h->ids, q_pass, qnorm_pass, h->tiles, h->words, h->words + 1, h->words + 2, h->partial,
                          ceil, enable, epoch, share_thr ? h->words + 32 : nullptr, h->d, h->d8, nb, kr, l2 ? 1 : 0};
        if (done) HIP_TRY(hipMemsetAsync(h->words + 32, 0, kQT * kIvfThrStride * sizeof(uint32_t), st));   // a round's thresholds bind that round's candidates only
        IvfScanFn fn = ivf_scan_fn(cap, ring);
        const size_t lds = ivf_scan_lds_bytes(h->d8, cap);
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, st, sp);
        HIP_TRY(hipGetLastError());
        KeyListSrc src{h->partial, grid, kr};
        MergeOut mo{out_s + done, out_i + done, last, k, qnorm_pass, 0, h->metric, 0, qmask, epoch};
        launch_merge(src, grid, nb, kr, merge_look(grid, kr, kr), mo, st);
        HIP_TRY(hipGetLastError());
        done += kr;
        flip ^= 1;
    }
    return RAG_OK;
}

// the whole search on `st`, device pointers.  `deferred_epoch` (host entry points, one pass): a two-stage search does not
// enqueue its fallback launches; it hands back its epoch and the caller — who waits for the results anyway — reads
// any_fallback with them and runs ivf_exact_pass only when a certificate failed (as rag_index_search does)
int ivf_search_locked(rag_ivf* h, const float* q_dev, int nq, int k, int nprobe, float* out_s, long long* out_i, hipStream_t st,
                      uint32_t* deferred_epoch = nullptr) {
    if (deferred_epoch) *deferred_epoch = 0;
    using namespace ragk;
    if (h->ws_used && h->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, h->ws_event, 0));
    struct Mark {  // record the workspace hand-over point on every exit path
        rag_ivf* h;
        hipStream_t st;
        ~Mark() {
            if (hipEventRecord(h->ws_event, st) == hipSuccess) {
                h->ws_stream = st;
                h->ws_used = true;
            }
        }
    } mark{h, st};
    const int np = (int)std::min<long long>(nprobe, h->nlist);
    const int kmax = ivf_max_k(h->d8);
    const int grid = std::max(1, h->n_cus);
    if (grid > 256 * kMergeMaxOwned) return fail(RAG_ERR_UNSUPPORTED, "scan grid %d too large for the merge kernel", grid);
    int rc;
    if ((rc = ivf_grow(&h->qnorm, &h->qn_cap, (size_t)nq))) return rc;
    {
        size_t cap2 = h->probe_cap;
        if ((rc = ivf_grow(&h->c_scores, &cap2, (size_t)nq * np))) return rc;
        if ((rc = ivf_grow(&h->probe, &h->probe_cap, (size_t)nq * np))) return rc;
    }
    if ((rc = ivf_grow(&h->tiles, &h->tiles_cap, (size_t)(h->rows_padded / kTileRows + 1)))) return rc;
    if ((rc = ivf_grow(&h->partial, &h->partial_cap, (size_t)kQT * grid * std::max(std::min(k, kmax), 256)))) return rc;
    if (k > kmax && !h->round_keys && (rc = dev_alloc(&h->round_keys, (size_t)2 * kQT))) return rc;
    const bool l2 = h->metric == RAG_METRIC_L2;
    const bool cl2 = h->coarse->metric == RAG_METRIC_L2;
    const bool need_qn = l2 || cl2;
    if (need_qn && np > 256) {   // (up to 256 probes the selection kernel computes the norms itself)
        query_sqnorm_kernel<<<dim3(nq), dim3(64), (size_t)h->d8 * sizeof(float), st>>>(q_dev, h->d, h->qnorm);
        HIP_TRY(hipGetLastError());
    }
    // step 1: the coarse quantizer — the nprobe nearest centroids per query, ties by ascending list number.  Up to 256
    // probes: the flat scan kernel parks the inner products and ivf_coarse_select_kernel ranks them (ivf_kernels.hip.h);
    // more: the flat search itself
    if (np <= 256) {
        rag_index* c = h->coarse;
        const int n_tiles = (int)((c->n + kTileRows - 1) / kTileRows);
        if (!h->c_acc && (rc = dev_alloc(&h->c_acc, (size_t)n_tiles * kTileRows * kQT))) return rc;
        const int grid = std::max(1, std::min(c->n_cus, n_tiles));
        ScanFn fn = scan_fn(8, 64, pick_ring(c->d8 / 8), cl2);
        const size_t lds = scan_lds_bytes(c->d8, 64);
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
        for (int b0 = 0; b0 < nq; b0 += kQT) {
            const int nb = std::min(kQT, nq - b0);
            ScanParams sp{};
            sp.X = c->X;
            sp.xnorm = c->xnorm;
            sp.qnorm = h->qnorm + b0;
            sp.Q = q_dev + (size_t)b0 * h->d;
            sp.acc_io = h->c_acc;
            sp.dc8 = c->d8;
            sp.acc_out = 1;
            sp.n_rows = c->n;
            sp.row_stride = c->d8;
            sp.d = c->d;
            sp.d8 = c->d8;
            sp.nq = nb;
            sp.k = 1;
            sp.kout = 1;
            sp.n_tiles = n_tiles;
            sp.n_iters = (n_tiles + grid * 8 - 1) / (grid * 8);
            sp.n_full = n_tiles / (grid * 8);
            sp.tile_step = 1;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, st, sp);
            HIP_TRY(hipGetLastError());
            long long* pr = h->probe + (size_t)b0 * np;
            using SelFn = void (*)(const float*, const float*, const float*, const float*, int, float*, int, int, int, long long*);
            const bool wide = c->n >= 2048;   // 8 waves per query from 2048 lists up
            SelFn sel = np <= 64 ? (wide ? (SelFn)ivf_coarse_select_kernel<1, 8> : (SelFn)ivf_coarse_select_kernel<1, 4>)
                        : np <= 128 ? (wide ? (SelFn)ivf_coarse_select_kernel<2, 8> : (SelFn)ivf_coarse_select_kernel<2, 4>)
                                    : (wide ? (SelFn)ivf_coarse_select_kernel<4, 8> : (SelFn)ivf_coarse_select_kernel<4, 4>);
            hipLaunchKernelGGL(sel, dim3(nb), dim3(wide ? 512 : 256), need_qn ? (size_t)h->d8 * sizeof(float) : 0, st, h->c_acc, c->xnorm,
                               (const float*)nullptr, q_dev + (size_t)b0 * h->d, h->d, need_qn ? h->qnorm + b0 : (float*)nullptr, (int)c->n, np,
                               cl2 ? 1 : 0, pr);
            HIP_TRY(hipGetLastError());
        }
    } else {
        rc = rag_index_search_device(h->coarse, q_dev, nq, np, h->c_scores, reinterpret_cast<int64_t*>(h->probe), (void*)st);
        if (rc) return rc;
    }
    const bool share_thr = env_int("RAG_AMD_IVF_SHARED_THRESHOLDS", 1) != 0;   // (experiment switch; results do not depend on it)
    // two-stage when the fp16 copy is valid and k is inside what the screening pass and ONE exact fallback round cover
    const bool two_stage = h->two_stage && h->rows_padded > 0 && k <= kScreenMaxK && k <= kmax && h->d8 <= 1024 &&
                           screen_capacity(h->rowsidx->d64, k) > 0 &&
                           scan_lds_bytes_map(h->rowsidx->d64 / 2, screen_capacity(h->rowsidx->d64, k)) <= 160 * 1024 &&
                           env_int("RAG_AMD_IVF_TWO_STAGE", 1) != 0;
    for (int b0 = 0; b0 < nq; b0 += kQT) {
        const int nb = std::min(kQT, nq - b0);
        // step 2: this pass's probe table -> list masks -> work items
        IvfPlanParams pp{h->probe + (size_t)b0 * np, h->tile_off, h->tiles, h->words, h->words + 32, nb, np, (int)h->nlist};
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(&ivf_plan_kernel), (size_t)h->nlist * sizeof(uint32_t)))) return rc;
        ivf_plan_kernel<<<dim3(1), dim3(1024), (size_t)h->nlist * sizeof(uint32_t), st>>>(pp);
        HIP_TRY(hipGetLastError());
        // step 3: the probed lists, each read once, against the pass's queries; step 4: merge of the workgroups' lists.
        // Two-stage form (k <= 100, the fp16 copy valid): the flat search's screening pass over the tile map — half the bytes —
        // then its resolve kernel (canonical fp32 scores of the band, certificate, results ranked by stored id), then the
        // exact scan below as the fallback: a no-op unless a certificate failed, and then only those queries are rewritten
        const uint32_t* fb_enable = nullptr;
        const uint32_t* fb_qmask = nullptr;
        uint32_t fb_epoch = 0;
        if (two_stage) {
            rag_index* R = h->rowsidx;
            const int kp = 240;
            const int cap = screen_capacity(R->d64, k);
            const int kout = std::min(kp, cap);
            if (++R->screen_epoch == 0) R->screen_epoch = 1;
            const uint32_t epoch = R->screen_epoch;
            ScanParams sp{};
            sp.X = reinterpret_cast<const float*>(R->X16);
            sp.xnorm = R->xnorm;
            sp.qnorm = h->qnorm + b0;
            sp.Q = q_dev + (size_t)b0 * h->d;
            sp.partial = h->partial;
            sp.dc8 = R->d64 / 2;
            sp.n_rows = h->rows_padded;
            sp.row_stride = R->d64 / 2;
            sp.d = h->d;
            sp.d8 = h->d8;
            sp.nq = nb;
            sp.k = k;
            sp.n_tiles = (int)(h->rows_padded / kTileRows);
            sp.epoch = epoch;
            sp.tile_step = 1;
            sp.kout = kout;
            sp.x_absmax = R->x_absmax;
            sp.x_normmax = R->x_normmax;
            sp.x_scale = R->x_scale;
            sp.margin_out = R->sq->margin;
            sp.lossy = R->sq->lossy;
            sp.wg_lossy = R->wg_lossy;
            sp.map = h->tiles;
            sp.map_count = h->words;
            sp.map_ids = h->ids;
            sp.map_thr = share_thr ? h->words + 32 + kQT * kIvfThrStride : nullptr;
            sp.map_ticket = h->words + 1;   // (the exact scan's counters: the two kernels of a pass run one after the other)
            sp.map_done = h->words + 2;
            const int S16 = R->d64 / 16;
            ScanFn fn = screen_map_fn(cap, S16 % 8 == 0 ? 8 : 4, l2);
            const size_t lds = scan_lds_bytes_map(R->d64 / 2, cap);
            if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, st, sp);
            HIP_TRY(hipGetLastError());
            ResolveParams rp{};
            rp.src = KeyListSrc{h->partial, grid, kout};
            rp.n_lists = grid;
            rp.k = k;
            rp.kp = kp;
            rp.look = merge_look(grid, kout, k);
            rp.X = R->X;
            rp.row_stride = R->d8;
            rp.xnorm = R->xnorm;
            rp.Q = q_dev + (size_t)b0 * h->d;
            rp.d = h->d;
            rp.d8 = h->d8;
            rp.l2 = l2 ? 1 : 0;
            rp.qnorm = h->qnorm + b0;
            rp.id_offset = 0;
            rp.qs = R->sq;
            rp.wg_lossy = R->wg_lossy;
            rp.epoch = epoch;
            rp.ctr = R->sctr;
            rp.out_s = out_s + (size_t)b0 * k;
            rp.out_i = out_i + (size_t)b0 * k;
            rp.flag_out = nullptr;
            rp.id_map = h->ids;
            const size_t rlds = resolve_lds_bytes(h->d8, grid, rp.look, kp);
            using ResolveFn = void (*)(const ResolveParams);
            ResolveFn resolve = grid <= 256 ? (ResolveFn)screen_resolve_kernel<4, 1, 8>
                                            : (grid <= 512 ? (ResolveFn)screen_resolve_kernel<2, 4, 8> : (ResolveFn)screen_resolve_kernel<kMergeMaxOwned, 4, 8>);
            if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(resolve), rlds))) return rc;
            hipLaunchKernelGGL(resolve, dim3(nb), dim3(256), rlds, st, rp);
            HIP_TRY(hipGetLastError());
            fb_enable = &R->sq->any_fallback;
            fb_qmask = R->sq->fallback;
            fb_epoch = epoch;
        }
        if (two_stage && deferred_epoch && nq <= kQT) {
            *deferred_epoch = fb_epoch;
            continue;
        }
        if ((rc = ivf_exact_pass(h, q_dev + (size_t)b0 * h->d, h->qnorm + b0, nb, k, out_s + (size_t)b0 * k, out_i + (size_t)b0 * k, st,
                                 fb_enable, fb_qmask, fb_epoch)))
            return rc;
    }
    return RAG_OK;
}

int ivf_check_args(rag_ivf* h, const void* q, int nq, int k, int nprobe, const void* os, const void* oi) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    if (nq < 0 || k <= 0 || nprobe <= 0) return fail(RAG_ERR_INVALID_ARG, "nq=%d k=%d nprobe=%d out of range", nq, k, nprobe);
    if (nq > 0 && (!q || !os || !oi)) return fail(RAG_ERR_INVALID_ARG, "null query or output buffer");
    if (nq > 0 && h->nlist == 0) return fail(RAG_ERR_STATE, "rag_ivf_set_lists has not been called");
    if (nq > 65535) return fail(RAG_ERR_UNSUPPORTED, "more than 65535 queries per call");
    if (k > 2048) return fail(RAG_ERR_UNSUPPORTED, "k = %d: the nprobe mode returns up to 2048 results per query", k);
    return RAG_OK;
}
}  // namespace

extern "C" int rag_ivf_search_device(rag_ivf* h, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                                     float* out_scores_dev, int64_t* out_ids_dev, void* stream) {
    int rc = ivf_check_args(h, queries_dev, nq, k, nprobe, out_scores_dev, out_ids_dev);
    if (rc || nq == 0) return rc;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    return ivf_search_locked(h, queries_dev, nq, k, nprobe, out_scores_dev, reinterpret_cast<long long*>(out_ids_dev), (hipStream_t)stream);
}

namespace {
// results of a search on `st` to host memory: one read-back per array, one wait
int ivf_search_to_host(rag_ivf* h, const float* q_dev, int nq, int k, int nprobe, float* out_scores, int64_t* out_ids, hipStream_t st) {
    int rc;
    {
        size_t cap2 = h->out_cap;
        if ((rc = ivf_grow(&h->out_s, &cap2, (size_t)nq * k))) return rc;
        if ((rc = ivf_grow(&h->out_i, &h->out_cap, (size_t)nq * k))) return rc;
    }
    uint32_t epoch = 0;
    if ((rc = ivf_search_locked(h, q_dev, nq, k, nprobe, h->out_s, h->out_i, st, &epoch))) return rc;
    if (epoch && !h->fb_pin) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->fb_pin), sizeof(uint32_t), hipHostMallocDefault));
    if (epoch) HIP_TRY(hipMemcpyAsync(h->fb_pin, &h->rowsidx->sq->any_fallback, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_scores, h->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_ids, h->out_i, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (epoch && *h->fb_pin == epoch) {   // a certificate failed: the exact scan rewrites those queries (the plan's tiles are still in place)
        if ((rc = ivf_exact_pass(h, q_dev, h->qnorm, nq, k, h->out_s, h->out_i, st, nullptr, h->rowsidx->sq->fallback, epoch))) return rc;
        HIP_TRY(hipMemcpyAsync(out_scores, h->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_ids, h->out_i, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RAG_OK;
}
}  // namespace

extern "C" int rag_ivf_search(rag_ivf* h, const float* queries_host, int32_t nq, int32_t k, int32_t nprobe,
                              float* out_scores, int64_t* out_ids) {
    int rc = ivf_check_args(h, queries_host, nq, k, nprobe, out_scores, out_ids);
    if (rc || nq == 0) return rc;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = h->stream;
    if ((rc = ivf_grow(&h->q_dev, &h->q_cap, (size_t)nq * h->d))) return rc;
    HIP_TRY(hipMemcpyAsync(h->q_dev, queries_host, (size_t)nq * h->d * sizeof(float), hipMemcpyHostToDevice, st));
    return ivf_search_to_host(h, h->q_dev, nq, k, nprobe, out_scores, out_ids, st);
}

extern "C" int rag_ivf_search_device_host_out(rag_ivf* h, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                                              float* out_scores, int64_t* out_ids, void* stream) {
    int rc = ivf_check_args(h, queries_dev, nq, k, nprobe, out_scores, out_ids);
    if (rc || nq == 0) return rc;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    return ivf_search_to_host(h, queries_dev, nq, k, nprobe, out_scores, out_ids, (hipStream_t)stream);
}

extern "C" int rag_ivf_search_gather_device(rag_ivf* h, rag_comm* c, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                                            void* pack_dev, void* gathered_dev, float* out_scores_dev, int64_t* out_ids_dev,
                                            uint32_t* any_flag_dev, void* host_mirror, void* stream, void* comm_stream) {
    if (!c) return fail(RAG_ERR_INVALID_ARG, "null communicator");
    if (nq <= 0) return fail(RAG_ERR_INVALID_ARG, "nq must be positive");
    if (!pack_dev || !gathered_dev || !any_flag_dev) return fail(RAG_ERR_INVALID_ARG, "null pack / gather / flag buffer");
    if (reinterpret_cast<uintptr_t>(pack_dev) % 8 || reinterpret_cast<uintptr_t>(gathered_dev) % 8)
        return fail(RAG_ERR_INVALID_ARG, "packed blocks must be 8-byte aligned");
    int64_t s_off = 0, f_off = 0, blk = 0;
    int rc = rag_pack_layout(nq, k, &s_off, &f_off, &blk);
    if (rc) return rc;
    char* pack = static_cast<char*>(pack_dev);
    float* pack_s = reinterpret_cast<float*>(pack + s_off);
    long long* pack_i = reinterpret_cast<long long*>(pack);
    uint32_t* pack_f = reinterpret_cast<uint32_t*>(pack + f_off);
    rc = ivf_check_args(h, queries_dev, nq, k, nprobe, pack_s, pack_i);
    if (rc) return rc;
    if (ragc_comm_device(c) != h->device) return fail(RAG_ERR_INVALID_ARG, "communicator lives on device %d, the index on %d", ragc_comm_device(c), h->device);
    const int world = ragc_comm_world(c);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipStream_t cst = comm_stream ? (hipStream_t)comm_stream : st;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        HIP_TRY(hipMemsetAsync(pack_f, 0, sizeof(uint32_t), st));   // this rank's list is final
        rc = ivf_search_locked(h, queries_dev, nq, k, nprobe, pack_s, pack_i, st);   // (records h->ws_event on st when it returns)
        if (rc) return rc;
        if (cst != st) HIP_TRY(hipStreamWaitEvent(cst, h->ws_event, 0));
        rc = ragc_comm_all_gather(c, pack_dev, gathered_dev, (size_t)blk, cst);
        if (rc) return rc;
    }
    return rag_merge_topk_packed_flagged_device(h->device, h->metric, world, nq, k, gathered_dev, blk, s_off, f_off, out_scores_dev,
                                                out_ids_dev, any_flag_dev, host_mirror, (void*)cst);
}
