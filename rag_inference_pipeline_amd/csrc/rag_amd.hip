// rag_amd.hip — C ABI (include/rag_amd.h) over the gfx950 kernels in flat_kernels.hip.h.
// Host side of the flat index: device memory, streams, launch geometry, status codes.
// There is no CPU compute path in this file: without a HIP device every entry point fails.
#include "../../include/rag_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "flat_kernels.hip.h"
#include "rag_comm_internal.h"
#include "rag_common.h"

namespace {

thread_local char g_err[512] = "";

#define fail ragc_fail

#define HIP_TRY RAGC_HIP_TRY
using DeviceGuard = RagcDeviceGuard;

template <typename T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) return RAG_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return RAG_OK;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Candidate-buffer capacity (64 * E) for a (d8, k) pair; 0 when nothing fits the 160 KiB LDS.
int pick_capacity(int d8, int k) {
    constexpr size_t kLdsMax = 160 * 1024;
    const int want = k <= 32 ? 64 : (k <= 96 ? 128 : 256);
    for (int c = want; c <= 256; c <<= 1)
        if (c >= k + 16 && ragk::scan_lds_bytes(d8, c) <= kLdsMax) return c;
    for (int c = want >> 1; c >= 64; c >>= 1)
        if (c >= k + 16 && ragk::scan_lds_bytes(d8, c) <= kLdsMax) return c;
    return 0;
}

// Column chunk held in LDS per launch: the whole row up to 1024 columns, else 768 (so k up to 112 fits
// beside the query image); the last chunk takes the remainder.
constexpr int kChunkCols = 768;
int chunk_cols(int d8) { return d8 <= 1024 ? d8 : kChunkCols; }

int pick_ring(int S) {
    for (int dpt : {8, 4, 2})
        if (S % dpt == 0 && S >= dpt) return dpt;
    return 1;
}
bool ring_ok(int S, int ring) {
    return (ring == 1 || ring == 2 || ring == 4 || ring == 8 || ring == 12 || ring == 16) && S % ring == 0 && S >= ring;
}

using ScanFn = void (*)(const ragk::ScanParams);

template <int NW, int E, int D>
ScanFn scan_fn_metric(bool l2) {
    return l2 ? (ScanFn)ragk::scan_topk_kernel<NW, E, D, true, 0> : (ScanFn)ragk::scan_topk_kernel<NW, E, D, false, 0>;
}

// screening pass (fp16 copy): 8 waves, ring 8 or 4
template <int E>
ScanFn screen_fn_cap(int ring, bool l2) {
    if (ring == 8)
        return l2 ? (ScanFn)ragk::scan_topk_kernel<8, E, 8, true, 1> : (ScanFn)ragk::scan_topk_kernel<8, E, 8, false, 1>;
    return l2 ? (ScanFn)ragk::scan_topk_kernel<8, E, 4, true, 1> : (ScanFn)ragk::scan_topk_kernel<8, E, 4, false, 1>;
}
// the same pass over a tile map (MAP = true: the IVFFlat nprobe mode's screening pass, rag_ivf_host.hip.h)
template <int E>
ScanFn screen_map_fn_cap(int ring, bool l2) {
    if (ring == 8)
        return l2 ? (ScanFn)ragk::scan_topk_kernel<8, E, 8, true, 1, true> : (ScanFn)ragk::scan_topk_kernel<8, E, 8, false, 1, true>;
    return l2 ? (ScanFn)ragk::scan_topk_kernel<8, E, 4, true, 1, true> : (ScanFn)ragk::scan_topk_kernel<8, E, 4, false, 1, true>;
}
ScanFn screen_map_fn(int cap, int ring, bool l2) {
    switch (cap) {
        case 64: return screen_map_fn_cap<1>(ring, l2);
        case 128: return screen_map_fn_cap<2>(ring, l2);
        default: return screen_map_fn_cap<4>(ring, l2);
    }
}
ScanFn screen_fn(int cap, int ring, bool l2) {
    switch (cap) {
        case 64: return screen_fn_cap<1>(ring, l2);
        case 128: return screen_fn_cap<2>(ring, l2);
        default: return screen_fn_cap<4>(ring, l2);
    }
}
template <int NW, int E>
ScanFn scan_fn_ring(int ring, bool l2) {
    switch (ring) {
#ifdef RAGK_TUNING
        case 16: return scan_fn_metric<NW, E, 16>(l2);
        case 12: return scan_fn_metric<NW, E, 12>(l2);
#endif
        case 8: return scan_fn_metric<NW, E, 8>(l2);
        case 4: return scan_fn_metric<NW, E, 4>(l2);
        case 2: return scan_fn_metric<NW, E, 2>(l2);
        default: return scan_fn_metric<NW, E, 1>(l2);
    }
}
template <int NW>
ScanFn scan_fn_cap(int cap, int ring, bool l2) {
    switch (cap) {
        case 64: return scan_fn_ring<NW, 1>(ring, l2);
        case 128: return scan_fn_ring<NW, 2>(ring, l2);
        default: return scan_fn_ring<NW, 4>(ring, l2);
    }
}
ScanFn scan_fn(int waves, int cap, int ring, bool l2) {
    switch (waves) {
#ifdef RAGK_TUNING  // measured on 10M x 768: 8 waves x ring 8 is as fast as any of these
        case 16: return scan_fn_cap<16>(cap, ring, l2);
        case 12: return scan_fn_cap<12>(cap, ring, l2);
#endif
        default: return scan_fn_cap<8>(cap, ring, l2);
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of on every
// launch: the call takes the runtime's global lock and showed up between the launches of a search.
int ensure_dyn_lds(const void* fn, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    RAGC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = granted[{dev, fn}];
    if (lds <= have) return RAG_OK;
    RAGC_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
    return RAG_OK;
}

#ifdef RAGK_STAMPS
unsigned long long* g_stamps = nullptr;  // [2048][8], experiment build only
#endif

constexpr int kWgLossyLists = 2048;  // = 256 * kMergeMaxOwned, the largest scan grid the merge takes

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

}  // namespace

int ragc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct rag_index {
    int device = 0;
    int d = 0, d8 = 0, metric = 0;
    int n_cus = 0;
    long long n = 0, cap_rows = 0;
    long long id_offset = 0;
    float* X = nullptr;        // cap_rows x d8
    float* xnorm = nullptr;    // cap_rows (L2 only; padded to a multiple of 4 rows)
    hipStream_t stream = nullptr;
    std::mutex mu;

    // per-search workspace (grown on demand, device memory)
    float* q_dev = nullptr;  size_t q_dev_cap = 0;          // queries staged from the host
    float* qnorm = nullptr;                                  // kQT floats
    ragk::u64* round_keys = nullptr;                         // 2 x kQT keys (k > max_k rounds)
    float* acc_io = nullptr; long long acc_rows = 0;         // running sums when d > 1024 is scanned in chunks
    ragk::u64* partial = nullptr; size_t partial_cap = 0;   // kQT * grid * k keys
    float* out_s_dev = nullptr; long long* out_i_dev = nullptr; size_t out_cap = 0;
    // pinned host staging
    float* q_pin = nullptr; size_t q_pin_cap = 0;
    float* out_s_pin = nullptr; long long* out_i_pin = nullptr; size_t out_pin_cap = 0;
    uint32_t* fb_pin = nullptr;  // the two-stage search's any_fallback word, read back with the results

    // the search workspace is shared by every stream searches are issued on: a search that
    // follows one on a different stream first waits for ws_event
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;

    // two-stage search (rag_index_set_screening): scaled fp16 copy of the corpus + certificate state
    bool screen_on = false;      // requested
    bool screen_valid = false;   // the copy exists and the corpus is inside the range the bound covers
    _Float16* X16 = nullptr;     // cap_rows x d64
    int d64 = 0;
    float x_absmax = 0.f, x_normmax = 0.f, x_scale = 0.f;  // x_scale == 0: not chosen yet
    ragk::ScreenCorpusStats* sc_stats = nullptr;
    ragk::ScreenQueryState* sq = nullptr;
    ragk::ScreenCounters* sctr = nullptr;
    uint32_t screen_epoch = 0;   // id of the last two-stage search (ragk::ScreenQueryState flags are epoch-valued)
    uint32_t* wg_lossy = nullptr;  // kQT x kWgLossyLists epoch-valued words: (query, workgroup) pairs that gave up a band

    // dynamic deal of a long scan's last rounds: one ticket counter per launch out of a ring — launch i zeroes the
    // counter of launch i + 1 (every scan launch of an index is stream-ordered behind the previous one)
    uint32_t* dyn_ctrs = nullptr;
    unsigned dyn_slot = 0;
    int dyn_rounds = 2;          // rounds' worth of tiles dealt by ticket (0: static deal only); RAG_AMD_SCAN_DYN_ROUNDS
    int dyn_singles_per_wg = 0;  // single-tile tickets per workgroup at the very end; RAG_AMD_SCAN_DYN_SINGLES

    // sample pass (starting thresholds for k >= kSampleMinK)
    ragk::u64* sample_heads = nullptr;  // kQT x kSampleLists workgroup maxima
    ragk::u64* thr_keys = nullptr;      // kQT

    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    double prof_ms = 0.0;
    long long prof_launches = 0;
};

namespace {

int grow_rows(rag_index* h, long long want) {
    if (want <= h->cap_rows) return RAG_OK;
    long long cap = std::max<long long>(want, h->cap_rows + h->cap_rows / 2);
    cap = (cap + 31) / 32 * 32;
    float* nx = nullptr;
    int rc = dev_alloc(&nx, (size_t)cap * h->d8);
    if (rc) return rc;
    float* nn = nullptr;
    if (h->metric == RAG_METRIC_L2) {
        rc = dev_alloc(&nn, (size_t)cap);
        if (rc) {
            (void)hipFree(nx);
            return rc;
        }
    }
    if (h->n > 0) {
        HIP_TRY(hipMemcpyAsync(nx, h->X, (size_t)h->n * h->d8 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        if (nn) HIP_TRY(hipMemcpyAsync(nn, h->xnorm, (size_t)h->n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->screen_on) {
        _Float16* n16 = nullptr;
        rc = dev_alloc(&n16, (size_t)cap * h->d64);
        if (rc) {
            (void)hipFree(nx);
            if (nn) (void)hipFree(nn);
            return rc;
        }
        if (h->n > 0 && h->X16) {
            HIP_TRY(hipMemcpyAsync(n16, h->X16, (size_t)h->n * h->d64 * sizeof(_Float16), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        if (h->X16) (void)hipFree(h->X16);
        h->X16 = n16;
    }
    if (h->X) (void)hipFree(h->X);
    if (h->xnorm) (void)hipFree(h->xnorm);
    h->X = nx;
    h->xnorm = nn;
    h->cap_rows = cap;
    if (h->acc_io) {  // sized by capacity: reallocated by the next chunked search
        (void)hipFree(h->acc_io);
        h->acc_io = nullptr;
    }
    return RAG_OK;
}

// Bring the fp16 screening copy up to date for rows [row0, row0 + n) that have just landed in X.
// Synchronises `st` (the maxima decide the scale, and with it whether everything is converted again).
int screen_sync_rows(rag_index* h, long long row0, long long n, hipStream_t st) {
    using namespace ragk;
    if (!h->screen_on || n <= 0) return RAG_OK;
    const unsigned grid = (unsigned)std::min<long long>((n + 3) / 4, 4096);
    screen_stats_kernel<<<dim3(grid), dim3(256), 0, st>>>(h->X, h->d8, h->d8, row0, n, h->sc_stats);
    HIP_TRY(hipGetLastError());
    ScreenCorpusStats cs;
    HIP_TRY(hipMemcpyAsync(&cs, h->sc_stats, sizeof cs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float absmax, sqn;
    std::memcpy(&absmax, &cs.absmax_bits, 4);
    std::memcpy(&sqn, &cs.sqnorm_bits, 4);
    const bool was_valid = h->screen_valid;
    h->screen_valid = false;
    if (cs.bad || !(sqn <= 1.0e24f) || (absmax != 0.f && !(absmax >= 1.0e-12f && absmax <= 1.0e12f)))
        return RAG_OK;  // outside the range the error bound covers: searches use the fp32 scan
    h->x_absmax = absmax;
    h->x_normmax = std::sqrt(sqn) * 1.0001f;
    long long c0 = row0, cn = n;
    if (!was_valid || h->x_scale == 0.f || !(absmax * h->x_scale < 32768.f)) {  // (re)choose the scale: convert every row
        int e = 0;
        if (absmax > 0.f) (void)std::frexp(absmax, &e);
        h->x_scale = std::ldexp(1.f, 14 - e);
        c0 = 0;
        cn = row0 + n;
    }
    const unsigned cgrid = (unsigned)std::min<long long>((cn + 3) / 4, 4096);
    screen_convert_kernel<<<dim3(cgrid), dim3(256), 0, st>>>(h->X, h->d8, h->d8, c0, cn, h->x_scale, h->X16, h->d64);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    h->screen_valid = true;
    return RAG_OK;
}

int after_add(rag_index* h, long long row0, long long n, hipStream_t st) {
    if (h->metric == RAG_METRIC_L2 && n > 0) {
        const int bs = 256;
        ragk::row_sqnorm_kernel<<<dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st>>>(h->X, h->d8, h->d, row0, n,
                                                                                         h->xnorm);
        HIP_TRY(hipGetLastError());
    }
    return screen_sync_rows(h, row0, n, st);
}

template <class Src>
void launch_merge(const Src& src, int n_lists, int nq, int k, int look, const ragk::MergeOut& mo, hipStream_t st) {
    using namespace ragk;
    const size_t lds = (size_t)n_lists * look * 8;
    if (n_lists <= 256)  // one wave, four lists per lane: no barrier inside a round
        tournament_merge_kernel<Src, 4, 1><<<dim3(nq), dim3(64), lds, st>>>(src, n_lists, k, look, mo);
    else if (n_lists <= 512)
        tournament_merge_kernel<Src, 2, 4><<<dim3(nq), dim3(256), lds, st>>>(src, n_lists, k, look, mo);
    else
        tournament_merge_kernel<Src, kMergeMaxOwned, 4><<<dim3(nq), dim3(256), lds, st>>>(src, n_lists, k, look, mo);
}

void prof_push(rag_index* h, hipEvent_t e0, hipEvent_t e1) {
    h->prof_events.emplace_back(e0, e1);
    if (h->prof_events.size() >= 4096) {  // nobody is collecting: fold what has finished into the totals
        for (auto& ev : h->prof_events) {
            float ms = 0.f;
            if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                h->prof_ms += ms;
                h->prof_launches += 1;
            }
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        h->prof_events.clear();
    }
}

int ensure_search_ws(rag_index* h, int nq_total, int k, int grid) {
    const size_t part = (size_t)ragk::kQT * grid * k;
    if (part > h->partial_cap) {
        if (h->partial) (void)hipFree(h->partial);
        h->partial = nullptr;
        h->partial_cap = 0;
        int rc = dev_alloc(&h->partial, part);
        if (rc) return rc;
        h->partial_cap = part;
    }
    if (!h->qnorm) {
        int rc = dev_alloc(&h->qnorm, (size_t)ragk::kQT);
        if (rc) return rc;
    }
    (void)nq_total;
    return RAG_OK;
}

// Dynamic deal of the last rounds of a long scan (flat_kernels.hip.h "tile walk"): fills the dyn_* fields of `sp`
// and shortens its static part to n_full - dyn_rounds rounds.  Short scans (fewer than kDynMinRounds full rounds)
// and chunked scans keep the static deal.
constexpr int kDynMinRounds = 6;
constexpr int kDynSlots = 64;
int setup_dynamic_deal(rag_index* h, ragk::ScanParams& sp, int grid, int waves) {
    sp.dyn_ctr = sp.dyn_ctr_next = nullptr;
    sp.dyn_tile0 = sp.n_dyn_groups = sp.n_singles = 0;
    if (h->dyn_rounds <= 0 || sp.n_full < kDynMinRounds || waves != 8) return RAG_OK;
    if (!h->dyn_ctrs) {
        int rc = dev_alloc(&h->dyn_ctrs, (size_t)kDynSlots);
        if (rc) return rc;
        HIP_TRY(hipMemset(h->dyn_ctrs, 0, kDynSlots * sizeof(uint32_t)));
    }
    const int n_static = std::max(4, sp.n_full - h->dyn_rounds);  // the warm start and the first sorts stay in lockstep
    const int tile0 = n_static * grid * waves;
    const int n_dyn_tiles = sp.n_tiles - tile0;
    const int want_singles = std::min(n_dyn_tiles, grid * std::max(0, h->dyn_singles_per_wg));
    const int n_groups = (n_dyn_tiles - want_singles) / waves;
    // (the slot is consumed by the caller, once the launch that zeroes the next counter has been enqueued: an error
    // return between here and there must not leave the next launch on a counter nobody reset)
    sp.dyn_ctr = h->dyn_ctrs + h->dyn_slot % kDynSlots;
    sp.dyn_ctr_next = h->dyn_ctrs + (h->dyn_slot + 1) % kDynSlots;
    sp.dyn_tile0 = tile0;
    sp.n_dyn_groups = n_groups;
    sp.n_singles = n_dyn_tiles - n_groups * waves;
    sp.n_full = n_static;
    sp.n_iters = n_static;
    return RAG_OK;
}

// Sample pass (flat_kernels.hip.h, "sample pass -> starting thresholds"): the scan kernel with k = 1 over
// kSampleLists * 8 tiles spread through the corpus, then the k-th largest workgroup maximum per query
// into h->thr_keys.  `base` is the ScanParams of the real scan (same corpus view, metric, ceiling).
constexpr int kSampleMinK = 32;        // below this the pass costs about what it saves
constexpr int kSampleLists = 256;      // workgroups of the sample pass (<= 256: one sort of 4 keys per lane)
constexpr long long kSampleMinTiles = 8LL * kSampleLists * 8;   // only when the corpus dwarfs the sample (>= 524k rows)

int run_sample_pass(rag_index* h, const ragk::ScanParams& base, ScanFn fn, size_t lds, int nq, int k, hipStream_t st) {
    using namespace ragk;
    int rc;
    if (!h->sample_heads && (rc = dev_alloc(&h->sample_heads, (size_t)kQT * kSampleLists))) return rc;
    if (!h->thr_keys && (rc = dev_alloc(&h->thr_keys, (size_t)kQT))) return rc;
    const int lists = std::min(h->n_cus, kSampleLists);
    ScanParams ss = base;
    ss.partial = h->sample_heads;
    ss.k = 1;
    ss.kout = 1;
    ss.n_tiles = lists * 8;
    ss.n_iters = 1;
    ss.n_full = 1;
    ss.tile_step = (int)(base.n_tiles / ss.n_tiles);
    ss.thr_key = nullptr;
    ss.enable = nullptr;
#ifdef RAGK_STAMPS
    ss.stamps = nullptr;
#endif
    if (ss.lossy) ss.lossy = h->sq->sample_lossy;
    ss.wg_lossy = nullptr;  // the sample pass's lists only seed thresholds
    ss.flag_clear = nullptr;
    ss.dyn_ctr = ss.dyn_ctr_next = nullptr;
    if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
    hipLaunchKernelGGL(fn, dim3(lists), dim3(8 * 64), lds, st, ss);
    HIP_TRY(hipGetLastError());
    sample_threshold_kernel<<<dim3(nq), dim3(64), 0, st>>>(h->sample_heads, lists, k, h->thr_keys);
    HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// One pass of <= 32 queries: scan + merge into out (device pointers, row stride k).
// `ceil` (device, kQT keys, or null) restricts candidates to keys below it; `last_key` (device, kQT keys, or
// null) receives each query's k-th key; results go to columns [0, k) of rows of `out_stride` elements.
// `enable` / `qmask` (device words, or null) make this the fallback of the two-stage search: with
// *enable == 0 every launch returns at once, otherwise only queries with qmask[q] != 0 are written.
int search_round(rag_index* h, const float* q_dev, int nq, int k, const ragk::u64* ceil, ragk::u64* last_key,
                 float* out_s, long long* out_i, int out_stride, hipStream_t st, const uint32_t* enable = nullptr,
                 const uint32_t* qmask = nullptr, uint32_t epoch = 0) {
    using namespace ragk;
    const int dc_full = chunk_cols(h->d8);
    const int cap = pick_capacity(dc_full, k);
    if (cap == 0)
        return fail(RAG_ERR_UNSUPPORTED, "k=%d with d=%d exceeds the fused scan kernel's LDS budget (max k %d)", k, h->d,
                    rag_index_max_k(h->d, nq));
    int waves = 8;
    const bool l2 = h->metric == RAG_METRIC_L2;
    const long long n_tiles_ll = (h->n + kTileRows - 1) / kTileRows;
    const int n_tiles = (int)n_tiles_ll;
    // a corpus of fewer tiles than the chip has wave slots is spread over ALL CUs (the tile walk's partial round deals one
    // tile per workgroup before any gets two) instead of filling ceil(n_tiles / 8) of them: the IVF mode's coarse
    // quantizer — 4096 centroids, k = nprobe = 64 — ran as 16 workgroups of 256 rows each, three overflow sorts per query
    // (84 us); as 128 workgroups of one tile nothing overflows
    int grid = (int)std::min<long long>(h->n_cus, n_tiles_ll);
    grid = std::max(grid, 1);
    const int n_iters = (n_tiles + grid * waves - 1) / (grid * waves);

    int rc = ensure_search_ws(h, nq, k, grid);
    if (rc) return rc;
    const bool chunked = h->d8 > dc_full;
    if (chunked && !h->acc_io) {
        rc = dev_alloc(&h->acc_io, (size_t)((h->cap_rows + 31) / 32 * 32) * kQT);
        if (rc) return rc;
        h->acc_rows = h->cap_rows;
    }

    if (l2) {
        query_sqnorm_kernel<<<dim3(nq), dim3(64), (size_t)h->d8 * sizeof(float), st>>>(q_dev, h->d, h->qnorm);
        HIP_TRY(hipGetLastError());
    }

    // larger k on a large corpus: seed the filters from a sample pass (not part of the timed scan launch)
    const bool sampled = !enable && !chunked && k >= kSampleMinK && n_tiles_ll >= kSampleMinTiles;
    auto make_params = [&](int col0, int dc8) {
        ScanParams sp;
        sp.X = h->X;
        sp.xnorm = h->xnorm;
        sp.qnorm = h->qnorm;
        sp.Q = q_dev;
        sp.partial = h->partial;
        sp.ceil = ceil;
        sp.acc_io = h->acc_io;
        sp.col0 = col0;
        sp.dc8 = dc8;
        sp.acc_in = col0 > 0;
        sp.acc_out = col0 + dc8 < h->d8;
        sp.n_rows = h->n;
        sp.row_stride = h->d8;
        sp.d = h->d;
        sp.d8 = h->d8;
        sp.nq = nq;
        sp.k = k;
        sp.n_tiles = n_tiles;
        sp.n_iters = n_iters;
        sp.n_full = n_tiles / (grid * waves);
        sp.enable = enable;
        sp.epoch = epoch;
        sp.thr_key = sampled ? h->thr_keys : nullptr;
        sp.tile_step = 1;
        sp.kout = k;
        sp.x_absmax = sp.x_normmax = sp.x_scale = 0.f;
        sp.margin_out = nullptr;
        sp.lossy = nullptr;
        sp.wg_lossy = nullptr;
        sp.flag_clear = nullptr;
        sp.dyn_ctr = sp.dyn_ctr_next = nullptr;
        sp.dyn_tile0 = sp.n_dyn_groups = sp.n_singles = 0;
        sp.map = nullptr;
        sp.map_count = sp.map_ids = nullptr;
        sp.map_thr = sp.map_ticket = sp.map_done = nullptr;
#ifdef RAGK_STAMPS
        if (!g_stamps) (void)hipMalloc(reinterpret_cast<void**>(&g_stamps), 2048 * 8 * sizeof(unsigned long long));
        sp.stamps = enable ? nullptr : g_stamps;
#endif
        return sp;
    };
    if (sampled) {
        rc = run_sample_pass(h, make_params(0, h->d8), scan_fn(8, 64, pick_ring(h->d8 / 8), l2), scan_lds_bytes(h->d8, 64),
                             nq, k, st);
        if (rc) return rc;
    }

    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = h->prof && !enable;
    if (timed) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, st));
    }
    for (int col0 = 0; col0 < h->d8; col0 += dc_full) {
        const int dc8 = std::min(dc_full, h->d8 - col0);
        const int S = dc8 / 8;
        int ring = pick_ring(S);
#ifdef RAGK_TUNING
        {   // tuning overrides (scripts/tune_scan.py builds a side library with -DRAGK_TUNING)
            const int w = env_int("RAG_AMD_SCAN_WAVES", 0);
            if (w == 8 || w == 12 || w == 16) waves = w;
            const int rg = env_int("RAG_AMD_SCAN_RING", 0);
            if (ring_ok(S, rg)) ring = rg;
        }
#endif
        ScanParams sp = make_params(col0, dc8);
        if (!chunked && (rc = setup_dynamic_deal(h, sp, grid, waves))) return rc;
        ScanFn fn = scan_fn(waves, cap, ring, l2);
        const size_t lds = scan_lds_bytes(dc8, cap);
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), lds, st, sp);
        HIP_TRY(hipGetLastError());
        if (sp.dyn_ctr) ++h->dyn_slot;
    }
    if (timed) {
        HIP_TRY(hipEventRecord(e1, st));
        prof_push(h, e0, e1);
    }

    // merge: per query the top-k of the grid sorted lists, decoded into (score, id)
    if (grid > 256 * kMergeMaxOwned) return fail(RAG_ERR_UNSUPPORTED, "scan grid %d too large for the merge kernel", grid);
    KeyListSrc src{h->partial, grid, k};
    MergeOut mo{out_s, out_i, last_key, out_stride, h->qnorm, h->id_offset, h->metric, 0, qmask, epoch};
    const int look = merge_look(grid, k, k);
    launch_merge(src, grid, nq, k, look, mo, st);
    HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// Exact search of one block of <= 32 queries, any k: rounds of kmax when k exceeds what one fused pass
// can select (LDS budget), each round a full scan restricted to keys below the previous round's last key.
int search_exact_block(rag_index* h, const float* qp, int nb, int k, float* os, long long* oi, hipStream_t st,
                       const uint32_t* enable = nullptr, const uint32_t* qmask = nullptr, uint32_t epoch = 0) {
    const int kmax = rag_index_max_k(h->d, nb);
    if (kmax <= 0) return fail(RAG_ERR_UNSUPPORTED, "dimension %d is not supported by the scan kernel", h->d);
    if (k <= kmax) return search_round(h, qp, nb, k, nullptr, nullptr, os, oi, k, st, enable, qmask, epoch);
    if (!h->round_keys) {
        int rc = dev_alloc(&h->round_keys, (size_t)2 * ragk::kQT);
        if (rc) return rc;
    }
    int done = 0, flip = 0;
    while (done < k) {
        const int kr = std::min(kmax, k - done);
        const ragk::u64* ceil = done ? h->round_keys + (size_t)flip * ragk::kQT : nullptr;
        ragk::u64* last = h->round_keys + (size_t)(flip ^ 1) * ragk::kQT;
        int rc = search_round(h, qp, nb, kr, ceil, last, os + done, oi + done, k, st, enable, qmask, epoch);
        if (rc) return rc;
        done += kr;
        flip ^= 1;
    }
    return RAG_OK;
}

// Two-stage search of one block of <= 32 queries (see flat_kernels.hip.h, "two-stage exact search"):
// fp16 screening scan -> k' candidates -> canonical fp32 scores -> certificate; queries that fail it
// are answered by the fallback launches of the fp32 scan (no-ops otherwise).
constexpr int kScreenMaxK = 100;
constexpr int kScreenMaxD64 = 2048;

// LDS buffer capacity of the screening pass for (d64, k): the usual 64 / 128 / 256 by k, as long as it
// fits beside the fp16 query image (64 bytes per column: d = 1536 leaves room for k <= 48, d = 2048 for
// k <= 16); 0 = this (d, k) is answered by the fp32 scan.
int screen_capacity(int d64, int k) {
    const int cap = k <= 16 ? 64 : (k <= 48 ? 128 : 256);
    return ragk::scan_lds_bytes(d64 / 2, cap) <= 160 * 1024 ? cap : 0;
}

// `deferred_epoch` (host-pointer entry point, one block): do not enqueue the fallback launches; hand back the
// search's epoch so the caller, who synchronises anyway, can read `any_fallback` with the results and run the
// fallback only when a certificate failed (two launches, 8.5 us, saved on every batch that needs none).
// `flag_dev` (RAG_SEARCH_DEFER_FALLBACK): no fallback launches either; the resolve kernel sets *flag_dev = 1 when
// a certificate failed, and with `flag_clear` this block's scan zeroes the word first (the first block of a search).
int search_screened_block(rag_index* h, const float* qp, int nb, int k, float* os, long long* oi, hipStream_t st,
                          uint32_t* deferred_epoch = nullptr, uint32_t* flag_dev = nullptr, bool flag_clear = false) {
    using namespace ragk;
    // candidate slots per query: always the most the resolve kernel's final sort takes (empty slots cost
    // nothing: candidates are packed at the front of an LDS list and scored 32 at a time), so a dense
    // neighbourhood has to put 240 rows inside the band before the fp32 fallback is needed
    const int kp = 240;
    const int cap = screen_capacity(h->d64, k);
    const bool l2 = h->metric == RAG_METRIC_L2;
    constexpr int waves = 8;  // the screening prologue is written for 8 waves (static_assert in the kernel)
    const long long n_tiles_ll = (h->n + kTileRows - 1) / kTileRows;
    const int n_tiles = (int)n_tiles_ll;
    int grid = (int)std::min<long long>(h->n_cus, (n_tiles_ll + waves - 1) / waves);
    grid = std::max(grid, 1);
    const int n_iters = (n_tiles + grid * waves - 1) / (grid * waves);
    if (grid > 256 * kMergeMaxOwned) return fail(RAG_ERR_UNSUPPORTED, "scan grid %d too large for the merge kernel", grid);

    int rc = ensure_search_ws(h, nb, kp, grid);
    if (rc) return rc;
    if (l2) {
        query_sqnorm_kernel<<<dim3(nb), dim3(64), (size_t)h->d8 * sizeof(float), st>>>(qp, h->d, h->qnorm);
        HIP_TRY(hipGetLastError());
    }
    // one id per two-stage search: the per-search flag words are "set" when they hold it (never 0)
    if (++h->screen_epoch == 0) h->screen_epoch = 1;
    const uint32_t epoch = h->screen_epoch;

    // stage 1: screening scan over the fp16 copy (addresses in 4-byte units: 16 halves = 8 units per step)
    const bool sampled = k >= kSampleMinK && n_tiles_ll >= kSampleMinTiles;
    ScanParams sp;
    sp.X = reinterpret_cast<const float*>(h->X16);
    sp.xnorm = h->xnorm;
    sp.qnorm = h->qnorm;
    sp.Q = qp;
    sp.partial = h->partial;
    sp.ceil = nullptr;
    sp.acc_io = nullptr;
    sp.col0 = 0;
    sp.dc8 = h->d64 / 2;
    sp.acc_in = 0;
    sp.acc_out = 0;
    sp.n_rows = h->n;
    sp.row_stride = h->d64 / 2;
    sp.d = h->d;
    sp.d8 = h->d8;
    sp.nq = nb;
    sp.k = k;
    sp.n_tiles = n_tiles;
    sp.n_iters = n_iters;
    sp.n_full = n_tiles / (grid * waves);
    sp.enable = nullptr;
    sp.epoch = epoch;
    sp.thr_key = nullptr;
    sp.tile_step = 1;
    // keys a workgroup emits per query: its LDS buffer never holds more than `cap`, so longer lists would
    // be zero padding (at kp = 240 that was 15.7 MB of zeros written per batch and left dirty in L2)
    const int kout = std::min(kp, cap);
    sp.kout = kout;
    sp.x_absmax = h->x_absmax;
    sp.x_normmax = h->x_normmax;
    sp.x_scale = h->x_scale;
    sp.margin_out = h->sq->margin;
    sp.lossy = h->sq->lossy;
    sp.wg_lossy = h->wg_lossy;
    sp.flag_clear = flag_clear ? flag_dev : nullptr;
    sp.dyn_ctr = sp.dyn_ctr_next = nullptr;
    sp.dyn_tile0 = sp.n_dyn_groups = sp.n_singles = 0;
    sp.map = nullptr;
    sp.map_count = sp.map_ids = nullptr;
    sp.map_thr = sp.map_ticket = sp.map_done = nullptr;
#ifdef RAGK_STAMPS
    if (!g_stamps) (void)hipMalloc(reinterpret_cast<void**>(&g_stamps), 2048 * 8 * sizeof(unsigned long long));
    sp.stamps = g_stamps;
#endif
    const int S = h->d64 / 16;
    if (sampled) {
        rc = run_sample_pass(h, sp, screen_fn(64, S % 8 == 0 ? 8 : 4, l2), scan_lds_bytes(h->d64 / 2, 64), nb, k, st);
        if (rc) return rc;
        sp.thr_key = h->thr_keys;  // allocated by the first sample pass
    }
    if ((rc = setup_dynamic_deal(h, sp, grid, waves))) return rc;   // after the sample pass took its copy of sp
    ScanFn fn = screen_fn(cap, S % 8 == 0 ? 8 : 4, l2);
    const size_t lds = scan_lds_bytes(h->d64 / 2, cap);
    if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), lds))) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, st));
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), lds, st, sp);
    HIP_TRY(hipGetLastError());
    if (sp.dyn_ctr) ++h->dyn_slot;
    if (h->prof) {
        HIP_TRY(hipEventRecord(e1, st));
        prof_push(h, e0, e1);
    }

    // resolve: band below the k-th best approximate key -> canonical fp32 scores -> certificate -> results
    {
        const int look = merge_look(grid, kout, k);
        ResolveParams rp;
        rp.src = KeyListSrc{h->partial, grid, kout};
        rp.n_lists = grid;
        rp.k = k;
        rp.kp = kp;
        rp.look = look;
        rp.X = h->X;
        rp.row_stride = h->d8;
        rp.xnorm = h->xnorm;
        rp.Q = qp;
        rp.d = h->d;
        rp.d8 = h->d8;
        rp.l2 = l2 ? 1 : 0;
        rp.qnorm = h->qnorm;
        rp.id_offset = h->id_offset;
        rp.qs = h->sq;
        rp.wg_lossy = h->wg_lossy;
        rp.epoch = epoch;
        rp.ctr = h->sctr;
        rp.out_s = os;
        rp.out_i = oi;
        rp.flag_out = flag_dev;
        rp.id_map = nullptr;
#ifdef RAGK_STAMPS
        rp.stamps = g_stamps ? g_stamps + 2048 * 4 : nullptr;  // second half of the stamp buffer
#endif
        const size_t rlds = resolve_lds_bytes(h->d8, grid, look, kp);
        using ResolveFn = void (*)(const ResolveParams);
        ResolveFn resolve;
        if (h->d8 <= 1024)
            resolve = grid <= 256 ? (ResolveFn)screen_resolve_kernel<4, 1, 8>
                                  : (grid <= 512 ? (ResolveFn)screen_resolve_kernel<2, 4, 8> : (ResolveFn)screen_resolve_kernel<kMergeMaxOwned, 4, 8>);
        else
            resolve = grid <= 256 ? (ResolveFn)screen_resolve_kernel<4, 1, 4>
                                  : (grid <= 512 ? (ResolveFn)screen_resolve_kernel<2, 4, 4> : (ResolveFn)screen_resolve_kernel<kMergeMaxOwned, 4, 4>);
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(resolve), rlds))) return rc;
        hipLaunchKernelGGL(resolve, dim3(nb), dim3(256), rlds, st, rp);
        HIP_TRY(hipGetLastError());
    }

    // fallback: the fp32 search of this block, enqueued unconditionally, a no-op unless a certificate failed
    if (deferred_epoch) {
        *deferred_epoch = epoch;
        return RAG_OK;
    }
    if (flag_dev) return RAG_OK;  // the caller reads the word and re-runs the batch through the fp32 scan if it is set
    return search_exact_block(h, qp, nb, k, os, oi, st, &h->sq->any_fallback, h->sq->fallback, epoch);
}

int search_device_locked(rag_index* h, const float* q_dev, int nq, int k, float* out_s, long long* out_i,
                         hipStream_t st, uint32_t* deferred_epoch = nullptr, int mode = RAG_SEARCH_DEFAULT,
                         uint32_t* flag_dev = nullptr) {
    if (deferred_epoch) *deferred_epoch = 0;  // 0: nothing was deferred
    if (h->ws_used && h->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, h->ws_event, 0));
    struct Mark {  // record the workspace hand-over point on every exit path
        rag_index* h;
        hipStream_t st;
        ~Mark() {
            if (hipEventRecord(h->ws_event, st) == hipSuccess) {
                h->ws_stream = st;
                h->ws_used = true;
            }
        }
    } mark{h, st};
    const bool screened = h->n > 0 && mode != RAG_SEARCH_EXACT_ONE_PASS && h->screen_on && h->screen_valid &&
                          k <= kScreenMaxK && h->d64 <= kScreenMaxD64 && screen_capacity(h->d64, k) > 0;
    const bool flagged = mode == RAG_SEARCH_DEFER_FALLBACK && flag_dev && screened;
    // the caller's word is always written: a search that defers nothing (one-pass, or a two-stage search with its
    // fallback enqueued) is final
    if (flag_dev && !flagged) HIP_TRY(hipMemsetAsync(flag_dev, 0, sizeof(uint32_t), st));
    if (h->n == 0) {
        const int total = nq * k;
        ragk::fill_neutral_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(out_s, out_i, total, h->metric);
        HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
    for (int q0 = 0; q0 < nq; q0 += ragk::kQT) {
        const int nb = std::min(ragk::kQT, nq - q0);
        const float* qp = q_dev + (size_t)q0 * h->d;
        float* os = out_s + (size_t)q0 * k;
        long long* oi = out_i + (size_t)q0 * k;
        const bool defer = deferred_epoch && screened && nq <= ragk::kQT;  // one block: its flags stay valid until read
        int rc = screened ? search_screened_block(h, qp, nb, k, os, oi, st, defer ? deferred_epoch : nullptr,
                                                  flagged ? flag_dev : nullptr, q0 == 0)
                          : search_exact_block(h, qp, nb, k, os, oi, st);
        if (rc) return rc;
    }
    return RAG_OK;
}

// Row count and global-id range after adding n rows; the caller holds h->mu (a concurrent set_id_offset and add
// could otherwise each pass their own check and together wrap the 32-bit ids of the shard merge).
int check_add_range(const rag_index* h, long long n) {
    if (h->n + n > 0xFFFFFFFEll) return fail(RAG_ERR_UNSUPPORTED, "more than 2^32-2 rows per index");
    if (h->id_offset + h->n + n > 0xFFFFFFFFll)
        return fail(RAG_ERR_UNSUPPORTED, "id_offset %lld + %lld rows exceeds 2^32-1: global ids must fit 32 bits",
                    h->id_offset, (long long)(h->n + n));
    return RAG_OK;
}

int check_search_args(const rag_index* h, const void* q, int nq, int k, const void* s, const void* i) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    if (nq < 0 || k <= 0) return fail(RAG_ERR_INVALID_ARG, "nq=%d k=%d out of range", nq, k);
    if (nq > 0 && (!q || !s || !i)) return fail(RAG_ERR_INVALID_ARG, "null query or output buffer");
    return RAG_OK;
}

}  // namespace

// ---- library ------------------------------------------------------------------------------------

extern "C" int rag_abi_version(void) { return RAG_AMD_ABI_VERSION; }

extern "C" int rag_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* rag_last_error(void) { return g_err; }

// ---- flat index ------------------------------------------------------------------------------------

extern "C" int rag_index_create(int32_t d, int32_t metric, int32_t device, rag_index** out) {
    if (!out) return fail(RAG_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (d <= 0 || d > 65536) return fail(RAG_ERR_INVALID_ARG, "dimension %d out of range", d);
    if (metric != RAG_METRIC_INNER_PRODUCT && metric != RAG_METRIC_L2)
        return fail(RAG_ERR_INVALID_ARG, "unknown metric %d", metric);
    const int ndev = rag_device_count();
    if (ndev <= 0) return fail(RAG_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RAG_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    const int d8 = round_up(d, 8);
    DeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    rag_index* h = new (std::nothrow) rag_index();
    if (!h) return fail(RAG_ERR_OOM, "host allocation failed");
    h->device = device;
    h->d = d;
    h->d8 = d8;
    h->metric = metric;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete h;
        return fail(RAG_ERR_HIP, "hipGetDeviceProperties failed");
    }
    h->n_cus = prop.multiProcessorCount;
    h->dyn_rounds = std::max(0, std::min(1 << 20, env_int("RAG_AMD_SCAN_DYN_ROUNDS", h->dyn_rounds)));
    h->dyn_singles_per_wg = std::max(0, std::min(16, env_int("RAG_AMD_SCAN_DYN_SINGLES", h->dyn_singles_per_wg)));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ws_event, hipEventDisableTiming) != hipSuccess) {
        if (h->stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return fail(RAG_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
    }
    *out = h;
    return RAG_OK;
}

extern "C" int rag_index_destroy(rag_index* h) {
    if (!h) return RAG_OK;
    {
        DeviceGuard g(h->device);
        std::lock_guard<std::mutex> lk(h->mu);
        (void)hipDeviceSynchronize();
        for (auto& ev : h->prof_events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        void* dptrs[] = {h->X, h->xnorm, h->q_dev, h->qnorm, h->round_keys, h->acc_io, h->partial, h->out_s_dev, h->out_i_dev,
                         h->sample_heads, h->thr_keys, h->X16, h->sc_stats, h->sq, h->sctr, h->wg_lossy, h->dyn_ctrs};
        for (void* p : dptrs)
            if (p) (void)hipFree(p);
        void* hptrs[] = {h->q_pin, h->out_s_pin, h->out_i_pin, h->fb_pin};
        for (void* p : hptrs)
            if (p) (void)hipHostFree(p);
        if (h->ws_event) (void)hipEventDestroy(h->ws_event);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return RAG_OK;
}

extern "C" int rag_index_reserve(rag_index* h, int64_t n_total) {
    if (!h || n_total < 0) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (n_total > 0xFFFFFFFEll) return fail(RAG_ERR_UNSUPPORTED, "more than 2^32-2 rows per index");
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    return grow_rows(h, n_total);
}

extern "C" int rag_index_add(rag_index* h, const float* rows_host, int64_t n) {
    if (!h || n < 0 || (n > 0 && !rows_host)) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc0 = check_add_range(h, n)) return rc0;  // under the lock: set_id_offset validates and writes under it too
    int rc = grow_rows(h, h->n + n);
    if (rc) return rc;
    float* dst = h->X + (size_t)h->n * h->d8;
    if (h->d8 != h->d) HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * h->d8 * sizeof(float), h->stream));
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)h->d8 * sizeof(float), rows_host, (size_t)h->d * sizeof(float),
                             (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyHostToDevice, h->stream));
    rc = after_add(h, h->n, n, h->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->n += n;
    return RAG_OK;
}

extern "C" int rag_index_add_device(rag_index* h, const float* rows_dev, int64_t n, void* stream) {
    if (!h || n < 0 || (n > 0 && !rows_dev)) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc0 = check_add_range(h, n)) return rc0;  // under the lock: set_id_offset validates and writes under it too
    int rc = grow_rows(h, h->n + n);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;  // NULL is HIP's default stream
    float* dst = h->X + (size_t)h->n * h->d8;
    if (h->d8 != h->d) HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * h->d8 * sizeof(float), st));
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)h->d8 * sizeof(float), rows_dev, (size_t)h->d * sizeof(float),
                             (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, st));
    rc = after_add(h, h->n, n, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    h->n += n;
    return RAG_OK;
}

extern "C" int rag_index_add_synthetic(rag_index* h, int64_t n, uint64_t seed, int64_t row_number_offset) {
    if (!h || n < 0) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc0 = check_add_range(h, n)) return rc0;  // under the lock: set_id_offset validates and writes under it too
    int rc = grow_rows(h, h->n + n);
    if (rc) return rc;
    const long long chunk = 1 << 20;  // rows per generator launch (bounds the inv scratch)
    float* inv = nullptr;
    rc = dev_alloc(&inv, (size_t)std::min<long long>(chunk, n));
    if (rc) return rc;
    for (long long done = 0; done < n; done += chunk) {
        const long long m = std::min(chunk, n - done);
        const long long rn0 = row_number_offset + h->n + done;
        ragk::synth_inv_kernel<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream>>>(seed, rn0, m, h->d, inv);
        const long long total = m * h->d8;
        ragk::synth_fill_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream>>>(
            seed, rn0, m, h->d, h->d8, h->d8, inv, h->X, h->n + done);
        if (hipGetLastError() != hipSuccess) {
            (void)hipFree(inv);
            return fail(RAG_ERR_HIP, "synthetic fill launch failed");
        }
    }
    rc = after_add(h, h->n, n, h->stream);
    hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(inv);
    if (rc) return rc;
    if (e != hipSuccess) return fail(RAG_ERR_HIP, "synthetic fill: %s", hipGetErrorString(e));
    h->n += n;
    return RAG_OK;
}

extern "C" int64_t rag_index_ntotal(const rag_index* h) { return h ? h->n : 0; }
extern "C" int32_t rag_index_dim(const rag_index* h) { return h ? h->d : 0; }
extern "C" int32_t rag_index_metric(const rag_index* h) { return h ? h->metric : -1; }

extern "C" int rag_index_set_id_offset(rag_index* h, int64_t id_offset) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    // the ranking keys of the shard merge carry the GLOBAL id in 32 bits (flat_kernels.hip.h ShardListSrc):
    // an offset that can push an id past 2^32 - 2 would wrap there without a trace
    if (id_offset < 0 || id_offset > 0xFFFFFFFEll)
        return fail(RAG_ERR_UNSUPPORTED, "id_offset %lld outside [0, 2^32-2]: global ids must fit 32 bits", (long long)id_offset);
    std::lock_guard<std::mutex> lk(h->mu);
    if (id_offset + h->n > 0xFFFFFFFFll)
        return fail(RAG_ERR_UNSUPPORTED, "id_offset %lld + %lld rows exceeds 2^32-1: global ids must fit 32 bits",
                    (long long)id_offset, h->n);
    h->id_offset = id_offset;
    return RAG_OK;
}

extern "C" int rag_stream_create_masked(int32_t device, int32_t first_cu, int32_t n_cus, void** stream_out) {
    if (!stream_out) return fail(RAG_ERR_INVALID_ARG, "stream_out is null");
    *stream_out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RAG_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(RAG_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    hipDeviceProp_t prop;
    RAGC_HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int total = prop.multiProcessorCount;
    if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total)
        return fail(RAG_ERR_INVALID_ARG, "CU range [%d, %d) outside the device's %d", first_cu, first_cu + n_cus, total);
    RagcDeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
    for (int b = first_cu; b < first_cu + n_cus; ++b) mask[(size_t)b >> 5] |= 1u << (b & 31);
    hipStream_t st = nullptr;
    RAGC_HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream_out = (void*)st;
    return RAG_OK;
}

extern "C" int rag_stream_destroy(int32_t device, void* stream) {
    if (!stream) return RAG_OK;
    RagcDeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    RAGC_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    RAGC_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return RAG_OK;
}

extern "C" int rag_stream_wait(int32_t device, void* waiter, void* signaler) {
    RagcDeviceGuard g(device);
    if (!g.ok) return fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    hipEvent_t ev = nullptr;
    RAGC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaler);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);   // (released once the recorded point has passed)
    if (e != hipSuccess) return fail(RAG_ERR_HIP, "stream wait failed: %s", hipGetErrorString(e));
    return RAG_OK;
}

extern "C" int rag_device_cu_count(int32_t device, int32_t* n_cus) {
    if (!n_cus) return fail(RAG_ERR_INVALID_ARG, "n_cus is null");
    hipDeviceProp_t prop;
    RAGC_HIP_TRY(hipGetDeviceProperties(&prop, device));
    *n_cus = prop.multiProcessorCount;
    return RAG_OK;
}

extern "C" int rag_index_set_cu_budget(rag_index* h, int32_t n_cus) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    hipDeviceProp_t prop;
    RAGC_HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    if (n_cus < 0 || n_cus > prop.multiProcessorCount)
        return fail(RAG_ERR_INVALID_ARG, "CU budget %d outside [0, %d]", n_cus, prop.multiProcessorCount);
    std::lock_guard<std::mutex> lk(h->mu);
    h->n_cus = n_cus == 0 ? prop.multiProcessorCount : n_cus;
    return RAG_OK;
}

extern "C" int rag_index_set_screening(rag_index* h, int32_t mode) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    if (mode != RAG_SCREEN_OFF && mode != RAG_SCREEN_FP16) return fail(RAG_ERR_INVALID_ARG, "unknown screening mode %d", mode);
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipDeviceSynchronize());  // searches in flight may still read the copy
    if (mode == RAG_SCREEN_OFF) {
        if (h->X16) (void)hipFree(h->X16);
        h->X16 = nullptr;
        h->screen_on = h->screen_valid = false;
        h->x_scale = 0.f;
        return RAG_OK;
    }
    if (h->screen_on) return RAG_OK;
    const int d64 = round_up(h->d, 64);
    if (d64 > kScreenMaxD64)
        return fail(RAG_ERR_UNSUPPORTED, "two-stage search covers d <= %d (d = %d)", kScreenMaxD64, h->d);
    int rc;
    if (!h->sc_stats && (rc = dev_alloc(&h->sc_stats, 1))) return rc;
    if (!h->sq && (rc = dev_alloc(&h->sq, 1))) return rc;
    if (!h->sctr && (rc = dev_alloc(&h->sctr, 1))) return rc;
    HIP_TRY(hipMemset(h->sc_stats, 0, sizeof(ragk::ScreenCorpusStats)));
    HIP_TRY(hipMemset(h->sq, 0, sizeof(ragk::ScreenQueryState)));
    HIP_TRY(hipMemset(h->sctr, 0, sizeof(ragk::ScreenCounters)));
    if (!h->wg_lossy && (rc = dev_alloc(&h->wg_lossy, (size_t)ragk::kQT * kWgLossyLists))) return rc;
    HIP_TRY(hipMemset(h->wg_lossy, 0, (size_t)ragk::kQT * kWgLossyLists * sizeof(uint32_t)));
    if (h->cap_rows > 0 && (rc = dev_alloc(&h->X16, (size_t)h->cap_rows * d64))) return rc;
    h->d64 = d64;
    h->screen_on = true;
    h->screen_valid = false;
    h->x_scale = 0.f;
    rc = screen_sync_rows(h, 0, h->n, h->stream);
    if (rc) {
        if (h->X16) (void)hipFree(h->X16);
        h->X16 = nullptr;
        h->screen_on = false;
    }
    return rc;
}

extern "C" int32_t rag_index_screening(const rag_index* h) {
    if (!h || !h->screen_on) return RAG_SCREEN_OFF;
    return h->screen_valid ? RAG_SCREEN_FP16 : RAG_SCREEN_INACTIVE;
}

extern "C" int rag_index_screen_stats(rag_index* h, int64_t* queries, int64_t* fallbacks, double* max_err_ratio,
                                      int32_t reset) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    ragk::ScreenCounters c{};
    if (h->sctr) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&c, h->sctr, sizeof c, hipMemcpyDeviceToHost));
        if (reset) HIP_TRY(hipMemset(h->sctr, 0, sizeof c));
    }
    float ratio;
    std::memcpy(&ratio, &c.max_err_ratio_bits, 4);
    if (queries) *queries = (int64_t)c.queries;
    if (fallbacks) *fallbacks = (int64_t)c.fallbacks;
    if (max_err_ratio) *max_err_ratio = ratio;
    return RAG_OK;
}

extern "C" int32_t rag_index_max_k(int32_t d, int32_t nq) {
    (void)nq;
    if (d <= 0) return 0;
    const int dc = chunk_cols(round_up(d, 8));
    int best = 0;
    for (int c : {64, 128, 256})
        if (ragk::scan_lds_bytes(dc, c) <= 160 * 1024) best = c - 16;
    return best;
}

extern "C" int rag_index_search_device(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                                       float* out_scores_dev, int64_t* out_ids_dev, void* stream) {
    int rc = check_search_args(h, queries_dev, nq, k, out_scores_dev, out_ids_dev);
    if (rc) return rc;
    if (nq == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;  // exactly the caller's stream; NULL is HIP's default stream
    return search_device_locked(h, queries_dev, nq, k, out_scores_dev, reinterpret_cast<long long*>(out_ids_dev), st);
}

extern "C" int rag_index_search_device_ex(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                                          float* out_scores_dev, int64_t* out_ids_dev, int32_t mode,
                                          uint32_t* flag_dev, void* stream) {
    int rc = check_search_args(h, queries_dev, nq, k, out_scores_dev, out_ids_dev);
    if (rc) return rc;
    if (mode != RAG_SEARCH_DEFAULT && mode != RAG_SEARCH_EXACT_ONE_PASS && mode != RAG_SEARCH_DEFER_FALLBACK)
        return fail(RAG_ERR_INVALID_ARG, "unknown search mode %d", mode);
    if (mode == RAG_SEARCH_DEFER_FALLBACK && !flag_dev)
        return fail(RAG_ERR_INVALID_ARG, "RAG_SEARCH_DEFER_FALLBACK needs a device word for the flag");
    if (nq == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    return search_device_locked(h, queries_dev, nq, k, out_scores_dev, reinterpret_cast<long long*>(out_ids_dev),
                                (hipStream_t)stream, nullptr, mode, flag_dev);
}

namespace {
int ensure_host_out(rag_index* h, size_t on) {
    int rc;
    if (on > h->out_cap) {
        if (h->out_s_dev) (void)hipFree(h->out_s_dev);
        if (h->out_i_dev) (void)hipFree(h->out_i_dev);
        h->out_s_dev = nullptr;
        h->out_i_dev = nullptr;
        h->out_cap = 0;
        rc = dev_alloc(&h->out_s_dev, on);
        if (rc) return rc;
        rc = dev_alloc(&h->out_i_dev, on);
        if (rc) return rc;
        h->out_cap = on;
    }
    if (on > h->out_pin_cap) {
        if (h->out_s_pin) (void)hipHostFree(h->out_s_pin);
        if (h->out_i_pin) (void)hipHostFree(h->out_i_pin);
        h->out_s_pin = nullptr;
        h->out_i_pin = nullptr;
        h->out_pin_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->out_s_pin), on * sizeof(float), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->out_i_pin), on * sizeof(long long), hipHostMallocDefault));
        h->out_pin_cap = on;
    }
    if (!h->fb_pin) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->fb_pin), sizeof(uint32_t), hipHostMallocDefault));
    return RAG_OK;
}

// Results of the search just enqueued on `st` -> the caller's host buffers: one D2H per array, one sync; with a
// deferred two-stage fallback (`deferred` = its epoch) the flag word rides along and the fp32 scan runs only
// when a certificate failed.
int finish_to_host(rag_index* h, const float* q_dev, int nq, int k, uint32_t deferred, float* out_scores, int64_t* out_ids,
                   hipStream_t st) {
    const size_t on = (size_t)nq * k;
    HIP_TRY(hipMemcpyAsync(h->out_s_pin, h->out_s_dev, on * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h->out_i_pin, h->out_i_dev, on * sizeof(long long), hipMemcpyDeviceToHost, st));
    if (deferred) HIP_TRY(hipMemcpyAsync(h->fb_pin, &h->sq->any_fallback, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (deferred && *h->fb_pin == deferred) {
        // a certificate failed: the fp32 search of the block, restricted on the device to the flagged queries
        int rc = search_exact_block(h, q_dev, nq, k, h->out_s_dev, h->out_i_dev, st, &h->sq->any_fallback,
                                    h->sq->fallback, deferred);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(h->out_s_pin, h->out_s_dev, on * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h->out_i_pin, h->out_i_dev, on * sizeof(long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    std::memcpy(out_scores, h->out_s_pin, on * sizeof(float));
    std::memcpy(out_ids, h->out_i_pin, on * sizeof(long long));
    return RAG_OK;
}
}  // namespace

extern "C" int rag_index_search_device_host_out(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                                                float* out_scores, int64_t* out_ids, void* stream) {
    int rc = check_search_args(h, queries_dev, nq, k, out_scores, out_ids);
    if (rc) return rc;
    if (nq == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;
    rc = ensure_host_out(h, (size_t)nq * k);
    if (rc) return rc;
    uint32_t deferred = 0;
    rc = search_device_locked(h, queries_dev, nq, k, h->out_s_dev, h->out_i_dev, st, &deferred);
    if (rc) return rc;
    return finish_to_host(h, queries_dev, nq, k, deferred, out_scores, out_ids, st);
}

extern "C" int rag_index_search(rag_index* h, const float* queries_host, int32_t nq, int32_t k, float* out_scores,
                                int64_t* out_ids) {
    int rc = check_search_args(h, queries_host, nq, k, out_scores, out_ids);
    if (rc) return rc;
    if (nq == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t qn = (size_t)nq * h->d, on = (size_t)nq * k;
    if (qn > h->q_dev_cap) {
        if (h->q_dev) (void)hipFree(h->q_dev);
        h->q_dev = nullptr;
        h->q_dev_cap = 0;
        rc = dev_alloc(&h->q_dev, qn);
        if (rc) return rc;
        h->q_dev_cap = qn;
    }
    if (qn > h->q_pin_cap) {
        if (h->q_pin) (void)hipHostFree(h->q_pin);
        h->q_pin = nullptr;
        h->q_pin_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->q_pin), qn * sizeof(float), hipHostMallocDefault));
        h->q_pin_cap = qn;
    }
    rc = ensure_host_out(h, on);
    if (rc) return rc;
    std::memcpy(h->q_pin, queries_host, qn * sizeof(float));
    HIP_TRY(hipMemcpyAsync(h->q_dev, h->q_pin, qn * sizeof(float), hipMemcpyHostToDevice, h->stream));
    uint32_t deferred = 0;
    rc = search_device_locked(h, h->q_dev, nq, k, h->out_s_dev, h->out_i_dev, h->stream, &deferred);
    if (rc) return rc;
    return finish_to_host(h, h->q_dev, nq, k, deferred, out_scores, out_ids, h->stream);
}

extern "C" int rag_index_get_rows(rag_index* h, int64_t row0, int64_t n, float* out_rows_host) {
    if (!h || row0 < 0 || n < 0 || (n > 0 && !out_rows_host)) return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (row0 + n > h->n) return fail(RAG_ERR_INVALID_ARG, "rows [%lld, %lld) beyond ntotal %lld", (long long)row0,
                                     (long long)(row0 + n), h->n);
    if (n == 0) return RAG_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipMemcpy2DAsync(out_rows_host, (size_t)h->d * sizeof(float), h->X + (size_t)row0 * h->d8,
                             (size_t)h->d8 * sizeof(float), (size_t)h->d * sizeof(float), (size_t)n,
                             hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RAG_OK;
}

extern "C" int rag_index_profile_enable(rag_index* h, int32_t on) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->prof = on != 0;
    return RAG_OK;
}

extern "C" int rag_index_profile(rag_index* h, double* scan_ms_total, int64_t* scan_launches, int32_t reset) {
    if (!h) return fail(RAG_ERR_INVALID_ARG, "null index handle");
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto& ev : h->prof_events) {
        HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        h->prof_ms += ms;
        h->prof_launches += 1;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->prof_events.clear();
    if (scan_ms_total) *scan_ms_total = h->prof_ms;
    if (scan_launches) *scan_launches = h->prof_launches;
    if (reset) {
        h->prof_ms = 0.0;
        h->prof_launches = 0;
    }
    return RAG_OK;
}

// ---- shard merge ----------------------------------------------------------------------------------

namespace {
int merge_shards(int device, int metric, int n_shards, int nq, int k, const float* scores, long long score_stride,
                 const long long* ids, long long id_stride, float* out_s, long long* out_i, void* stream,
                 const uint32_t* flags = nullptr, long long flag_stride = 0, uint32_t* any_flag = nullptr,
                 float* host_s = nullptr, long long* host_i = nullptr, uint32_t* host_any = nullptr) {
    using namespace ragk;
    if (n_shards <= 0 || nq < 0 || k <= 0 || !scores || !ids || !out_s || !out_i)
        return fail(RAG_ERR_INVALID_ARG, "bad arguments");
    if (metric != RAG_METRIC_INNER_PRODUCT && metric != RAG_METRIC_L2)
        return fail(RAG_ERR_INVALID_ARG, "unknown metric %d", metric);
    if (n_shards > 256 * kMergeMaxOwned) return fail(RAG_ERR_UNSUPPORTED, "more than %d shards", 256 * kMergeMaxOwned);
    const int ndev = rag_device_count();
    if (ndev <= 0) return fail(RAG_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RAG_ERR_NO_DEVICE, "device %d out of range", device);
    if (nq == 0) return RAG_OK;
    DeviceGuard g(device);
    ShardListSrc src{scores, ids, score_stride, id_stride, k, metric};
    MergeOut mo{out_s, out_i, nullptr, k, nullptr, 0, metric, 1, nullptr, 0, flags, flag_stride, n_shards, any_flag,
                host_s, host_i, host_any};
    const int look = merge_look(n_shards, k, k);
    launch_merge(src, n_shards, nq, k, look, mo, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RAG_OK;
}
}  // namespace

extern "C" int rag_merge_topk_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                                     const float* scores_dev, const int64_t* ids_dev, float* out_scores_dev,
                                     int64_t* out_ids_dev, void* stream) {
    return merge_shards(device, metric, n_shards, nq, k, scores_dev, (long long)nq * k,
                        reinterpret_cast<const long long*>(ids_dev), (long long)nq * k, out_scores_dev,
                        reinterpret_cast<long long*>(out_ids_dev), stream);
}

extern "C" int rag_merge_topk_packed_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                                            const void* packed_dev, int64_t shard_stride_bytes,
                                            int64_t scores_offset_bytes, float* out_scores_dev, int64_t* out_ids_dev,
                                            void* stream) {
    if (!packed_dev || shard_stride_bytes % 8 || scores_offset_bytes % 4 || shard_stride_bytes <= 0)
        return fail(RAG_ERR_INVALID_ARG, "packed layout must keep ids 8-byte and scores 4-byte aligned");
    const char* base = static_cast<const char*>(packed_dev);
    return merge_shards(device, metric, n_shards, nq, k, reinterpret_cast<const float*>(base + scores_offset_bytes),
                        shard_stride_bytes / 4, reinterpret_cast<const long long*>(base), shard_stride_bytes / 8,
                        out_scores_dev, reinterpret_cast<long long*>(out_ids_dev), stream);
}

extern "C" int rag_merge_topk_packed_flagged_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                                                    const void* packed_dev, int64_t shard_stride_bytes,
                                                    int64_t scores_offset_bytes, int64_t flag_offset_bytes,
                                                    float* out_scores_dev, int64_t* out_ids_dev,
                                                    uint32_t* any_flag_dev, void* host_mirror, void* stream) {
    if (!packed_dev || shard_stride_bytes % 8 || scores_offset_bytes % 4 || shard_stride_bytes <= 0)
        return fail(RAG_ERR_INVALID_ARG, "packed layout must keep ids 8-byte and scores 4-byte aligned");
    if (!any_flag_dev || flag_offset_bytes < 0 || flag_offset_bytes % 4 || flag_offset_bytes + 4 > shard_stride_bytes)
        return fail(RAG_ERR_INVALID_ARG, "flag word must be a 4-byte aligned word inside every shard's block");
    if (nq <= 0) return fail(RAG_ERR_INVALID_ARG, "nq must be positive (the flags are reduced by query 0's workgroup)");
    if (host_mirror && (reinterpret_cast<uintptr_t>(host_mirror) % 8))
        return fail(RAG_ERR_INVALID_ARG, "host mirror must be 8-byte aligned");
    const char* base = static_cast<const char*>(packed_dev);
    char* hm = static_cast<char*>(host_mirror);
    return merge_shards(device, metric, n_shards, nq, k, reinterpret_cast<const float*>(base + scores_offset_bytes),
                        shard_stride_bytes / 4, reinterpret_cast<const long long*>(base), shard_stride_bytes / 8,
                        out_scores_dev, reinterpret_cast<long long*>(out_ids_dev), stream,
                        reinterpret_cast<const uint32_t*>(base + flag_offset_bytes), shard_stride_bytes / 4, any_flag_dev,
                        hm ? reinterpret_cast<float*>(hm + scores_offset_bytes) : nullptr, reinterpret_cast<long long*>(hm),
                        hm ? reinterpret_cast<uint32_t*>(hm + flag_offset_bytes) : nullptr);
}


// ---- the shard step in one call (C1) ------------------------------------------------------------------

extern "C" int rag_pack_layout(int32_t nq, int32_t k, int64_t* scores_offset, int64_t* flag_offset, int64_t* block_bytes) {
    if (nq <= 0 || k <= 0) return fail(RAG_ERR_INVALID_ARG, "nq=%d k=%d out of range", nq, k);
    const int64_t n = (int64_t)nq * k;
    if (scores_offset) *scores_offset = 8 * n;
    if (flag_offset) *flag_offset = 12 * n;
    if (block_bytes) *block_bytes = (12 * n + 4 + 7) / 8 * 8;
    return RAG_OK;
}

extern "C" int rag_index_search_gather_device(rag_index* h, rag_comm* c, const float* queries_dev, int32_t nq, int32_t k,
                                              int32_t mode, void* pack_dev, void* gathered_dev, float* out_scores_dev,
                                              int64_t* out_ids_dev, uint32_t* any_flag_dev, void* host_mirror,
                                              void* stream, void* comm_stream) {
    if (!c) return fail(RAG_ERR_INVALID_ARG, "null communicator");
    if (nq <= 0) return fail(RAG_ERR_INVALID_ARG, "nq must be positive");
    if (!pack_dev || !gathered_dev || !any_flag_dev) return fail(RAG_ERR_INVALID_ARG, "null pack / gather / flag buffer");
    if (reinterpret_cast<uintptr_t>(pack_dev) % 8 || reinterpret_cast<uintptr_t>(gathered_dev) % 8)
        return fail(RAG_ERR_INVALID_ARG, "packed blocks must be 8-byte aligned");
    if (mode != RAG_SEARCH_DEFAULT && mode != RAG_SEARCH_EXACT_ONE_PASS && mode != RAG_SEARCH_DEFER_FALLBACK)
        return fail(RAG_ERR_INVALID_ARG, "unknown search mode %d", mode);
    int64_t s_off = 0, f_off = 0, blk = 0;
    int rc = rag_pack_layout(nq, k, &s_off, &f_off, &blk);
    if (rc) return rc;
    char* pack = static_cast<char*>(pack_dev);
    float* pack_s = reinterpret_cast<float*>(pack + s_off);
    long long* pack_i = reinterpret_cast<long long*>(pack);
    uint32_t* pack_f = reinterpret_cast<uint32_t*>(pack + f_off);
    rc = check_search_args(h, queries_dev, nq, k, pack_s, pack_i);
    if (rc) return rc;
    if (ragc_comm_device(c) != h->device) return fail(RAG_ERR_INVALID_ARG, "communicator lives on device %d, the index on %d", ragc_comm_device(c), h->device);
    const int world = ragc_comm_world(c);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipStream_t cst = comm_stream ? (hipStream_t)comm_stream : st;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        // the flag word of the block is written in every mode (0: this rank's list is final)
        rc = search_device_locked(h, queries_dev, nq, k, pack_s, pack_i, st, nullptr, mode, pack_f);
        if (rc) return rc;
        if (cst != st) {   // the collective and the merge start when the local search has finished (h->ws_event was just recorded on st)
            HIP_TRY(hipStreamWaitEvent(cst, h->ws_event, 0));
        }
        rc = ragc_comm_all_gather(c, pack_dev, gathered_dev, (size_t)blk, cst);
        if (rc) return rc;
    }
    return rag_merge_topk_packed_flagged_device(h->device, h->metric, world, nq, k, gathered_dev, blk, s_off, f_off, out_scores_dev,
                                                out_ids_dev, any_flag_dev, host_mirror, (void*)cst);
}

#include "rag_ivf_host.hip.h"   // the IVFFlat nprobe mode (rag_ivf_*)

#ifdef RAGK_STAMPS
// experiment build: phase stamps of the last stamped scan launch, [n_wg][8] (device-synchronising)
extern "C" int rag_debug_scan_stamps(unsigned long long* out, int32_t n_wg) {
    if (!g_stamps || !out || n_wg <= 0 || n_wg > 2048) return RAG_ERR_INVALID_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return RAG_ERR_HIP;
    if (hipMemcpy(out, g_stamps, (size_t)n_wg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return RAG_ERR_HIP;
    return RAG_OK;
}
// phase stamps of the last resolve launch, [nq][8]
extern "C" int rag_debug_resolve_stamps(unsigned long long* out, int32_t nq) {
    if (!g_stamps || !out || nq <= 0 || nq > 32) return RAG_ERR_INVALID_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return RAG_ERR_HIP;
    if (hipMemcpy(out, g_stamps + 2048 * 4, (size_t)nq * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return RAG_ERR_HIP;
    return RAG_OK;
}
#endif
