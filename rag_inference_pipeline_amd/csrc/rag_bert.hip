// rag_bert.hip — C ABI (rag_bert_* in include/rag_amd.h) over the kernels in bert_kernels.hip.h.
// Runs a BERT-family encoder over packed sequences: the query embedder (mean / CLS pooling +
// L2 normalisation) and the cross-encoder reranker (classifier head + sigmoid).
// Weights stay in caller-owned device memory (PyTorch-ROCm tensors); this file owns only the
// activation workspace.  No CPU compute path.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "bert_kernels.hip.h"
#include "rag_common.h"

namespace {

constexpr int kPerLayer = 12;
constexpr int kEmbEntries = 5;
constexpr int kGraphTokens = 1024;   // a graphed pass holds at most this many (padded) tokens: the small-batch kernels' range
constexpr int kGraphSeqs = 72;       // ... and this many (padded) sequences
constexpr int kGraphCache = 24;      // instantiated graphs kept per model (least recently used goes)
constexpr int kGraphCellsAt = 2 * kGraphTokens + (kGraphSeqs + 1 + 1) / 2 * 2;   // ints: the 8-byte cells start 8-byte aligned
constexpr int kGraphBlockInts = kGraphCellsAt + 8;
constexpr int kGraphStages = 2;      // pinned staging blocks that take turns

int weight_count(const rag_bert_config& c) { return kEmbEntries + kPerLayer * c.n_layers + (c.head != RAG_HEAD_NONE ? 4 : 0); }

int map_act(int act) {
    switch (act) {
        case RAG_ACT_GELU: return ragb::ACT_GELU_ERF;
        case RAG_ACT_GELU_TANH: return ragb::ACT_GELU_TANH;
        case RAG_ACT_RELU: return ragb::ACT_RELU;
        default: return -1;
    }
}

}  // namespace

struct rag_bert {
    rag_bert_config cfg{};
    int device = 0;
    int n_cus = 256;
    bool valu_attention = false;  // RAG_AMD_VALU_ATTENTION=1: the VALU attention kernel (A/B checks)
    bool row_major_only = false;  // RAG_AMD_ROW_MAJOR=1: RAG_GEMM_F16 keeps fp32 row-major activations (A/B checks)
    bool tiled_attention_f32 = false;  // RAG_AMD_TILED_ATTENTION_F32=1: fp32-MFMA attention on the tiled path (A/B checks)
    std::vector<const float*> w;
    // Fragment-order images of the four GEMM weights of every layer, owned here (gemm_wl.hip.h):
    //   wx2  RAG_GEMM_F32: two fp16 planes (hi, scaled lo) — the default big-batch path
    //   wx   RAG_GEMM_F32: three bf16 planes — built only when a value outside fp16's range has been seen
    //   wxf  RAG_GEMM_F16: one fp16 plane
    std::vector<_Float16*> wx2;
    std::vector<__bf16*> wx;
    std::vector<_Float16*> wxf;
    std::vector<_Float16*> wffn2p;    // RAG_GEMM_F16, hidden 384: W2 of every layer in accumulator column order (ffn_fused_t16.hip.h)
    // LayerNorm folded into its consumers (gemm_wl.hip.h; query-encoder path, <= 1024 tokens): per layer the two-plane
    // images of the QKV and feed-forward input weights with the preceding LayerNorm's gamma folded in, and for each the
    // vectors s[n] = sum_k W'[n][k] and bias2 = bias + W beta.  Built at create when the model qualifies (lnf_ok).
    std::vector<_Float16*> wfold;     // [2 l] QKV, [2 l + 1] feed-forward input
    std::vector<float*> fold_s, fold_b;
    bool lnf_ok = false;          // the query-encoder form (<= 1024 tokens) is on: RAG_AMD_ENCODER_LN_FOLD=1
    bool lnf_built = false;       // the folded images exist
    bool use_w6 = false;          // RAG_AMD_GEMM_W6=1: big-batch two-plane GEMMs on 128 x 192 tiles (measured slower: off)
    bool lnf_big = false;         // the big-batch form (> 1024 tokens, gemm_nt_wl_kernel) is on: the default, RAG_AMD_LN_FOLD=0 turns it off
    float* y1 = nullptr;              // [ws_tokens][H]: the second pre-LayerNorm row buffer (x is the first)
    float2 *ts_a = nullptr, *ts_b = nullptr;   // [ws_tokens][H / 32] block statistics of x / y1
    float2 *rs_a = nullptr, *rs_b = nullptr;   // [ws_tokens] finished (mean, rstd) of x / y1 (big-batch form)
    unsigned* tile_ctr = nullptr;     // arrival counters of the split-K tiles (zero between launches)
    uint32_t* range_pin = nullptr;    // pinned host word the GEMM kernels write themselves (posted store): a two-plane
                                      // GEMM met |a| >= 65504 (or a weight did, at creation); read after a sync
    bool weights_fit_f16 = true;      // every GEMM weight is inside fp16's range: the two-plane images are valid
    bool force_x6 = false;            // this forward pass runs its big-batch GEMMs on the split-bf16 images
    bool background = false;          // rag_bert_set_background
    long long range_events = 0;       // forward passes repeated on the split-bf16 path
    hipStream_t stream = nullptr;         // the handle's own stream
    hipStream_t user_stream = nullptr;    // rag_bert_set_stream: the caller's stream for the host-ids entry points (not owned)
    std::mutex mu;
    // activation workspace, sized for ws_tokens tokens / ws_seqs sequences
    long long ws_tokens = 0, ws_seqs = 0;
    float *x = nullptr, *y = nullptr, *qkv = nullptr, *ctx = nullptr, *ffn = nullptr;
    float *pooled = nullptr, *pooled2 = nullptr, *logits = nullptr, *probs = nullptr;
    // host-path staging
    int *ids_dev = nullptr, *types_dev = nullptr, *cu_dev = nullptr;
    long long ids_cap = 0, types_cap = 0, cu_cap = 0;
    float* out_dev = nullptr;
    long long out_cap = 0;
    // pinned staging of the asynchronous host-ids entry point: [ids | type_ids | cu_seqlens]
    int* stage_pin = nullptr;
    long long stage_cap = 0;
    hipEvent_t stage_event = nullptr;   // the staging buffer's last upload
    bool stage_used = false;
    // the activation workspace is shared by every stream forwards are issued on: a forward that follows one
    // on a different stream first waits for ws_event (same rule as the index's search workspace)
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    // ---- the query encoder as a hipGraph (rag_bert_forward_to_device, small batches): one replay per call instead of
    // ~45-90 kernel launches.  Shapes are padded to buckets with dummy sequences; every address in the graph is fixed:
    // ids / type ids / cu_seqlens and three "cells" (where the result goes, where the range flag goes, how many floats)
    // travel through one pinned block, the last node copies the pooled rows out through the cell.
    struct EncGraph {
        int nseq_pad, T_pad, maxlen_b, out_kind, normalize;
        hipGraphExec_t exec;
        unsigned long long last_use;
    };
    std::vector<EncGraph> graphs;
    unsigned long long graph_clock = 0;
    bool use_graphs = true;           // RAG_AMD_ENCODER_GRAPH=0 turns the path off
    bool in_capture = false;          // forward_locked is being captured: no event traffic, no reallocation
    // One block layout, host and device: [ids kGraphTokens | types kGraphTokens | cu kGraphSeqs + 1 (+ pad) | cells 8 ints].
    // The upload of a call's block is ONE copy enqueued in front of the replay (outside the graph, so the graph holds no
    // host address); kGraphStages pinned blocks take turns, each with the event of its last upload, so a call fills its
    // block while the previous call's pass is still running (round 3 had one block and waited for the whole pass).
    int* g_pin[kGraphStages] = {};
    hipEvent_t g_up[kGraphStages] = {};
    bool g_up_used[kGraphStages] = {};
    unsigned g_turn = 0;
    int* g_blk = nullptr;             // device copy of the block (passes on one stream run in order, so one is enough)
    int *g_ids = nullptr, *g_types = nullptr, *g_cu = nullptr;   // views into g_blk
    unsigned long long* g_cells = nullptr;   // view: out pointer, flag pointer, float count
    float* g_out = nullptr;           // [kGraphSeqs][hidden]
    uint32_t* g_flag = nullptr;       // the range flag of a graph pass (device memory; published through the cell)
    unsigned long long ws_generation = 0;    // bumped whenever ensure_ws reallocates: cached graphs hold the old addresses
};

namespace {

template <typename T>
int grow(T** p, long long* cap, long long want) {
    if (want <= *cap) return RAG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), (size_t)want * sizeof(T)));
    *cap = want;
    return RAG_OK;
}

// Cached encoder graphs hold the workspace's addresses as kernel arguments: once a buffer is reallocated every one of
// them would replay into freed memory.  Callers have synchronised (nothing is in flight) before ensure_ws reallocates.
void drop_graphs(rag_bert* h) {
    for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);
    h->graphs.clear();
}

int ensure_ws(rag_bert* h, long long tokens, long long nseq) {
    const rag_bert_config& c = h->cfg;
    if (tokens > h->ws_tokens || nseq > h->ws_seqs) {
        drop_graphs(h);
        ++h->ws_generation;
    }
    if (tokens > h->ws_tokens) {
        float** bufs[] = {&h->x, &h->y, &h->qkv, &h->ctx, &h->ffn, &h->y1};
        for (float** b : bufs) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
        float2** tsb[] = {&h->ts_a, &h->ts_b, &h->rs_a, &h->rs_b};
        for (float2** b : tsb) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
        h->ws_tokens = 0;
        const long long t = (tokens + tokens / 8 + 31) & ~31LL;   // (whole 32-token row blocks: the tiled layout's unit)
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->x), (size_t)t * c.hidden * sizeof(float)));
        // y doubles as the split-K slab area of the small-M path (kMaxSplits slabs)
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->y), (size_t)std::max<long long>(t, 8 * std::min<long long>(t, 1152)) * c.hidden * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->qkv), (size_t)t * 3 * c.hidden * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ctx), (size_t)t * c.hidden * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ffn), (size_t)t * c.intermediate * sizeof(float)));
        if (h->lnf_ok || h->lnf_big) {
            // block statistics: 32-column blocks for up to 1024 tokens (query encoder), 128-column blocks beyond
            const long long ts_rows = std::min<long long>(t, 1056);
            const size_t ts_n = std::max<size_t>((size_t)ts_rows * (c.hidden / 32), (size_t)t * ((c.hidden + 63) / 64));
            if (h->lnf_ok) RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->y1), (size_t)ts_rows * c.hidden * sizeof(float)));
            RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ts_a), ts_n * sizeof(float2)));
            RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ts_b), ts_n * sizeof(float2)));
            if (h->lnf_big) {
                RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->rs_a), (size_t)t * sizeof(float2)));
                RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->rs_b), (size_t)t * sizeof(float2)));
            }
        }
        h->ws_tokens = t;
    }
    if (nseq > h->ws_seqs) {
        float** bufs[] = {&h->pooled, &h->pooled2, &h->logits, &h->probs};
        for (float** b : bufs) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
        h->ws_seqs = 0;
        const long long s = nseq + nseq / 8;
        const int nl = std::max(1, c.n_labels);
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->pooled), (size_t)s * c.hidden * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->pooled2), (size_t)s * c.hidden * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->logits), (size_t)s * nl * sizeof(float)));
        RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->probs), (size_t)s * nl * sizeof(float)));
        h->ws_seqs = s;
    }
    return RAG_OK;
}

// The three forms a GEMM weight can take: fp32 (always), an fp16 copy (RAG_GEMM_F16, caller-supplied),
// a split-bf16 image (RAG_GEMM_F32, built at create).
struct WRef {
    const float* w;
    const _Float16* wx2;   // two fp16 planes, fragment order (null: not available / not valid)
    const __bf16* wx;      // three bf16 planes, fragment order (null: not built)
    const _Float16* wxf;   // one fp16 plane, fragment order (RAG_GEMM_F16)
    uint32_t* range_flag;  // this pass's "outside fp16's range" word (pinned host memory the device writes)
    bool background;       // rag_bert_set_background: small-batch GEMMs in their 32-KiB-LDS form
    bool w6 = false;       // big-batch two-plane GEMMs on 128 x 192 tiles (RAG_AMD_GEMM_W6=1)
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel (sizes above 64 KiB need it)
int ensure_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, const void*>, int>> granted;
    int dev = 0;
    RAGC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (auto& g : granted)
        if (g.first.first == dev && g.first.second == fn && g.second >= bytes) return RAG_OK;
    RAGC_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    granted.push_back({{dev, fn}, bytes});
    return RAG_OK;
}

template <int MODE, int KS, int NS, bool PIPE>
int launch_wl(const ragb::GemmWlParams& g, hipStream_t st) {
    using Geo = ragb::WlGeom<MODE, KS, NS>;
    auto fn = &ragb::gemm_nt_wl_kernel<MODE, KS, NS, PIPE>;
    int rc = ensure_lds(reinterpret_cast<const void*>(fn), Geo::LDS);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(ragb::xcd_grid(g.M, g.N, 128, 128)), dim3(256), Geo::LDS, st, g);
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// The default mode's big-batch GEMM (two fp16 planes per operand): 128 x 128 tiles; with `w6` (RAG_AMD_GEMM_W6=1 at
// rag_bert_create) 128 x 192 tiles where 192 divides N.  The wider tile was built on the reading that the kernel is bound by
// LDS-DMA instructions per MFMA (0.28 instead of 0.33 at the same two workgroups per CU) and measured 0-7 % SLOWER on every
// shape (round 4, gpurun_out/r4o: 568 vs 531 us at N = 384, 1082 vs 1038 at N = 1536, 673 vs 671 at N = 1152) — the reading
// was wrong, or 254 registers leave the scheduler nothing to work with.  Kept as an opt-in with its tests; off by default.
int launch_wl2(const ragb::GemmWlParams& g, hipStream_t st, bool w6) {
    if (w6 && g.N % 192 == 0) {
        auto fn = &ragb::gemm_nt_w6_kernel;
        int rc = ensure_lds(reinterpret_cast<const void*>(fn), ragb::W6Geom::LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(fn, dim3(ragb::xcd_grid(g.M, g.N, 128, 192)), dim3(256), ragb::W6Geom::LDS, st, g);
        RAGC_HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
    return launch_wl<2, 1, 3, false>(g, st);
}
// columns per block of the big-batch tile statistics a model of hidden size H gets (the kernel its N = H GEMMs run on)
int lnf_big_blkw(int H, bool w6) { return w6 && H % 192 == 0 ? 64 : 128; }

template <int AK, int NW, int NB, int KS, int NS, int OCC>
int launch_wt(const ragb::GemmWtParams& g, hipStream_t st) {
    using Geo = ragb::WtGeom<AK, NW, NB, KS, NS>;
    auto fn = &ragb::gemm_nt_wt_kernel<AK, NW, NB, KS, NS, OCC, 0>;
    int rc = ensure_lds(reinterpret_cast<const void*>(fn), Geo::LDS);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(ragb::xcd_grid(g.M, g.N, Geo::TM, Geo::TN)), dim3(Geo::THREADS), Geo::LDS, st, g);
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// GEMM over fp16 activations in fragment order (gemm_wt.hip.h): 256 x 128 tiles; one 16-deep K-step per stage and
// three workgroups per CU, or two steps per stage and two workgroups for long K (scripts/exp/gemm_wh_bench.hip).
int launch_gemm_t16(const _Float16* A, int lda, const _Float16* Wimg, const float* bias, const _Float16* R, int ldr,
                    _Float16* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
    const ragb::GemmWtParams g{A, Wimg, bias, R, C, M, N, K, lda, ldr, ldc, act, nullptr};
    if (K >= 1536 && K % 32 == 0) return launch_wt<1, 8, 4, 2, 3, 2>(g, st);
    return launch_wt<1, 8, 4, 1, 4, 3>(g, st);
}

// The feed-forward block of a layer as one kernel (ffn_fused_t16.hip.h): hidden 384, GELU (erf), I % 128 == 0.
int launch_ffn_fused_t16(const _Float16* X, const _Float16* W1img, const float* b1, const _Float16* W2pimg, const float* b2,
                         _Float16* Y, int M, int I, hipStream_t st) {
    using Geo = ragb::FfnFusedGeom<12>;
    auto fn = &ragb::ffn_fused_t16_kernel<12>;
    const int lds = Geo::lds(I);
    int rc = ensure_lds(reinterpret_cast<const void*>(fn), lds);
    if (rc) return rc;
    const ragb::FfnFusedParams g{X, W1img, b1, W2pimg, b2, Y, M, I};
    const int nrb = (M + 31) / 32;
    hipLaunchKernelGGL(fn, dim3((unsigned)((nrb + 3) / 4)), dim3(256), lds, st, g);
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// Small batches on the two-plane path: 64 x 64 tiles while they fit the chip in one round, 64 x 128 beyond that.
int launch_ws(ragb::GemmWsParams g, int splits, int n_cus, hipStream_t st, bool background) {
    const long long mt = (g.M + 63) / 64;
    if (background) {   // kernels that fit beside another stream's resident workgroups: 32 KiB of LDS, four waves
        g.splits = splits;
        using Bg = ragb::WsGeom<1, 1>;
        auto fn = &ragb::gemm_nt_ws_kernel<1, 1>;
        int rc0 = ensure_lds(reinterpret_cast<const void*>(fn), Bg::LDS);
        if (rc0) return rc0;
        hipLaunchKernelGGL(fn, dim3((unsigned)ragb::ws_grid(g.M, g.N, 1, splits)), dim3(Bg::THREADS), Bg::LDS, st, g);
        RAGC_HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
    const bool wide = mt * ((g.N + 63) / 64) * splits > n_cus;
    g.splits = splits;
    const dim3 grid((unsigned)ragb::ws_grid(g.M, g.N, wide ? 2 : 1, splits), 1, 1);
    int rc;
    if (wide) {
        auto fn = &ragb::gemm_nt_ws_kernel<2>;
        if ((rc = ensure_lds(reinterpret_cast<const void*>(fn), ragb::WsGeom<2>::LDS))) return rc;
        hipLaunchKernelGGL(fn, grid, dim3(ragb::WsGeom<2>::THREADS), ragb::WsGeom<2>::LDS, st, g);
    } else {
        auto fn = &ragb::gemm_nt_ws_kernel<1>;
        if ((rc = ensure_lds(reinterpret_cast<const void*>(fn), ragb::WsGeom<1>::LDS))) return rc;
        hipLaunchKernelGGL(fn, grid, dim3(ragb::WsGeom<1>::THREADS), ragb::WsGeom<1>::LDS, st, g);
    }
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// Plain GEMM with fused epilogue.  Small M uses 64x64 tiles so the grid still covers the chip; big M
// runs on the bf16 matrix cores with fp32 accuracy (split-bf16 image), or takes fp16 inputs when the
// model was created with RAG_GEMM_F16, or stays on the fp32 MFMA (RAG_GEMM_F32_STRICT).
int launch_gemm(const float* A, int lda, const WRef& Wr, int ldw, const float* bias, const float* R,
                int ldr, float* C, int ldc, int M, int N, int K, int act, hipStream_t st, int n_cus = 256) {
    const float* W = Wr.w;
    if (M <= 0) return RAG_OK;
    // big batches: both operands through LDS-DMA (gemm_wl.hip.h); the variants per shape are the faster ones of
    // scripts/exp/gemm_wl_bench.hip on the cross-encoder's shapes
    const bool wl_ok = M > 1024 && ldw == K && N % 32 == 0 && (lda % 4) == 0;
    if (wl_ok && Wr.wx2 && K % 16 == 0)   // fp32 results: two fp16 planes per operand, three products
        return launch_wl2(ragb::GemmWlParams{A, Wr.wx2, bias, R, C, M, N, K, lda, ldr, ldc, act, Wr.range_flag}, st, Wr.w6);
    if (wl_ok && Wr.wx && K % 16 == 0) {  // fp32 results, fp32 range: three bf16 planes, six products
        const ragb::GemmWlParams g{A, Wr.wx, bias, R, C, M, N, K, lda, ldr, ldc, act, nullptr};
        return N <= 512 ? launch_wl<0, 1, 3, true>(g, st) : launch_wl<0, 1, 3, false>(g, st);
    }
    if (wl_ok && Wr.wxf && K % 32 == 0) {  // fp16 inputs (the reference's GPU reranker precision)
        const ragb::GemmWlParams g{A, Wr.wxf, bias, R, C, M, N, K, lda, ldr, ldc, act, nullptr};
        return K <= 512 ? launch_wl<1, 1, 4, true>(g, st) : launch_wl<1, 2, 3, false>(g, st);
    }
    if (M <= 1024 && Wr.wx2 && ldw == K && K % 64 == 0 && N % 32 == 0 && (lda % 4) == 0)
        return launch_ws(ragb::GemmWsParams{A, Wr.wx2, bias, R, C, M, N, K, lda, ldr, ldc, act, K, 1, Wr.range_flag}, 1, n_cus, st, Wr.background);
    ragb::GemmParams g;
    g.A = A; g.W = W; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.act = act;
    g.k_per_split = K;
    if (M > 1024) {
        dim3 grid(ragb::xcd_grid(M, N, 128, 128), 1, 1);
        ragb::gemm_nt_kernel<2, 2><<<grid, dim3(256), 0, st>>>(g);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64, 1);
        // more 64 x 64 tiles than CUs but no more 64 x 96 tiles than CUs: one round of bigger tiles instead of
        // a full round plus a mostly empty one (bge-base FFN input projection at 450 tokens: 336 -> 224 tiles)
        const long long t64 = (long long)grid.x * grid.y, t96 = (long long)((N + 95) / 96) * grid.y;
        if (K % ragb::SBK == 0 && t64 > n_cus && t96 <= n_cus) {
            grid.x = (N + 95) / 96;
            ragb::gemm_nt_small_kernel<3><<<grid, dim3(768), 0, st>>>(g);
        } else if (K % ragb::SBK == 0) {
            ragb::gemm_nt_small_kernel<2><<<grid, dim3(512), 0, st>>>(g);
        } else {
            ragb::gemm_nt_kernel<1, 1><<<grid, dim3(256), 0, st>>>(g);
        }
    }
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// y = LayerNorm(A Wᵀ + bias + R).  Small M: split-K GEMM into partial slabs (`part`, room for
// kMaxSplits * M * N floats) reduced by the LayerNorm kernel; big M: fused-epilogue GEMM into `part`,
// then LayerNorm.
constexpr int kMaxSplits = ragb::kMaxSplitK;
int launch_gemm_ln(const float* A, int lda, const WRef& Wr, int ldw, const float* bias, const float* R, float* part,
                   const float* ln_g, const float* ln_b, float* y, int M, int N, int K, float eps, int n_cus,
                   hipStream_t st) {
    if (M <= 0) return RAG_OK;
    const float* W = Wr.w;
    int splits = 1;
    if (M <= 1024 && Wr.wx2 && K % 64 == 0 && N % 32 == 0) {
        // two-plane path: 64 x 64 tiles, one workgroup per CU; split K while that still adds workgroups to an
        // under-filled chip (a split's K range stays a multiple of 64 and at least 128)
        const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
        int sp = 1;
        for (int c = 2; c <= kMaxSplits; ++c)
            if (K % (64 * c) == 0 && K / c >= 128 && tiles * c <= n_cus) sp = c;
        if (sp == 1) {
            int rc = launch_ws(ragb::GemmWsParams{A, Wr.wx2, bias, R, part, M, N, K, lda, N, N, ragb::ACT_NONE, K, 1, Wr.range_flag},
                               1, n_cus, st, Wr.background);
            if (rc) return rc;
            ragb::splitk_bias_res_ln_kernel<<<dim3((M + 3) / 4), dim3(256), 0, st>>>(part, 1, nullptr, nullptr, ln_g, ln_b, y, M, N, eps);
        } else {
            int rc = launch_ws(ragb::GemmWsParams{A, Wr.wx2, nullptr, nullptr, part, M, N, K, lda, 0, N, ragb::ACT_NONE, K / sp, sp,
                                                  Wr.range_flag}, sp, n_cus, st, Wr.background);
            if (rc) return rc;
            ragb::splitk_bias_res_ln_kernel<<<dim3((M + 3) / 4), dim3(256), 0, st>>>(part, sp, bias, R, ln_g, ln_b, y, M, N, eps);
        }
        RAGC_HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
    if (M <= 1024) {
        // One wave multiplies one 32x32 output tile over its K range, and the chip has 4 * n_cus SIMDs:
        // pick the split that minimises (rounds of wave-tiles) x (K per split), e.g. 336 tiles at
        // K = 3072 are one round of K = 1024 with 3 splits but two rounds of K = 768 with 4.
        const long long wave_tiles = (long long)((N + 31) / 32) * ((M + 31) / 32);
        const long long simds = 4LL * n_cus;
        long long best = -1;
        for (int sp = 1; sp <= kMaxSplits; ++sp) {
            if (K % (ragb::SBK * sp)) continue;
            const long long rounds = (wave_tiles * sp + simds - 1) / simds;
            const long long cost = rounds * (K / sp) + 16 * sp;  // + a little per slab for the reduce pass
            if (best < 0 || cost < best) {
                best = cost;
                splits = sp;
            }
        }
    }
    if (splits == 1) {
        int rc = launch_gemm(A, lda, Wr, ldw, bias, R, N, part, N, M, N, K, ragb::ACT_NONE, st);
        if (rc) return rc;
        ragb::splitk_bias_res_ln_kernel<<<dim3((M + 3) / 4), dim3(256), 0, st>>>(part, 1, nullptr, nullptr, ln_g, ln_b, y, M, N, eps);
    } else {
        ragb::GemmParams g;
        g.A = A; g.W = W; g.bias = nullptr; g.R = nullptr; g.C = part;
        g.M = M; g.N = N; g.K = K;
        g.lda = lda; g.ldw = ldw; g.ldr = 0; g.ldc = N;
        g.act = ragb::ACT_NONE;
        const bool small8 = K % ragb::SBK == 0;
        const int step = small8 ? ragb::SBK : ragb::GBK;
        const int kt = K / step;
        g.k_per_split = (kt + splits - 1) / splits * step;
        splits = (K + g.k_per_split - 1) / g.k_per_split;
        dim3 grid((N + 63) / 64, (M + 63) / 64, splits);
        if (small8)
            ragb::gemm_nt_small_kernel<2><<<grid, dim3(512), 0, st>>>(g);
        else
            ragb::gemm_nt_kernel<1, 1><<<grid, dim3(256), 0, st>>>(g);
        RAGC_HIP_TRY(hipGetLastError());
        ragb::splitk_bias_res_ln_kernel<<<dim3((M + 3) / 4), dim3(256), 0, st>>>(part, splits, bias, R, ln_g, ln_b, y, M, N, eps);
    }
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// ---- LayerNorm folded into its consumers (gemm_wl.hip.h): the two GEMM forms of the query-encoder path ----------------
// Form A: C = act(LN(y) W^T + b) computed as act(rstd (y W'^T - mean s) + b2) — A = y raw, image W' = W gamma.
int launch_ws_lnf_a(rag_bert* h, const float* Y, const float2* ts_in, const _Float16* img, const float* s_vec, const float* b2,
                    float* C, int M, int N, int K, int act, uint32_t* range_flag, hipStream_t st) {
    ragb::GemmWsParams g{Y, img, b2, nullptr, C, M, N, K, K, 0, N, act, K, 1, range_flag};
    g.ts_in = ts_in;
    g.nblk_in = K / 32;
    g.ln_eps = h->cfg.ln_eps;
    g.fold_s = s_vec;
    return launch_ws(g, 1, h->n_cus, st, h->background);
}

// Form B: Yout = A W^T + b + LN(Ry), split-K finished by the last split of a tile to arrive; leaves Yout's block statistics.
constexpr int kLnfMaxTiles = 4096;
int launch_ws_lnf_b(rag_bert* h, const float* A, int K, const _Float16* img, const float* bias, const float* Ry,
                    const float2* ts_in, const float* ln_g, const float* ln_b, float* Yout, float2* ts_out, int M, int N,
                    uint32_t* range_flag, hipStream_t st) {
    // split K while that still adds workgroups to an under-filled chip (as launch_gemm_ln)
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
    int sp = 1;
    for (int c = 2; c <= kMaxSplits; ++c)
        if (K % (64 * c) == 0 && K / c >= 128 && tiles * c <= h->n_cus) sp = c;
    if (tiles > kLnfMaxTiles) return ragc_fail(RAG_ERR_UNSUPPORTED, "too many output tiles for the arrival counters");
    ragb::GemmWsParams g{A, img, bias, nullptr, h->y, M, N, K, K, N, N, ragb::ACT_NONE, K / sp, sp, range_flag};
    g.ts_in = ts_in;
    g.nblk_in = N / 32;      // the residual's LayerNorm is as wide as this GEMM's output (N == H)
    g.ln_eps = h->cfg.ln_eps;
    g.Ry = Ry;
    g.ln_g = ln_g;
    g.ln_b = ln_b;
    g.Yout = Yout;
    g.ts_out = ts_out;
    g.tile_ctr = h->tile_ctr;
    return launch_ws(g, sp, h->n_cus, st, h->background);
}

// Big-batch forms (gemm_nt_wl_kernel<2, 1, 3> with wl_epilogue_lnf): statistics per (row, 128-column tile).
int launch_wl_lnf_a(rag_bert* h, const float* Y, const float2* rs_in, const _Float16* img, const float* s_vec, const float* b2,
                    float* C, int M, int N, int K, int act, uint32_t* range_flag, hipStream_t st) {
    ragb::GemmWlParams g{Y, img, b2, nullptr, C, M, N, K, K, 0, N, act, range_flag};
    g.rs_in = rs_in;
    g.fold_s = s_vec;
    return launch_wl2(g, st, h->use_w6);
}

int launch_wl_lnf_b(rag_bert* h, const float* A, int K, const _Float16* img, const float* bias, const float* R_plain,
                    const float* Ry, const float2* rs_in, const float* ln_g, const float* ln_b, float* Yout, float2* ts_out,
                    float2* rs_out, int M, int N, uint32_t* range_flag, hipStream_t st) {
    ragb::GemmWlParams g{A, img, bias, R_plain, Yout, M, N, K, K, N, N, ragb::ACT_NONE, range_flag};
    g.rs_in = Ry ? rs_in : nullptr;
    g.Ry = Ry;
    g.ln_g = ln_g;
    g.ln_b = ln_b;
    g.ts_out = ts_out;
    int rc = launch_wl2(g, st, h->use_w6);
    if (rc) return rc;
    // the rows' (mean, rstd) from their tile statistics, for the kernels that read them next
    const int bw = lnf_big_blkw(N, h->use_w6);
    ragb::lnf_finalize_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st>>>(ts_out, N / bw, (float)bw, h->cfg.ln_eps, rs_out, M);
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

size_t out_elems(const rag_bert_config& c, int out_kind, long long nseq, long long tokens) {
    switch (out_kind) {
        case RAG_BERT_OUT_MEAN:
        case RAG_BERT_OUT_CLS: return (size_t)nseq * c.hidden;
        case RAG_BERT_OUT_LOGITS:
        case RAG_BERT_OUT_PROBS: return (size_t)nseq * std::max(1, c.n_labels);
        case RAG_BERT_OUT_HIDDEN: return (size_t)tokens * c.hidden;
        default: return 0;
    }
}

int forward_locked(rag_bert* h, const int* ids, const int* types, const int* cu, int nseq, int T, int max_len,
                   int out_kind, int normalize, float* out, hipStream_t st, uint32_t* range_flag) {
    using namespace ragb;
    const rag_bert_config& c = h->cfg;
    const int H = c.hidden, I = c.intermediate, heads = c.n_heads, dh = H / heads;
    if (!h->in_capture && h->ws_used && h->ws_stream != st) RAGC_HIP_TRY(hipStreamWaitEvent(st, h->ws_event, 0));
    struct Mark {  // record the workspace hand-over point on every exit path
        rag_bert* h;
        hipStream_t st;
        ~Mark() {
            if (h->in_capture) return;   // (the graph's launcher does the event traffic around the replay)
            if (h->ws_event && hipEventRecord(h->ws_event, st) == hipSuccess) {
                h->ws_stream = st;
                h->ws_used = true;
            }
        }
    } mark{h, st};
    if (h->in_capture && (T > h->ws_tokens || nseq > h->ws_seqs))
        return ragc_fail(RAG_ERR_STATE, "workspace too small inside a graph capture");
    if (T > h->ws_tokens || nseq > h->ws_seqs) {
        // the workspace is about to be reallocated: nothing enqueued earlier (on any stream) may still use it
        if (h->ws_used) RAGC_HIP_TRY(hipEventSynchronize(h->ws_event));
        RAGC_HIP_TRY(hipStreamSynchronize(st));
    }
    int rc = ensure_ws(h, T, nseq);
    if (rc) return rc;
    const float* const* w = h->w.data();

    EmbedParams ep;
    ep.ids = ids; ep.type_ids = c.type_vocab > 0 ? types : nullptr; ep.cu = cu;
    ep.word_emb = w[0]; ep.pos_emb = w[1]; ep.type_emb = c.type_vocab > 0 ? w[2] : nullptr;
    ep.ln_g = w[3]; ep.ln_b = w[4];
    ep.out = h->x;
    ep.T = T; ep.nseq = nseq; ep.H = H; ep.pos_offset = c.pos_offset; ep.max_pos = c.max_positions;
    ep.vocab = c.vocab_size; ep.type_vocab = c.type_vocab > 0 ? c.type_vocab : 1; ep.eps = c.ln_eps;

    const int act = map_act(c.act);
    const float scale = 1.0f / sqrtf((float)dh);
    const dim3 agrid((max_len + 63) / 64, heads, nseq);
    const dim3 mgrid((max_len + 31) / 32, heads, nseq);
    // Outputs that read first tokens only (CLS pooling, classifier head): after the last layer's attention nothing
    // but each sequence's first row is ever looked at, so that layer's output projection, LayerNorms and
    // feed-forward run on nseq rows instead of T — 9 of the 12 H^2 GEMM flops per token of a layer; one layer of
    // six is 12 % of a MiniLM cross-encoder pass (32 x 100 pairs: 3200 rows instead of 178 000).  Same arithmetic
    // per row, but not the same bits as a full pass: the GEMM kernel and its split-K count are picked from the row
    // count (launch_gemm / launch_gemm_ln: split-bf16 above 1024 rows, fp32 MFMA at or below; split-K from the
    // number of wave tiles), so with nseq rows instead of T the last layer's sums can be taken in another order.
    // The two agree to fp32 rounding (tests/test_bert_gpu.py::test_first_token_last_layer_matches_the_full_pass
    // holds them within 1e-5 on both sides of the 1024-row switch).
    const bool first_only_out = out_kind == RAG_BERT_OUT_CLS || out_kind == RAG_BERT_OUT_LOGITS || out_kind == RAG_BERT_OUT_PROBS;
    bool compact = false;  // h->pooled holds the final hidden state of the first tokens, one row per sequence
    // RAG_GEMM_F16, big batches: activations are fp16 in MFMA-fragment order from the embeddings to the last layer
    // (gemm_wt.hip.h, bert_tiled.hip.h); the first-token tail of the last layer and the heads continue in row-major fp32.
    const bool tiled = c.gemm_mode == RAG_GEMM_F16 && T > 1024 && !h->wxf.empty() && !h->valu_attention && !h->row_major_only;
    using half_t = _Float16;
    half_t* xh = reinterpret_cast<half_t*>(h->x);
    half_t* parth = reinterpret_cast<half_t*>(h->y);
    half_t* qkvh = reinterpret_cast<half_t*>(h->qkv);
    half_t* ctxh = reinterpret_cast<half_t*>(h->ctx);
    half_t* ffnh = reinterpret_cast<half_t*>(h->ffn);
    const int nrb = (T + 31) / 32;
    const int row_lds = 32 * (H + ragb::kRowPad) * (int)sizeof(float);
    auto ln_tiled = [&](const half_t* src, const float* g, const float* b, half_t* dst) {   // steps per wave: H / 64
        if (H <= 384) hipLaunchKernelGGL((ln_tiled_kernel<half_t, 6>), dim3(nrb), dim3(256), 0, st, src, g, b, dst, T, H, c.ln_eps);
        else if (H <= 768) hipLaunchKernelGGL((ln_tiled_kernel<half_t, 12>), dim3(nrb), dim3(256), 0, st, src, g, b, dst, T, H, c.ln_eps);
        else hipLaunchKernelGGL((ln_tiled_kernel<half_t, 16>), dim3(nrb), dim3(256), 0, st, src, g, b, dst, T, H, c.ln_eps);
    };
    // The query-encoder path (<= 1024 tokens, two-plane images valid): LayerNorms folded into their consumers — five
    // launches per layer instead of seven (gemm_wl.hip.h).  x holds the layer's PRE-LayerNorm input, ts_a its statistics.
    const bool lnf = h->lnf_ok && T <= 1024 && T <= h->ws_tokens && !h->force_x6 && !h->valu_attention && !tiled &&
                     h->weights_fit_f16 && h->ts_a != nullptr;
    const float* xfinal = h->x;   // what the output stage reads as the last hidden state
    if (tiled) {
        if (H <= 384) hipLaunchKernelGGL((embed_ln_tiled_kernel<half_t, 6>), dim3(nrb), dim3(256), 0, st, ep, xh);
        else if (H <= 768) hipLaunchKernelGGL((embed_ln_tiled_kernel<half_t, 12>), dim3(nrb), dim3(256), 0, st, ep, xh);
        else hipLaunchKernelGGL((embed_ln_tiled_kernel<half_t, 16>), dim3(nrb), dim3(256), 0, st, ep, xh);
    } else if (lnf) {
        embed_pre_kernel<<<dim3((T + 3) / 4), dim3(256), 0, st>>>(ep, h->ts_a);
    } else {
        embed_ln_kernel<<<dim3((T + 3) / 4), dim3(256), 0, st>>>(ep);
    }
    RAGC_HIP_TRY(hipGetLastError());

    for (int l = 0; lnf && l < c.n_layers; ++l) {
        const float* const* lw = w + kEmbEntries + kPerLayer * l;
        const float* g_prev = l == 0 ? w[3] : (lw - kPerLayer)[10];   // the LayerNorm pending on x
        const float* b_prev = l == 0 ? w[4] : (lw - kPerLayer)[11];
        const size_t at = (size_t)4 * l;
        const bool last_first_only = first_only_out && l == c.n_layers - 1 && nseq < T;
        rc = launch_ws_lnf_a(h, h->x, h->ts_a, h->wfold[2 * l], h->fold_s[2 * l], h->fold_b[2 * l], h->qkv, T, 3 * H, H, ACT_NONE,
                             range_flag, st);
        if (rc) return rc;
        if (last_first_only) {
            // the rest of this layer runs on the sequences' first rows only, with materialised LayerNorms (nseq rows)
            const dim3 fgrid(1, heads, nseq);
            if (dh == 32)
                attention_mfma_kernel<32><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            else
                attention_mfma_kernel<64><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            RAGC_HIP_TRY(hipGetLastError());
            gather_rows_ln_kernel<<<dim3((nseq + 3) / 4), dim3(256), 0, st>>>(h->x, h->ts_a, g_prev, b_prev, cu, h->pooled, nseq, H, c.ln_eps, 32);
            RAGC_HIP_TRY(hipGetLastError());
            const WRef w1{lw[2], h->wx2[at + 1], nullptr, nullptr, range_flag, h->background};
            const WRef w2{lw[6], h->wx2[at + 2], nullptr, nullptr, range_flag, h->background};
            const WRef w3{lw[8], h->wx2[at + 3], nullptr, nullptr, range_flag, h->background};
            rc = launch_gemm_ln(h->ctx, H, w1, H, lw[3], h->pooled, h->y, lw[4], lw[5], h->pooled, nseq, H, H, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            rc = launch_gemm(h->pooled, H, w2, H, lw[7], nullptr, 0, h->ffn, I, nseq, I, H, act, st, h->n_cus);
            if (rc) return rc;
            rc = launch_gemm_ln(h->ffn, I, w3, I, lw[9], h->pooled, h->y, lw[10], lw[11], h->pooled, nseq, H, I, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            compact = true;
            break;
        }
        if (dh == 32)
            attention_mfma_kernel<32><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        else
            attention_mfma_kernel<64><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        RAGC_HIP_TRY(hipGetLastError());
        // y1 = ctx Wo^T + bo + LN_prev(x): finished by the last K-split of each tile; statistics to ts_b
        rc = launch_ws_lnf_b(h, h->ctx, H, h->wx2[at + 1], lw[3], h->x, h->ts_a, g_prev, b_prev, h->y1, h->ts_b, T, H, range_flag, st);
        if (rc) return rc;
        rc = launch_ws_lnf_a(h, h->y1, h->ts_b, h->wfold[2 * l + 1], h->fold_s[2 * l + 1], h->fold_b[2 * l + 1], h->ffn, T, I, H, act,
                             range_flag, st);
        if (rc) return rc;
        // x = ffn W2^T + b2 + LN1(y1): the next layer's pre-LayerNorm input; statistics to ts_a
        rc = launch_ws_lnf_b(h, h->ffn, I, h->wx2[at + 3], lw[9], h->y1, h->ts_b, lw[4], lw[5], h->x, h->ts_a, T, H, range_flag, st);
        if (rc) return rc;
        if (l == c.n_layers - 1) {   // the encoder's last LayerNorm, materialised once for the output stage
            ln_from_tiles_kernel<<<dim3((T + 3) / 4), dim3(256), 0, st>>>(h->x, h->ts_a, lw[10], lw[11], h->y1, T, H, c.ln_eps, 32);
            RAGC_HIP_TRY(hipGetLastError());
            xfinal = h->y1;
        }
    }

    // Big batches in the default mode (> 1024 tokens, two-plane GEMMs): the same folding without any hand-off — these GEMMs
    // are not split along K, so the epilogue of the GEMM that writes a pre-LayerNorm sum leaves its rows' statistics per
    // 128-column tile and the GEMMs that read it apply them (wl_epilogue_lnf).  Ten of a six-layer cross-encoder's twelve
    // LayerNorm passes (0.55 GB read + written each at 178 k tokens) disappear; layer 0 reads the materialised embeddings.
    const bool lnfb = h->lnf_big && h->lnf_built && T > 1024 && !tiled && !h->force_x6 && h->weights_fit_f16 &&
                      !h->valu_attention && c.gemm_mode == RAG_GEMM_F32 && H % 128 == 0 && (H % 16) == 0 && (I % 16) == 0 &&
                      h->ts_a != nullptr;
    for (int l = 0; lnfb && l < c.n_layers; ++l) {
        const float* const* lw = w + kEmbEntries + kPerLayer * l;
        const bool pending = l > 0;   // x holds a pre-LayerNorm sum (statistics in ts_a) instead of a finished row
        const float* g_prev = pending ? (lw - kPerLayer)[10] : nullptr;
        const float* b_prev = pending ? (lw - kPerLayer)[11] : nullptr;
        const size_t at = (size_t)4 * l;
        const WRef wq{lw[0], h->wx2[at + 0], nullptr, nullptr, range_flag, h->background, h->use_w6};
        const bool last_first_only = first_only_out && l == c.n_layers - 1 && nseq < T;
        if (pending)
            rc = launch_wl_lnf_a(h, h->x, h->rs_a, h->wfold[2 * l], h->fold_s[2 * l], h->fold_b[2 * l], h->qkv, T, 3 * H, H, ACT_NONE,
                                 range_flag, st);
        else
            rc = launch_gemm(h->x, H, wq, H, lw[1], nullptr, 0, h->qkv, 3 * H, T, 3 * H, H, ACT_NONE, st, h->n_cus);
        if (rc) return rc;
        if (last_first_only) {
            const dim3 fgrid(1, heads, nseq);
            if (dh == 32)
                attention_mfma_kernel<32><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            else
                attention_mfma_kernel<64><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            RAGC_HIP_TRY(hipGetLastError());
            if (pending) {
                gather_rows_ln_kernel<<<dim3((nseq + 3) / 4), dim3(256), 0, st>>>(h->x, h->ts_a, g_prev, b_prev, cu, h->pooled, nseq, H, c.ln_eps, lnf_big_blkw(H, h->use_w6));
            } else {
                const int total = nseq * H;
                gather_rows_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(h->x, cu, h->pooled, nseq, H);
            }
            RAGC_HIP_TRY(hipGetLastError());
            const WRef w1{lw[2], h->wx2[at + 1], nullptr, nullptr, range_flag, h->background};
            const WRef w2{lw[6], h->wx2[at + 2], nullptr, nullptr, range_flag, h->background};
            const WRef w3{lw[8], h->wx2[at + 3], nullptr, nullptr, range_flag, h->background};
            rc = launch_gemm_ln(h->ctx, H, w1, H, lw[3], h->pooled, h->y, lw[4], lw[5], h->pooled, nseq, H, H, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            rc = launch_gemm(h->pooled, H, w2, H, lw[7], nullptr, 0, h->ffn, I, nseq, I, H, act, st, h->n_cus);
            if (rc) return rc;
            rc = launch_gemm_ln(h->ffn, I, w3, I, lw[9], h->pooled, h->y, lw[10], lw[11], h->pooled, nseq, H, I, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            compact = true;
            break;
        }
        if (dh == 32)
            attention_mfma_kernel<32><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        else
            attention_mfma_kernel<64><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        RAGC_HIP_TRY(hipGetLastError());
        // y1 = ctx Wo^T + bo + (x | LN_prev(x)) into h->y, statistics to ts_b
        rc = launch_wl_lnf_b(h, h->ctx, H, h->wx2[at + 1], lw[3], pending ? nullptr : h->x, pending ? h->x : nullptr, h->rs_a, g_prev,
                             b_prev, h->y, h->ts_b, h->rs_b, T, H, range_flag, st);
        if (rc) return rc;
        rc = launch_wl_lnf_a(h, h->y, h->rs_b, h->wfold[2 * l + 1], h->fold_s[2 * l + 1], h->fold_b[2 * l + 1], h->ffn, T, I, H, act,
                             range_flag, st);
        if (rc) return rc;
        // x = ffn W2^T + b2 + LN1(y1): the next layer's pre-LayerNorm input, statistics to ts_a / rs_a
        rc = launch_wl_lnf_b(h, h->ffn, I, h->wx2[at + 3], lw[9], nullptr, h->y, h->rs_b, lw[4], lw[5], h->x, h->ts_a, h->rs_a, T, H,
                             range_flag, st);
        if (rc) return rc;
        if (l == c.n_layers - 1) {   // the last LayerNorm, materialised once for the output stage
            ln_from_tiles_kernel<<<dim3((T + 3) / 4), dim3(256), 0, st>>>(h->x, h->ts_a, lw[10], lw[11], h->y, T, H, c.ln_eps, lnf_big_blkw(H, h->use_w6));
            RAGC_HIP_TRY(hipGetLastError());
            xfinal = h->y;
        }
    }

    for (int l = 0; !lnf && !lnfb && l < c.n_layers; ++l) {
        const float* const* lw = w + kEmbEntries + kPerLayer * l;
        const int wsrc[4] = {0, 2, 6, 8};  // qkv_w, attn_out_w, ffn_in_w, ffn_out_w in the layer's table
        auto wref = [&](int i) -> WRef {
            const size_t at = (size_t)4 * l + i;
            const bool two_plane = !h->force_x6 && h->weights_fit_f16 && !h->wx2.empty();
            return WRef{lw[wsrc[i]], two_plane ? h->wx2[at] : nullptr, h->wx.empty() ? nullptr : h->wx[at],
                        h->wxf.empty() ? nullptr : h->wxf[at], range_flag, h->background, h->use_w6};
        };
        const bool last_first_only = first_only_out && l == c.n_layers - 1 && !h->valu_attention && nseq < T;
        if (tiled) {
            const size_t at = (size_t)4 * l;
            rc = launch_gemm_t16(xh, H, h->wxf[at + 0], lw[1], nullptr, 0, qkvh, 3 * H, T, 3 * H, H, ACT_NONE, st);
            if (rc) return rc;
            if (last_first_only) {
                // first tokens only: attention writes row-major fp32 rows, and the rest of the layer is the row-major tail
                const dim3 fgrid(1, heads, nseq);
                if (h->tiled_attention_f32) {
                    if (dh == 32)
                        attention_tiled_kernel<32, half_t><<<fgrid, dim3(64), 0, st>>>(qkvh, cu, nullptr, h->ctx, H, heads, scale, 1);
                    else
                        attention_tiled_kernel<64, half_t><<<fgrid, dim3(64), 0, st>>>(qkvh, cu, nullptr, h->ctx, H, heads, scale, 1);
                } else if (dh == 32) {
                    attention_t16_kernel<32, 1><<<fgrid, dim3(64), 0, st>>>(qkvh, cu, nullptr, h->ctx, H, heads, scale, 1);
                } else {
                    attention_t16_kernel<64, 1><<<fgrid, dim3(64), 0, st>>>(qkvh, cu, nullptr, h->ctx, H, heads, scale, 1);
                }
                RAGC_HIP_TRY(hipGetLastError());
                const int total = nseq * H;
                gather_rows_tiled_kernel<half_t><<<dim3((total + 255) / 256), dim3(256), 0, st>>>(xh, cu, h->pooled, nseq, H);
                RAGC_HIP_TRY(hipGetLastError());
                rc = launch_gemm_ln(h->ctx, H, wref(1), H, lw[3], h->pooled, h->y, lw[4], lw[5], h->pooled, nseq, H, H, c.ln_eps, h->n_cus, st);
                if (rc) return rc;
                rc = launch_gemm(h->pooled, H, wref(2), H, lw[7], nullptr, 0, h->ffn, I, nseq, I, H, act, st, h->n_cus);
                if (rc) return rc;
                rc = launch_gemm_ln(h->ffn, I, wref(3), I, lw[9], h->pooled, h->y, lw[10], lw[11], h->pooled, nseq, H, I, c.ln_eps, h->n_cus, st);
                if (rc) return rc;
                compact = true;
                break;
            }
            if (h->tiled_attention_f32) {
                if (dh == 32)
                    attention_tiled_kernel<32, half_t><<<mgrid, dim3(64), 0, st>>>(qkvh, cu, ctxh, nullptr, H, heads, scale, 0);
                else
                    attention_tiled_kernel<64, half_t><<<mgrid, dim3(64), 0, st>>>(qkvh, cu, ctxh, nullptr, H, heads, scale, 0);
            } else if (dh == 32) {   // (two query blocks per wave — K and V read once per 64 queries — measured 5 % slower)
                attention_t16_kernel<32, 1><<<mgrid, dim3(64), 0, st>>>(qkvh, cu, ctxh, nullptr, H, heads, scale, 0);
            } else {
                attention_t16_kernel<64, 1><<<mgrid, dim3(64), 0, st>>>(qkvh, cu, ctxh, nullptr, H, heads, scale, 0);
            }
            RAGC_HIP_TRY(hipGetLastError());
            // attention output projection + bias + residual (fp16, in the epilogue), LayerNorm back into x
            rc = launch_gemm_t16(ctxh, H, h->wxf[at + 1], lw[3], xh, H, parth, H, T, H, H, ACT_NONE, st);
            if (rc) return rc;
            ln_tiled(parth, lw[4], lw[5], xh);
            RAGC_HIP_TRY(hipGetLastError());
            if (!h->wffn2p.empty() && h->wffn2p[l]) {
                // FFN-in -> GELU -> FFN-out + residual in one kernel: the 4 x wider intermediate never reaches HBM
                rc = launch_ffn_fused_t16(xh, h->wxf[at + 2], lw[7], h->wffn2p[l], lw[9], parth, T, I, st);
                if (rc) return rc;
            } else {
                rc = launch_gemm_t16(xh, H, h->wxf[at + 2], lw[7], nullptr, 0, ffnh, I, T, I, H, act, st);
                if (rc) return rc;
                rc = launch_gemm_t16(ffnh, I, h->wxf[at + 3], lw[9], xh, H, parth, H, T, H, I, ACT_NONE, st);
                if (rc) return rc;
            }
            ln_tiled(parth, lw[10], lw[11], xh);
            RAGC_HIP_TRY(hipGetLastError());
            continue;
        }
        // QKV projection
        rc = launch_gemm(h->x, H, wref(0), H, lw[1], nullptr, 0, h->qkv, 3 * H, T, 3 * H, H, ACT_NONE, st, h->n_cus);
        if (rc) return rc;
        if (last_first_only) {
            const dim3 fgrid(1, heads, nseq);
            if (dh == 32)
                attention_mfma_kernel<32><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            else
                attention_mfma_kernel<64><<<fgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale, 1);
            RAGC_HIP_TRY(hipGetLastError());
            const int total = nseq * H;   // residual rows: the first tokens' x
            gather_rows_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(h->x, cu, h->pooled, nseq, H);
            RAGC_HIP_TRY(hipGetLastError());
            rc = launch_gemm_ln(h->ctx, H, wref(1), H, lw[3], h->pooled, h->y, lw[4], lw[5], h->pooled, nseq, H, H, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            rc = launch_gemm(h->pooled, H, wref(2), H, lw[7], nullptr, 0, h->ffn, I, nseq, I, H, act, st, h->n_cus);
            if (rc) return rc;
            rc = launch_gemm_ln(h->ffn, I, wref(3), I, lw[9], h->pooled, h->y, lw[10], lw[11], h->pooled, nseq, H, I, c.ln_eps, h->n_cus, st);
            if (rc) return rc;
            compact = true;
            break;
        }
        if (h->valu_attention) {
            if (dh == 32)
                attention_kernel<32><<<agrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
            else
                attention_kernel<64><<<agrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        } else {
            if (dh == 32)
                attention_mfma_kernel<32><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
            else
                attention_mfma_kernel<64><<<mgrid, dim3(64), 0, st>>>(h->qkv, cu, h->ctx, H, heads, scale);
        }
        RAGC_HIP_TRY(hipGetLastError());
        // attention output projection + residual + LayerNorm (x is both residual and destination:
        // each token's row is read and written by the same wave of the LayerNorm kernel)
        rc = launch_gemm_ln(h->ctx, H, wref(1), H, lw[3], h->x, h->y, lw[4], lw[5], h->x, T, H, H, c.ln_eps, h->n_cus, st);
        if (rc) return rc;
        // feed-forward: act(x W1ᵀ + b1) W2ᵀ + b2 + residual, LayerNorm
        rc = launch_gemm(h->x, H, wref(2), H, lw[7], nullptr, 0, h->ffn, I, T, I, H, act, st, h->n_cus);
        if (rc) return rc;
        rc = launch_gemm_ln(h->ffn, I, wref(3), I, lw[9], h->x, h->y, lw[10], lw[11], h->x, T, H, I, c.ln_eps, h->n_cus, st);
        if (rc) return rc;
    }

    switch (out_kind) {
        case RAG_BERT_OUT_HIDDEN:
            if (tiled) {
                if ((rc = ensure_lds(reinterpret_cast<const void*>(&untile_kernel<half_t>), row_lds))) return rc;
                hipLaunchKernelGGL(untile_kernel<half_t>, dim3(nrb), dim3(256), row_lds, st, xh, out, T, H);
                RAGC_HIP_TRY(hipGetLastError());
                break;
            }
            RAGC_HIP_TRY(hipMemcpyAsync(out, xfinal, (size_t)T * H * sizeof(float), hipMemcpyDeviceToDevice, st));
            break;
        case RAG_BERT_OUT_MEAN:
        case RAG_BERT_OUT_CLS:
            if (compact)
                pool_kernel<<<dim3(nseq), dim3(256), 0, st>>>(h->pooled, cu, out, H, 2, normalize);
            else if (tiled)
                pool_tiled_kernel<half_t><<<dim3(nseq), dim3(256), 0, st>>>(xh, cu, out, H, out_kind == RAG_BERT_OUT_MEAN ? 0 : 1, normalize);
            else
                pool_kernel<<<dim3(nseq), dim3(256), 0, st>>>(xfinal, cu, out, H, out_kind == RAG_BERT_OUT_MEAN ? 0 : 1, normalize);
            RAGC_HIP_TRY(hipGetLastError());
            break;
        case RAG_BERT_OUT_LOGITS:
        case RAG_BERT_OUT_PROBS: {
            const float* const* hw = w + kEmbEntries + kPerLayer * c.n_layers;
            if (!compact) {
                const int total = nseq * H;
                if (tiled)
                    gather_rows_tiled_kernel<half_t><<<dim3((total + 255) / 256), dim3(256), 0, st>>>(xh, cu, h->pooled, nseq, H);
                else
                    gather_rows_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(xfinal, cu, h->pooled, nseq, H);
                RAGC_HIP_TRY(hipGetLastError());
            }
            rc = launch_gemm(h->pooled, H, WRef{hw[0], nullptr, nullptr, nullptr, nullptr, false}, H, hw[1], nullptr, 0, h->pooled2, H, nseq, H, H, ACT_TANH, st);
            if (rc) return rc;
            const bool probs = out_kind == RAG_BERT_OUT_PROBS;
            head_out_kernel<<<dim3(nseq, c.n_labels), dim3(64), 0, st>>>(h->pooled2, hw[2], hw[3], probs ? h->logits : out,
                                                                        probs ? out : nullptr, H, c.n_labels);
            RAGC_HIP_TRY(hipGetLastError());
            break;
        }
        default: return ragc_fail(RAG_ERR_INVALID_ARG, "unknown out_kind %d", out_kind);
    }
    return RAG_OK;
}

int check_forward(const rag_bert* h, int nseq, int out_kind) {
    if (!h) return ragc_fail(RAG_ERR_INVALID_ARG, "null model handle");
    if (nseq <= 0) return ragc_fail(RAG_ERR_INVALID_ARG, "nseq=%d out of range", nseq);
    if (out_kind < RAG_BERT_OUT_MEAN || out_kind > RAG_BERT_OUT_HIDDEN)
        return ragc_fail(RAG_ERR_INVALID_ARG, "unknown out_kind %d", out_kind);
    if ((out_kind == RAG_BERT_OUT_LOGITS || out_kind == RAG_BERT_OUT_PROBS) && h->cfg.head == RAG_HEAD_NONE)
        return ragc_fail(RAG_ERR_STATE, "model was created without a classifier head");
    return RAG_OK;
}

}  // namespace

namespace {
// Build the fragment-order images of every layer's four GEMM weights: kind 0 three bf16 planes (wx), 1 one fp16
// plane (wxf), 2 two fp16 planes (wx2; weights outside fp16's range raise h->range_flag).  Synchronises h->stream.
int build_images(rag_bert* h, int kind) {
    const rag_bert_config& c = h->cfg;
    const int H = c.hidden, I = c.intermediate;
    const int shape[4][2] = {{3 * H, H}, {H, H}, {I, H}, {H, I}};
    const int wsrc[4] = {0, 2, 6, 8};
    const size_t n_img = (size_t)4 * c.n_layers;
    if (kind == 0 && !h->wx.empty()) return RAG_OK;
    if (kind == 0) h->wx.assign(n_img, nullptr);
    if (kind == 1) h->wxf.assign(n_img, nullptr);
    if (kind == 2) h->wx2.assign(n_img, nullptr);
    const size_t bytes_per_elem = kind == 0 ? 6 : (kind == 1 ? 2 : 4);
    for (int l = 0; l < c.n_layers; ++l) {
        for (int i = 0; i < 4; ++i) {
            const int N = shape[i][0], K = shape[i][1];
            void* img = nullptr;
            if (hipMalloc(&img, (size_t)N * K * bytes_per_elem) != hipSuccess)
                return ragc_fail(RAG_ERR_OOM, "device allocation of the fragment-order weight images failed");
            const size_t at = (size_t)4 * l + i;
            const float* W = h->w[kEmbEntries + kPerLayer * l + wsrc[i]];
            const dim3 grid((unsigned)(((long long)N * K + 255) / 256));
            if (kind == 0) {
                h->wx[at] = static_cast<__bf16*>(img);
                ragb::pack_x6_kernel<<<grid, dim3(256), 0, h->stream>>>(W, N, K, K, h->wx[at]);
            } else if (kind == 1) {
                h->wxf[at] = static_cast<_Float16*>(img);
                ragb::pack_f16_frag_kernel<<<grid, dim3(256), 0, h->stream>>>(W, N, K, K, h->wxf[at]);
            } else {
                h->wx2[at] = static_cast<_Float16*>(img);
                ragb::pack_f16x2_frag_kernel<<<grid, dim3(256), 0, h->stream>>>(W, N, K, K, h->wx2[at], h->range_pin);
            }
        }
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return ragc_fail(RAG_ERR_HIP, "building the fragment-order weight images failed");
    return RAG_OK;
}
}  // namespace


namespace {
// LayerNorm folded into its consumers (gemm_wl.hip.h): per layer, the QKV and feed-forward input weights with the
// preceding LayerNorm's gamma folded into their two-plane images, plus s[n] = sum_k W'[n][k] and bias2 = bias + W beta.
// Models whose widths the small-batch kernel does not take (K % 64, N % 32) keep the separate LayerNorm launches.
int build_folded(rag_bert* h) {
    const rag_bert_config& c = h->cfg;
    const int H = c.hidden, I = c.intermediate;
    if (H % 64 || I % 64 || h->wx2.empty()) return RAG_OK;
    const size_t n = (size_t)2 * c.n_layers;
    h->wfold.assign(n, nullptr);
    h->fold_s.assign(n, nullptr);
    h->fold_b.assign(n, nullptr);
    for (int l = 0; l < c.n_layers; ++l) {
        const float* const* lw = h->w.data() + kEmbEntries + kPerLayer * l;
        const float* g_prev = l == 0 ? h->w[3] : (lw - kPerLayer)[10];
        const float* b_prev = l == 0 ? h->w[4] : (lw - kPerLayer)[11];
        const struct { const float *W, *bias, *g, *b; int N; } jobs[2] = {{lw[0], lw[1], g_prev, b_prev, 3 * H},
                                                                         {lw[6], lw[7], lw[4], lw[5], I}};
        for (int j = 0; j < 2; ++j) {
            const size_t at = (size_t)2 * l + j;
            const int N = jobs[j].N, K = H;
            void* img = nullptr;
            float *sv = nullptr, *bv = nullptr;
            if (hipMalloc(&img, (size_t)N * K * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&sv), (size_t)N * sizeof(float)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&bv), (size_t)N * sizeof(float)) != hipSuccess) {
                if (img) (void)hipFree(img);
                if (sv) (void)hipFree(sv);
                return ragc_fail(RAG_ERR_OOM, "device allocation of the LayerNorm-folded weight images failed");
            }
            h->wfold[at] = static_cast<_Float16*>(img);
            h->fold_s[at] = sv;
            h->fold_b[at] = bv;
            ragb::pack_f16x2_frag_kernel<<<dim3((unsigned)(((long long)N * K + 255) / 256)), dim3(256), 0, h->stream>>>(
                jobs[j].W, N, K, K, h->wfold[at], h->range_pin, jobs[j].g);
            ragb::fold_prep_kernel<<<dim3((N + 3) / 4), dim3(256), 0, h->stream>>>(jobs[j].W, jobs[j].bias, jobs[j].g, jobs[j].b, N, K, sv, bv);
        }
    }
    if (hipMalloc(reinterpret_cast<void**>(&h->tile_ctr), kLnfMaxTiles * sizeof(unsigned)) != hipSuccess)
        return ragc_fail(RAG_ERR_OOM, "device allocation of the tile counters failed");
    if (hipMemsetAsync(h->tile_ctr, 0, kLnfMaxTiles * sizeof(unsigned), h->stream) != hipSuccess || hipGetLastError() != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
        return ragc_fail(RAG_ERR_HIP, "building the LayerNorm-folded weight images failed");
    if (*h->range_pin) {   // a folded weight (W gamma) outside fp16's range: keep the separate LayerNorm launches
        *h->range_pin = 0;
        return RAG_OK;
    }
    h->lnf_built = true;
    return RAG_OK;
}
}  // namespace

namespace {

// Last node of an encoder graph: copy the pooled rows to where this call's caller wants them and publish the range flag.
// Both destinations and the float count come from the cells (device memory, refreshed by the graph's first copies).
__global__ void graph_epilogue_kernel(const float* src, const unsigned long long* cells, const uint32_t* gflag) {
    float* dst = reinterpret_cast<float*>(cells[0]);
    uint32_t* flag = reinterpret_cast<uint32_t*>(cells[1]);
    const unsigned long long n = cells[2];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0 && flag && *gflag) *flag = 1u;
}

int ensure_graph_buffers(rag_bert* h) {
    if (h->g_blk) return RAG_OK;
    for (int i = 0; i < kGraphStages; ++i) {
        RAGC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->g_pin[i]), (size_t)kGraphBlockInts * sizeof(int), hipHostMallocDefault));
        RAGC_HIP_TRY(hipEventCreateWithFlags(&h->g_up[i], hipEventDisableTiming));
    }
    RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->g_out), (size_t)kGraphSeqs * h->cfg.hidden * sizeof(float)));
    RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->g_flag), sizeof(uint32_t)));
    int* blk = nullptr;
    RAGC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&blk), (size_t)kGraphBlockInts * sizeof(int)));
    h->g_ids = blk;
    h->g_types = blk + kGraphTokens;
    h->g_cu = blk + 2 * kGraphTokens;
    h->g_cells = reinterpret_cast<unsigned long long*>(blk + kGraphCellsAt);
    h->g_blk = blk;
    // A graphed shape must never grow the workspace: its graph — and every other cached one — holds the buffers'
    // addresses.  Size it once for the largest shape the graph path takes.  (A later eager pass with more tokens still
    // reallocates; ensure_ws drops the cached graphs when it does.)
    if (kGraphTokens > h->ws_tokens || kGraphSeqs > h->ws_seqs) {
        if (h->ws_used) RAGC_HIP_TRY(hipEventSynchronize(h->ws_event));
        int rc = ensure_ws(h, kGraphTokens, kGraphSeqs);
        if (rc) return rc;
    }
    return RAG_OK;
}

// everything a graphed pass replays, in order (the block's upload goes in front of it, outside the graph); called once
// eagerly (first-use set-up of the kernels) and once under capture
int graph_body(rag_bert* h, int nseq_pad, int T_pad, int maxlen_b, int out_kind, int normalize, hipStream_t st) {
    RAGC_HIP_TRY(hipMemsetAsync(h->g_flag, 0, sizeof(uint32_t), st));
    int rc = forward_locked(h, h->g_ids, h->cfg.type_vocab > 0 ? h->g_types : nullptr, h->g_cu, nseq_pad, T_pad, maxlen_b, out_kind,
                            normalize, h->g_out, st, h->g_flag);
    if (rc) return rc;
    graph_epilogue_kernel<<<dim3(32), dim3(256), 0, st>>>(h->g_out, h->g_cells, h->g_flag);
    RAGC_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// the graph for a padded shape: from the cache, or captured now (after one eager run of the same body)
int get_graph(rag_bert* h, int nseq_pad, int T_pad, int maxlen_b, int out_kind, int normalize, hipStream_t st, hipGraphExec_t* out) {
    for (auto& g : h->graphs)
        if (g.nseq_pad == nseq_pad && g.T_pad == T_pad && g.maxlen_b == maxlen_b && g.out_kind == out_kind && g.normalize == normalize) {
            g.last_use = ++h->graph_clock;
            *out = g.exec;
            return RAG_OK;
        }
    // first use of this shape: make sure the workspace fits and every kernel has had its one-time set-up, eagerly
    if (T_pad > h->ws_tokens || nseq_pad > h->ws_seqs) {
        if (h->ws_used) RAGC_HIP_TRY(hipEventSynchronize(h->ws_event));
        RAGC_HIP_TRY(hipStreamSynchronize(st));
        int rc = ensure_ws(h, T_pad, nseq_pad);
        if (rc) return rc;
    }
    int rc = graph_body(h, nseq_pad, T_pad, maxlen_b, out_kind, normalize, st);   // (writes through the cells: harmless, the replay follows)
    if (rc) return rc;
    RAGC_HIP_TRY(hipStreamSynchronize(st));
    hipGraph_t graph = nullptr;
    RAGC_HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    h->in_capture = true;
    rc = graph_body(h, nseq_pad, T_pad, maxlen_b, out_kind, normalize, st);
    h->in_capture = false;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return rc ? rc : ragc_fail(RAG_ERR_HIP, "capturing the encoder graph failed: %s", hipGetErrorString(e));
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) return ragc_fail(RAG_ERR_HIP, "instantiating the encoder graph failed: %s", hipGetErrorString(ei));
    if ((int)h->graphs.size() >= kGraphCache) {   // evict the least recently used
        size_t victim = 0;
        for (size_t i = 1; i < h->graphs.size(); ++i)
            if (h->graphs[i].last_use < h->graphs[victim].last_use) victim = i;
        RAGC_HIP_TRY(hipStreamSynchronize(st));
        (void)hipGraphExecDestroy(h->graphs[victim].exec);
        h->graphs.erase(h->graphs.begin() + (long)victim);
    }
    h->graphs.push_back(rag_bert::EncGraph{nseq_pad, T_pad, maxlen_b, out_kind, normalize, exec, ++h->graph_clock});
    *out = exec;
    return RAG_OK;
}

// rag_bert_forward_to_device through a graph.  Returns kNotGraphed when the batch does not qualify (the caller takes the
// eager path).
constexpr int kNotGraphed = -1;
int forward_to_device_graph(rag_bert* h, const int32_t* ids, const int32_t* type_ids, const int32_t* cu, int nseq, int T, int max_len,
                            int out_kind, int normalize, float* out_dev, uint32_t* range_flag, hipStream_t st) {
    if (!h->use_graphs || h->background || (out_kind != RAG_BERT_OUT_MEAN && out_kind != RAG_BERT_OUT_CLS)) return kNotGraphed;
    // padding: at least one dummy sequence, sequences to a multiple of 8, tokens to a multiple of 64 (every dummy >= 1 token)
    const int nseq_pad = (nseq + 1 + 7) / 8 * 8, n_dummy = nseq_pad - nseq;
    const int T_pad = (T + n_dummy + 63) / 64 * 64;
    if (nseq_pad > kGraphSeqs || T_pad > kGraphTokens) return kNotGraphed;
    const int extra = T_pad - T;                                  // tokens the dummies hold
    const int dummy_len = (extra + n_dummy - 1) / n_dummy;        // the longest dummy
    if (dummy_len + h->cfg.pos_offset > h->cfg.max_positions) return kNotGraphed;
    const int maxlen_b = (std::max(max_len, dummy_len) + 31) / 32 * 32;
    int rc = ensure_graph_buffers(h);
    if (rc) return rc;
    const unsigned turn = h->g_turn++ % kGraphStages;
    if (h->g_up_used[turn]) RAGC_HIP_TRY(hipEventSynchronize(h->g_up[turn]));   // this block's previous upload has left it
    int* pin = h->g_pin[turn];
    int* pin_ids = pin;
    int* pin_types = pin + kGraphTokens;
    int* pin_cu = pin + 2 * kGraphTokens;
    unsigned long long* pin_cells = reinterpret_cast<unsigned long long*>(pin + kGraphCellsAt);
    std::memcpy(pin_ids, ids, (size_t)T * sizeof(int));
    std::memset(pin_ids + T, 0, (size_t)extra * sizeof(int));
    if (type_ids) std::memcpy(pin_types, type_ids, (size_t)T * sizeof(int));
    else std::memset(pin_types, 0, (size_t)T * sizeof(int));
    std::memset(pin_types + T, 0, (size_t)extra * sizeof(int));
    std::memcpy(pin_cu, cu, (size_t)(nseq + 1) * sizeof(int));
    for (int i = 0, at = T, left = extra; i < n_dummy; ++i) {   // dummies share the padding tokens as evenly as possible
        const int len = (left + (n_dummy - i) - 1) / (n_dummy - i);
        at += len;
        left -= len;
        pin_cu[nseq + 1 + i] = at;
    }
    pin_cells[0] = reinterpret_cast<unsigned long long>(out_dev);
    pin_cells[1] = reinterpret_cast<unsigned long long>(range_flag);
    pin_cells[2] = (unsigned long long)nseq * h->cfg.hidden;
    // one upload for ids, types, boundaries and cells, behind whatever pass of this handle is still reading the device block
    if (h->ws_used && h->ws_stream != st) RAGC_HIP_TRY(hipStreamWaitEvent(st, h->ws_event, 0));
    RAGC_HIP_TRY(hipMemcpyAsync(h->g_blk, pin, (size_t)kGraphBlockInts * sizeof(int), hipMemcpyHostToDevice, st));
    RAGC_HIP_TRY(hipEventRecord(h->g_up[turn], st));
    h->g_up_used[turn] = true;
    hipGraphExec_t exec = nullptr;
    rc = get_graph(h, nseq_pad, T_pad, maxlen_b, out_kind, normalize, st, &exec);
    if (rc) {   // a runtime that cannot capture this (rag_last_error says why): the eager path serves, from now on
        h->use_graphs = false;
        (void)hipGetLastError();
        return kNotGraphed;
    }
    RAGC_HIP_TRY(hipGraphLaunch(exec, st));
    if (hipEventRecord(h->ws_event, st) == hipSuccess) {
        h->ws_stream = st;
        h->ws_used = true;
    }
    return RAG_OK;
}

}  // namespace

extern "C" int32_t rag_bert_weight_count(const rag_bert_config* cfg) { return cfg ? weight_count(*cfg) : 0; }

extern "C" int rag_bert_create(const rag_bert_config* cfg, const void* const* weights_dev, int32_t n_weights,
                               int32_t device, rag_bert** out) {
    if (!out) return ragc_fail(RAG_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!cfg || !weights_dev) return ragc_fail(RAG_ERR_INVALID_ARG, "null config or weight table");
    const rag_bert_config& c = *cfg;
    if (c.hidden <= 0 || c.hidden > 256 * ragb::kMaxChunks || c.hidden % 32)
        return ragc_fail(RAG_ERR_UNSUPPORTED, "hidden=%d (need a multiple of 32, <= 1024)", c.hidden);
    if (c.n_heads <= 0 || c.hidden % c.n_heads || (c.hidden / c.n_heads != 32 && c.hidden / c.n_heads != 64))
        return ragc_fail(RAG_ERR_UNSUPPORTED, "head dim %d (need 32 or 64)", c.n_heads > 0 ? c.hidden / c.n_heads : 0);
    if (c.intermediate <= 0 || c.intermediate % 32)
        return ragc_fail(RAG_ERR_UNSUPPORTED, "intermediate=%d (need a multiple of 32)", c.intermediate);
    if (c.n_layers <= 0 || c.vocab_size <= 0 || c.max_positions <= 0 || c.type_vocab < 0)
        return ragc_fail(RAG_ERR_INVALID_ARG, "bad model dimensions");
    if (map_act(c.act) < 0) return ragc_fail(RAG_ERR_INVALID_ARG, "unknown activation %d", c.act);
    if (c.head < RAG_HEAD_NONE || c.head > RAG_HEAD_ROBERTA) return ragc_fail(RAG_ERR_INVALID_ARG, "unknown head %d", c.head);
    if (c.head != RAG_HEAD_NONE && c.n_labels <= 0) return ragc_fail(RAG_ERR_INVALID_ARG, "n_labels must be positive");
    if (n_weights != weight_count(c))
        return ragc_fail(RAG_ERR_INVALID_ARG, "expected %d weight pointers, got %d", weight_count(c), n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!weights_dev[i] && !(i == 2 && c.type_vocab == 0))
            return ragc_fail(RAG_ERR_INVALID_ARG, "weight pointer %d is null", i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ragc_fail(RAG_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return ragc_fail(RAG_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    RagcDeviceGuard g(device);
    if (!g.ok) return ragc_fail(RAG_ERR_HIP, "hipSetDevice(%d) failed", device);
    rag_bert* h = new (std::nothrow) rag_bert();
    if (!h) return ragc_fail(RAG_ERR_OOM, "host allocation failed");
    h->cfg = c;
    h->device = device;
    h->w.assign(reinterpret_cast<const float* const*>(weights_dev), reinterpret_cast<const float* const*>(weights_dev) + n_weights);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ws_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->stage_event, hipEventDisableTiming) != hipSuccess) {
        if (h->stream) (void)hipStreamDestroy(h->stream);
        if (h->ws_event) (void)hipEventDestroy(h->ws_event);
        delete h;
        return ragc_fail(RAG_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cus = prop.multiProcessorCount;
    const char* va = getenv("RAG_AMD_VALU_ATTENTION");
    h->valu_attention = va && *va == '1';
    const char* rm = getenv("RAG_AMD_ROW_MAJOR");
    h->row_major_only = rm && *rm == '1';
    const char* eg = getenv("RAG_AMD_ENCODER_GRAPH");
    h->use_graphs = !(eg && *eg == '0');
    const char* w6e = getenv("RAG_AMD_GEMM_W6");
    h->use_w6 = w6e && *w6e == '1';
    const char* ta = getenv("RAG_AMD_TILED_ATTENTION_F32");
    h->tiled_attention_f32 = ta && *ta == '1';
    if (hipHostMalloc(reinterpret_cast<void**>(&h->range_pin), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        rag_bert_destroy(h);
        return ragc_fail(RAG_ERR_OOM, "allocation of the range flag failed");
    }
    *h->range_pin = 0;
    // fragment-order images of the layer GEMM weights (read once, here: later in-place edits of the caller's
    // tensors are not seen by the big-batch path)
    int rc = RAG_OK;
    if (c.gemm_mode == RAG_GEMM_F32) rc = build_images(h, 2);
    else if (c.gemm_mode == RAG_GEMM_F16) rc = build_images(h, 1);
    if (rc) {
        rag_bert_destroy(h);
        return rc;
    }
    if (c.gemm_mode == RAG_GEMM_F32) {
        // a weight outside fp16's range (never seen in a BERT-family checkpoint) invalidates the two-plane images
        if (*h->range_pin) {   // (build_images has synchronised the stream the pack kernels ran on)
            h->weights_fit_f16 = false;
            *h->range_pin = 0;
            if ((rc = build_images(h, 0))) {
                rag_bert_destroy(h);
                return rc;
            }
        }
    }
    if (c.gemm_mode == RAG_GEMM_F16 && c.hidden == 384 && c.intermediate % 128 == 0 && c.act == RAG_ACT_GELU) {
        // Built, exact, and NOT faster (round 4: 7.54 ms per 3200-pair pass against 7.46 with the two GEMMs): with one wave
        // per SIMD — the price of keeping a 32 x 384 fp32 tile and the X fragments in registers — nothing overlaps the
        // GELU's vector work or the weight stream (ablations: the skeleton alone, no DMA / MFMA / LDS reads, takes 0.49 of
        // the 0.73 ms per layer).  It does take the intermediate's 6.6 GB per pass off HBM.  Opt-in.
        const char* ff = getenv("RAG_AMD_FFN_FUSED");
        if (ff && *ff == '1') {
            h->wffn2p.assign((size_t)c.n_layers, nullptr);
            const int H = c.hidden, I = c.intermediate;
            for (int l = 0; l < c.n_layers && rc == RAG_OK; ++l) {
                void* img = nullptr;
                if (hipMalloc(&img, (size_t)H * I * sizeof(_Float16)) != hipSuccess) {
                    rc = ragc_fail(RAG_ERR_OOM, "device allocation of the fused feed-forward weight image failed");
                    break;
                }
                h->wffn2p[(size_t)l] = static_cast<_Float16*>(img);
                const float* W2 = h->w[kEmbEntries + kPerLayer * l + 8];
                ragb::pack_f16_accorder_kernel<<<dim3((unsigned)(((long long)H * I + 255) / 256)), dim3(256), 0, h->stream>>>(W2, H, I, I, h->wffn2p[(size_t)l]);
            }
            if (rc == RAG_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess))
                rc = ragc_fail(RAG_ERR_HIP, "building the fused feed-forward weight images failed");
            if (rc) {
                rag_bert_destroy(h);
                return rc;
            }
        }
    }
    if (c.gemm_mode == RAG_GEMM_F32 && h->weights_fit_f16) {
        // Measured slower and therefore OFF unless asked for (RAG_AMD_ENCODER_LN_FOLD=1): handing a tile's slabs from one
        // workgroup to another inside a launch costs two to three memory-side round trips (agent-scope stores, the arrival
        // atomic, agent-scope loads: 10-19 us per GEMM) — more than the kernel boundary and the 6 us LayerNorm launch it
        // replaces (bge-base, 32 queries: 1.37 ms folded against 1.00 ms).
        // The big-batch form has no hand-off (no split-K there): the GEMM epilogues leave the statistics and apply them —
        // on by default (RAG_AMD_LN_FOLD=0: the separate LayerNorm passes, A/B checks).
        const char* lf = getenv("RAG_AMD_ENCODER_LN_FOLD");
        const char* lb = getenv("RAG_AMD_LN_FOLD");
        const bool want_small = lf && *lf == '1', want_big = !(lb && *lb == '0') && c.hidden % 128 == 0;
        if ((want_small || want_big) && (rc = build_folded(h))) {
            rag_bert_destroy(h);
            return rc;
        }
        h->lnf_ok = h->lnf_built && want_small;
        h->lnf_big = h->lnf_built && want_big;
    }
    *out = h;
    return RAG_OK;
}

extern "C" int rag_bert_destroy(rag_bert* h) {
    if (!h) return RAG_OK;
    {
        RagcDeviceGuard g(h->device);
        std::lock_guard<std::mutex> lk(h->mu);
        (void)hipDeviceSynchronize();
        void* ptrs[] = {h->x, h->y, h->qkv, h->ctx, h->ffn, h->pooled, h->pooled2, h->logits, h->probs,
                        h->ids_dev, h->types_dev, h->cu_dev, h->out_dev};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        for (__bf16* p : h->wx)
            if (p) (void)hipFree(p);
        for (_Float16* p : h->wx2)
            if (p) (void)hipFree(p);
        for (_Float16* p : h->wxf)
            if (p) (void)hipFree(p);
        for (_Float16* p : h->wfold)
            if (p) (void)hipFree(p);
        for (_Float16* p : h->wffn2p)
            if (p) (void)hipFree(p);
        for (float* p : h->fold_s)
            if (p) (void)hipFree(p);
        for (float* p : h->fold_b)
            if (p) (void)hipFree(p);
        void* lptrs[] = {h->y1, h->ts_a, h->ts_b, h->rs_a, h->rs_b, h->tile_ctr};
        for (void* p : lptrs)
            if (p) (void)hipFree(p);
        for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
        void* gptrs[] = {h->g_blk, h->g_out, h->g_flag};
        for (void* p : gptrs)
            if (p) (void)hipFree(p);
        for (int i = 0; i < kGraphStages; ++i) {
            if (h->g_pin[i]) (void)hipHostFree(h->g_pin[i]);
            if (h->g_up[i]) (void)hipEventDestroy(h->g_up[i]);
        }
        if (h->range_pin) (void)hipHostFree(h->range_pin);
        if (h->stage_pin) (void)hipHostFree(h->stage_pin);
        if (h->ws_event) (void)hipEventDestroy(h->ws_event);
        if (h->stage_event) (void)hipEventDestroy(h->stage_event);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return RAG_OK;
}

extern "C" int rag_bert_forward_device(rag_bert* h, const int32_t* ids_dev, const int32_t* type_ids_dev,
                                       const int32_t* cu_seqlens_dev, int32_t nseq, int32_t total_tokens,
                                       int32_t max_seq_len, int32_t out_kind, int32_t normalize, float* out_dev,
                                       uint32_t* range_flag, void* stream) {
    int rc = check_forward(h, nseq, out_kind);
    if (rc) return rc;
    if (!ids_dev || !cu_seqlens_dev || !out_dev || total_tokens <= 0 || max_seq_len <= 0)
        return ragc_fail(RAG_ERR_INVALID_ARG, "null buffer or empty batch");
    RagcDeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    return forward_locked(h, ids_dev, type_ids_dev, cu_seqlens_dev, nseq, total_tokens, max_seq_len, out_kind, normalize,
                          out_dev, (hipStream_t)stream, range_flag ? range_flag : h->range_pin);
}

namespace {
// cu_seqlens checks shared by the host-ids entry points; returns the longest sequence through *max_len
int check_host_batch(const rag_bert* h, const int32_t* cu_seqlens, int nseq, int* max_len) {
    if (cu_seqlens[0] != 0) return ragc_fail(RAG_ERR_INVALID_ARG, "cu_seqlens[0] must be 0");
    int mx = 0;
    for (int s = 0; s < nseq; ++s) {
        const int len = cu_seqlens[s + 1] - cu_seqlens[s];
        if (len <= 0) return ragc_fail(RAG_ERR_INVALID_ARG, "sequence %d is empty", s);
        if (len + h->cfg.pos_offset > h->cfg.max_positions)
            return ragc_fail(RAG_ERR_INVALID_ARG, "sequence %d has %d tokens; the model holds %d positions", s, len,
                             h->cfg.max_positions - h->cfg.pos_offset);
        mx = std::max(mx, len);
    }
    *max_len = mx;
    return RAG_OK;
}
}  // namespace

extern "C" int rag_bert_forward_to_device(rag_bert* h, const int32_t* ids, const int32_t* type_ids,
                                          const int32_t* cu_seqlens, int32_t nseq, int32_t out_kind, int32_t normalize,
                                          float* out_dev, uint32_t* range_flag, void** stream_out) {
    int rc = check_forward(h, nseq, out_kind);
    if (rc) return rc;
    if (!ids || !cu_seqlens || !out_dev || !stream_out) return ragc_fail(RAG_ERR_INVALID_ARG, "null buffer");
    int max_len = 0;
    rc = check_host_batch(h, cu_seqlens, nseq, &max_len);
    if (rc) return rc;
    const int T = cu_seqlens[nseq];
    RagcDeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = h->user_stream ? h->user_stream : h->stream;
    // small batches (the query encoder): one graph replay instead of ~45-90 launches
    rc = forward_to_device_graph(h, ids, type_ids, cu_seqlens, nseq, T, max_len, out_kind, normalize, out_dev,
                                 range_flag ? range_flag : h->range_pin, st);
    if (rc == RAG_OK) {
        *stream_out = (void*)st;
        return RAG_OK;
    }
    if (rc != kNotGraphed) return rc;
    rc = grow(&h->ids_dev, &h->ids_cap, (long long)T);
    if (rc) return rc;
    if (type_ids && (rc = grow(&h->types_dev, &h->types_cap, (long long)T))) return rc;
    rc = grow(&h->cu_dev, &h->cu_cap, (long long)nseq + 1);
    if (rc) return rc;
    // one pinned block [ids | type_ids | cu]: the caller's arrays are free again when this call returns, and the
    // uploads are truly asynchronous (from pageable memory the runtime would stage them synchronously)
    const long long need = 2LL * T + nseq + 1;
    if (h->stage_used) RAGC_HIP_TRY(hipEventSynchronize(h->stage_event));  // the previous batch has left the block
    if (need > h->stage_cap) {
        if (h->stage_pin) (void)hipHostFree(h->stage_pin);
        h->stage_pin = nullptr;
        h->stage_cap = 0;
        RAGC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->stage_pin), (size_t)(need + need / 2) * sizeof(int), hipHostMallocDefault));
        h->stage_cap = need + need / 2;
    }
    int* pin_ids = h->stage_pin;
    int* pin_types = h->stage_pin + T;
    int* pin_cu = h->stage_pin + 2LL * T;
    std::memcpy(pin_ids, ids, (size_t)T * sizeof(int));
    if (type_ids) std::memcpy(pin_types, type_ids, (size_t)T * sizeof(int));
    std::memcpy(pin_cu, cu_seqlens, (size_t)(nseq + 1) * sizeof(int));
    RAGC_HIP_TRY(hipMemcpyAsync(h->ids_dev, pin_ids, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
    if (type_ids) RAGC_HIP_TRY(hipMemcpyAsync(h->types_dev, pin_types, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
    RAGC_HIP_TRY(hipMemcpyAsync(h->cu_dev, pin_cu, (size_t)(nseq + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    RAGC_HIP_TRY(hipEventRecord(h->stage_event, st));
    h->stage_used = true;
    rc = forward_locked(h, h->ids_dev, type_ids ? h->types_dev : nullptr, h->cu_dev, nseq, T, max_len, out_kind, normalize,
                        out_dev, st, range_flag ? range_flag : h->range_pin);
    if (rc) return rc;
    *stream_out = (void*)st;
    return RAG_OK;
}

extern "C" int rag_bert_forward(rag_bert* h, const int32_t* ids, const int32_t* type_ids, const int32_t* cu_seqlens,
                                int32_t nseq, int32_t out_kind, int32_t normalize, float* out) {
    int rc = check_forward(h, nseq, out_kind);
    if (rc) return rc;
    if (!ids || !cu_seqlens || !out) return ragc_fail(RAG_ERR_INVALID_ARG, "null buffer");
    int max_len = 0;
    rc = check_host_batch(h, cu_seqlens, nseq, &max_len);
    if (rc) return rc;
    const int T = cu_seqlens[nseq];
    RagcDeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    // the id / cu buffers below are also the asynchronous entry point's: nothing of its may still be in flight
    // when they are (re)allocated or overwritten from pageable memory
    hipStream_t st = h->user_stream ? h->user_stream : h->stream;
    RAGC_HIP_TRY(hipStreamSynchronize(st));
    rc = grow(&h->ids_dev, &h->ids_cap, (long long)T);
    if (rc) return rc;
    if (type_ids) {
        rc = grow(&h->types_dev, &h->types_cap, (long long)T);
        if (rc) return rc;
    }
    rc = grow(&h->cu_dev, &h->cu_cap, (long long)nseq + 1);
    if (rc) return rc;
    const size_t n_out = out_elems(h->cfg, out_kind, nseq, T);
    rc = grow(&h->out_dev, &h->out_cap, (long long)n_out);
    if (rc) return rc;
    RAGC_HIP_TRY(hipMemcpyAsync(h->ids_dev, ids, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
    if (type_ids) RAGC_HIP_TRY(hipMemcpyAsync(h->types_dev, type_ids, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
    RAGC_HIP_TRY(hipMemcpyAsync(h->cu_dev, cu_seqlens, (size_t)(nseq + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    *h->range_pin = 0;   // (the stream is idle: synchronised above)
    rc = forward_locked(h, h->ids_dev, type_ids ? h->types_dev : nullptr, h->cu_dev, nseq, T, max_len, out_kind, normalize,
                        h->out_dev, st, h->range_pin);
    if (rc) return rc;
    RAGC_HIP_TRY(hipMemcpyAsync(out, h->out_dev, n_out * sizeof(float), hipMemcpyDeviceToHost, st));
    RAGC_HIP_TRY(hipStreamSynchronize(st));
    if (*h->range_pin) {
        // An activation outside fp16's range (|x| >= 65504) went through the two-plane GEMMs: that pass is void.
        // Repeat it with fp32's exponent range: big batches on the split-bf16 images (built now if this is the
        // first time), small ones on the fp32 MFMA.
        *h->range_pin = 0;
        if ((rc = build_images(h, 0))) return rc;
        h->force_x6 = true;
        rc = forward_locked(h, h->ids_dev, type_ids ? h->types_dev : nullptr, h->cu_dev, nseq, T, max_len, out_kind,
                            normalize, h->out_dev, st, h->range_pin);
        h->force_x6 = false;
        if (rc) return rc;
        ++h->range_events;
        RAGC_HIP_TRY(hipMemcpyAsync(out, h->out_dev, n_out * sizeof(float), hipMemcpyDeviceToHost, st));
        RAGC_HIP_TRY(hipStreamSynchronize(st));
    }
    return RAG_OK;
}

extern "C" int rag_bert_set_background(rag_bert* h, int32_t on) {
    if (!h) return ragc_fail(RAG_ERR_INVALID_ARG, "null model handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->background = on != 0;
    return RAG_OK;
}

extern "C" int rag_bert_set_stream(rag_bert* h, void* stream) {
    if (!h) return ragc_fail(RAG_ERR_INVALID_ARG, "null model handle");
    RagcDeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    // nothing enqueued on the stream that is being left may still be running on the shared buffers
    RAGC_HIP_TRY(hipStreamSynchronize(h->user_stream ? h->user_stream : h->stream));
    if (!h->graphs.empty()) {   // (graphs replay on any stream, but their capture stream is gone: rebuild on the new one)
        for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);
        h->graphs.clear();
    }
    h->user_stream = (hipStream_t)stream;
    return RAG_OK;
}

extern "C" int rag_bert_set_cu_budget(rag_bert* h, int32_t n_cus) {
    if (!h) return ragc_fail(RAG_ERR_INVALID_ARG, "null model handle");
    hipDeviceProp_t prop;
    RAGC_HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    if (n_cus < 0 || n_cus > prop.multiProcessorCount)
        return ragc_fail(RAG_ERR_INVALID_ARG, "CU budget %d outside [0, %d]", n_cus, prop.multiProcessorCount);
    std::lock_guard<std::mutex> lk(h->mu);
    const int want = n_cus == 0 ? prop.multiProcessorCount : n_cus;
    if (want != h->n_cus && !h->graphs.empty()) {
        // cached encoder graphs carry the tile and split-K choices of the old budget: drop them (after their last replay)
        RagcDeviceGuard g(h->device);
        RAGC_HIP_TRY(hipDeviceSynchronize());
        for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);
        h->graphs.clear();
    }
    h->n_cus = want;
    return RAG_OK;
}

extern "C" int rag_bert_range_events(rag_bert* h, int64_t* repeated_passes, int32_t* pending) {
    if (!h) return ragc_fail(RAG_ERR_INVALID_ARG, "null model handle");
    RagcDeviceGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->mu);
    if (repeated_passes) *repeated_passes = h->range_events;
    if (pending) {   // an asynchronous pass that used the handle's own word raised it since it was last taken
        RAGC_HIP_TRY(hipDeviceSynchronize());
        *pending = *h->range_pin ? 1 : 0;
        *h->range_pin = 0;
    }
    return RAG_OK;
}
