// flat_kernels.hip.h — gfx950 kernels of the flat (exhaustive) index.
//
// Replaces the arithmetic behind faiss IndexFlat{IP,L2}.search as called from the reference at
// src/pipeline/components/faiss_store.py:152.  Semantics (summation order, tie rule, padding)
// are those of oracle/flat_oracle.c, which the tests hold this file to bit-for-bit.
//
// K1  scan_topk_kernel   one pass over the corpus: 32-row x 32-query score tiles on the exact
//                        fp32 MFMA (v_mfma_f32_32x32x2_f32 == fmaf chain), threshold filter in
//                        registers, candidates appended to per-workgroup LDS buffers, buffers
//                        compacted by a wave-level bitonic sort when one overflows.
// K2  tournament_merge_kernel  per query: top-k of the sorted per-workgroup (or per-shard) lists by a
//                        k-round workgroup-wide max tournament; decodes keys into (score, id).
// K1' scan_topk_kernel<P = 1> + screen_* kernels  the optional two-stage exact search: fp16 screening pass,
//                        candidate collection, canonical fp32 re-scoring, certificate, fp32 fallback.
// plus small helpers (sample pass thresholds, row norms, synthetic corpus, neutral fill).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Experiment-only switches (scripts/tune_scan.py builds side libraries with them; never defined in
// the product build):  RAGK_ABLATE_NO_MFMA keeps the load stream and drops the matrix work,
// RAGK_ABLATE_L2_WINDOW keeps the matrix work and reads a cache-resident window of the corpus.
#if defined(RAGK_ABLATE_NO_MFMA)
#define RAGK_MFMA(x, q, acc) ([&] { asm volatile("" ::"v"(x), "v"(q)); return acc; }())
#else
#define RAGK_MFMA(x, q, acc) __builtin_amdgcn_mfma_f32_32x32x2f32(x, q, acc, 0, 0, 0)
#endif

namespace ragk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

constexpr int kQT = 32;        // queries per scan pass (MFMA N)
constexpr int kTileRows = 32;  // corpus rows per wave tile (MFMA M)
constexpr int kMaxScanWaves = 16;

// Lowest ranking score a result may have: the filters test `score >= kScoreFloor`, i.e. score > -FLT_MAX.
// That is the contract of the heap behind faiss IndexFlat.search (the reference's call, faiss_store.py:152):
// its top-k heap starts at -FLT_MAX (IP; +FLT_MAX for L2) and a candidate enters only if it compares
// strictly better, so rows whose score is NaN, -inf or -FLT_MAX are never returned and their slots stay
// (-1, -FLT_MAX).  include/rag_amd.h states the same contract; oracle/flat_oracle.c follows it.
#define RAGK_SCORE_FLOOR __uint_as_float(0xFF7FFFFEu) /* nextafter(-FLT_MAX, 0) */

// ---- ranking keys -------------------------------------------------------------------------

__device__ __forceinline__ uint32_t ord32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// larger key = better: higher score first, then smaller row.  0 = empty slot.
__device__ __forceinline__ u64 make_key(float score, uint32_t row) {
    return ((u64)ord32(score) << 32) | (u64)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((u64)hi << 32) | lo;
}

// Wave-level bitonic sort, descending over index i = e*64 + lane, E keys per lane, of NQ independent
// key sets at once.  Each compare-exchange stage costs a ds_bpermute round trip per set; running the
// wave's NQ sets through the network together overlaps those latencies (one set at a time a
// compaction of 4 queries took ~6 us, i.e. most of the scan's fixed overhead on small shards).
template <int E, int NQ>
__device__ __forceinline__ void wave_sort_desc(u64 (&key)[NQ][E], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * E; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int se = stride >> 6;
#pragma unroll
                for (int s = 0; s < NQ; ++s) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & se) == 0) {
                            const bool up = (((e * 64) & size) == 0);  // size >= 128 here: lane-independent
                            u64 a = key[s][e], b = key[s][e | se];
                            u64 hi = umax64(a, b), lo = umin64(a, b);
                            key[s][e] = up ? hi : lo;
                            key[s][e | se] = up ? lo : hi;
                        }
                    }
                }
            } else {
                u64 other[NQ][E];
#pragma unroll
                for (int s = 0; s < NQ; ++s)
#pragma unroll
                    for (int e = 0; e < E; ++e) other[s][e] = shfl_xor_u64(key[s][e], stride);
                const bool lower = ((lane & stride) == 0);
#pragma unroll
                for (int s = 0; s < NQ; ++s) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool up = (((e * 64 + lane) & size) == 0);
                        key[s][e] = (lower == up) ? umax64(key[s][e], other[s][e]) : umin64(key[s][e], other[s][e]);
                    }
                }
            }
        }
    }
}

// ---- K1: scan + select ----------------------------------------------------------------------
// (The corpus stream uses plain loads.  With the non-temporal hint — a natural idea for bytes read once per
// batch — the scan ran at HALF the rate, 9.96 ms instead of 5.14 ms on 10M x 768: the four 16-byte pieces of a
// row's 128-byte line are requested by four separate instructions and rely on the vector L1 to merge them.)

// One ring slot (32 bytes of a row across the two lane halves) times the matching query fragment.
template <int P>
__device__ __forceinline__ f32x16 scan_step(const f32x4 x, const f32x4 q, f32x16 acc) {
    if constexpr (P == 1) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, q), acc,
                                                      0, 0, 0);
    } else {
        acc = RAGK_MFMA(x[0], q[0], acc);
        acc = RAGK_MFMA(x[1], q[1], acc);
        acc = RAGK_MFMA(x[2], q[2], acc);
        acc = RAGK_MFMA(x[3], q[3], acc);
        return acc;
    }
}

// MAP mode of the scan kernel (the IVFFlat nprobe mode's screening pass, ivf_kernels.hip.h): the tiles to visit and, per
// tile, which of the pass's queries may take candidates from it
struct TileMapEntry {
    uint32_t tile;    // tile number (rows 32 * tile ...)
    uint32_t mask;    // bit q: query q may take candidates from this tile
};

struct ScanParams {
    const float* X;        // corpus, row-major, row_stride floats per row (zero padded to d8)
    const float* xnorm;    // canonical squared norms (L2 metric only)
    const float* qnorm;    // canonical ||q||^2 per query (L2 metric only): a non-finite one means no results
    const float* Q;        // queries, row-major nq x d (un-padded)
    u64* partial;          // out: [kQT][grid][k] keys, sorted descending per (query, workgroup)
    const u64* ceil;       // [kQT] or null: only keys strictly below ceil[q] are candidates (k > max_k rounds)
    float* acc_io;         // [n_tiles*32][kQT] running inner products (d > 1024 is scanned in column chunks)
    int col0;              // first column of this launch's chunk (multiple of 8)
    int dc8;               // columns in this chunk, multiple of 8 (== d8 when the scan is not chunked)
    int acc_in;            // start from acc_io instead of zero
    int acc_out;           // store the accumulators to acc_io and skip the selection
    long long n_rows;
    long long row_stride;
    int d;                 // logical dimension
    int d8;                // d rounded up to a multiple of 8
    int nq;                // 1..32
    int k;
    int n_tiles;           // ceil(n_rows / 32)
    int n_iters;           // rounds of the tile walk (the last one may be partial)
    int n_full;            // rounds in which every wave of every workgroup has a tile: n_tiles / (grid * NW)
    const uint32_t* enable;  // null, or a device word: unless it equals `epoch` the whole launch is a no-op (fallback path)
    uint32_t epoch;          // id of the two-stage search this launch belongs to (never 0): per-search flags are
                             // "set" when they hold it, so nothing has to be cleared between searches
    const u64* thr_key;      // null, or [kQT] keys (0 = none) whose score >= k rows are known to reach (sample pass):
                             // the filter starts there instead of at -inf
    int tile_step;           // tile t covers rows [32 t tile_step, +32): 1 for a full scan, > 1 for the sample pass
    // screening pass (P == 1) only
    int kout;              // keys emitted per (query, workgroup), >= k
    float x_absmax, x_normmax, x_scale;  // corpus statistics behind the error bound, and the corpus' fp16 scale
    float* margin_out;     // [kQT] workgroup 0 publishes each query's band width for the resolve kernel
    uint32_t* lossy;       // [kQT] = epoch when the certificate is void from the start (query outside the range the
                           // bound covers)
    uint32_t* wg_lossy;    // [kQT][grid] = epoch when THIS workgroup could not hold a query's band and fell back to
                           // its best k for that query (null: not recorded — sample pass)
    uint32_t* flag_clear;  // null, or the caller's "a certificate failed" word (RAG_SEARCH_DEFER_FALLBACK): workgroup 0
                           // zeroes it here, one kernel boundary ahead of the resolve kernel that may set it
    // dynamic deal of the last rounds ("tile walk" in the kernel).  dyn_ctr == null: everything is dealt statically.
    uint32_t* dyn_ctr;       // this launch's ticket counter (0 at launch)
    uint32_t* dyn_ctr_next;  // the next launch's counter: workgroup 0 zeroes it (launches of an index are stream-ordered)
    int dyn_tile0;           // first tile of the dynamic region = n_full * grid * NW
    int n_dyn_groups;        // tickets [0, n_dyn_groups): whole NW-tile groups; then n_singles tickets of one tile each
    int n_singles;
    // MAP == true only: the tile walk goes over map[0 .. *map_count) — item g = entries [NW g, NW g + NW), entry w for wave w;
    // the first two items of a workgroup by formula, the rest by ticket — candidates only for the queries in an entry's mask
    // and never rows whose map_ids word is 0xFFFFFFFF
    const TileMapEntry* map;
    const uint32_t* map_count;
    const uint32_t* map_ids;   // [rows] (padding rows of the list layout)
    uint32_t* map_thr;         // null, or [kQT * 32] band edges shared by the workgroups (ord32, 0 = none; a 128-byte line per
                               // query): see "shared band edges" in the kernel
    uint32_t* map_ticket;      // the deal's ticket counter (0 at launch; the last workgroup to leave zeroes it again)
    uint32_t* map_done;        // workgroups that have left (same)
#ifdef RAGK_STAMPS
    unsigned long long* stamps;  // experiment build: [grid][8] s_memrealtime stamps (100 MHz) of workgroup phases
#endif
};

#ifdef RAGK_STAMPS
#define RAGK_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RAGK_STAMP(i) do { } while (0)
#endif

// LDS layout: [Q fragments d8*128 B][keys 32*C*8 B][cnt 32 u32][thr 32 f32][flag 4 u32][margin 32 f32]
//             [query scale 32 f32][1 / (query scale * corpus scale) 32 f32][ticket, iteration u32 + pad]
__host__ __device__ inline size_t scan_lds_bytes(int d8, int C) {  // d8 = columns held in LDS (one chunk)
    return (size_t)d8 * 128 + (size_t)kQT * C * 8 + kQT * 4 + kQT * 4 + 16 + kQT * 4 + 2 * kQT * 4 + 16;  // + ticket word
}
// MAP mode: + the deal's item records (2 x 8 entries of two words) and their sequence word
__host__ __device__ inline size_t scan_lds_bytes_map(int d8, int C) { return scan_lds_bytes(d8, C) + 144; }

// NW = waves per workgroup (NW/4 per SIMD), E = buffer capacity / 64, D = register ring depth
// (steps of 32 bytes per row in flight per wave), L2 = metric.
// P = 0: the exact pass.  X is fp32, one step = 8 columns = 4 x v_mfma_f32_32x32x2_f32.
// P = 1: the screening pass of the two-stage search.  X is the scaled fp16 copy of the corpus, one
//        step = 16 columns = 1 x v_mfma_f32_32x32x16_f16 (same 32 bytes per row per step, so the
//        addressing is shared: the host passes row_stride and dc8 in 4-byte units).  Scores are
//        approximate; instead of the best k the buffers keep every row within `margin` of the
//        running k-th best, which is what makes the exact second stage a proof rather than a guess.
template <int NW, int E, int D, bool L2, int P, bool MAP = false>
__global__ __launch_bounds__(NW * 64) void scan_topk_kernel(const ScanParams p) {
    // the screening prologue reduces ||q||^2 and max|q| through 2 * NW = 16 partials per query
    static_assert(P != 1 || NW == 8, "the screening pass (P == 1) is written for 8 waves per workgroup");
    if (p.dyn_ctr_next && blockIdx.x == 0 && threadIdx.x == 0) *p.dyn_ctr_next = 0u;  // also by a switched-off launch
    if (p.enable && *p.enable != p.epoch) return;  // launch-uniform
    RAGK_STAMP(0);
    if (P == 1 && p.flag_clear && blockIdx.x == 0 && threadIdx.x == 0) *p.flag_clear = 0u;
    constexpr int C = 64 * E;
    constexpr int kScanWaves = NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int S = p.dc8 >> 3;

    f32x4* qf = reinterpret_cast<f32x4*>(smem);
    u64* keys = reinterpret_cast<u64*>(smem + (size_t)S * 1024);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(keys + (size_t)kQT * C);
    float* thr = reinterpret_cast<float*>(cnt + kQT);
    uint32_t* flag = reinterpret_cast<uint32_t*>(thr + kQT);
    float* mrg = reinterpret_cast<float*>(flag + 4);
    float* qsc_s = mrg + kQT;   // P == 1: per-query scales, computed in the prologue
    float* uns_s = qsc_s + kQT;
    // dynamic deal: tickw[0] = ticket, tickw[1] = the iteration it is for.  Written in that order by one thread and
    // read in the opposite order (LDS operations of a wave execute in order): a reader that sees the iteration
    // number sees its ticket
    volatile uint32_t* tickw = reinterpret_cast<volatile uint32_t*>(uns_s + kQT);

    // ---- tile walk.  Rounds 0 .. n_full-1: wave w of workgroup b takes tile (b NW + w) + round * grid * NW
    // (the 8 waves of a workgroup read 8 consecutive tiles).  The last, partial round hands its leftover
    // tiles out ONE PER WORKGROUP first (tile = base + w * grid + b): a leftover tile done by a lone wave
    // on an otherwise idle CU costs a third of a full round, and every CU gets one before any gets two.
    // (With whole 8-tile groups in the last round, 1.25M rows = 19.07 rounds ran as 20: 19 CUs streamed
    // a 20th group while 237 idled — 5 % of the scan.)
    //
    // Dynamic deal (p.dyn_ctr != null; long scans only).  CUs do not stream at the same rate: dealt statically,
    // the workgroups of a 1.25M-row scan finish 20-40 us apart (phase stamps, DESIGN "fixed cost") and the chip
    // waits for the slowest.  So only rounds 0 .. n_full-1 are static here and the rest of the corpus — two to
    // three rounds' worth — goes out by TICKET from one global counter: first whole NW-tile groups (the quantum
    // at which a CU streams at full rate), then n_singles single tiles, one per ticket, done by a lone wave (a
    // third of a group's time: they fill the gaps the groups leave).  A workgroup draws the ticket of iteration
    // it + 1 at the top of iteration it (thread 0; one returning atomic, a tile time before its result is needed)
    // and publishes it through an LDS word {iteration, ticket} that every wave reads before it prefetches the next
    // tile's first rows; a ticket beyond the last single tile ends the workgroup's walk — uniformly, so the
    // per-iteration barriers stay matched.  Which workgroup scans which tile does not change the result: the
    // merge takes the exact top-k of the union of the workgroups' lists.
    const long long last_row = p.n_rows - 1;
    const long long tile_rows = (long long)kTileRows * p.tile_step;  // distance between consecutive tiles' first rows
    const int tiles_per_iter = gridDim.x * kScanWaves;
    auto tile_of = [&](int it) -> int {
        return it < p.n_full ? (int)blockIdx.x * kScanWaves + wave + it * tiles_per_iter
                             : p.n_full * tiles_per_iter + wave * (int)gridDim.x + (int)blockIdx.x;
    };
    auto row_ptr = [&](int t) -> const float* {
#if defined(RAGK_ABLATE_L2_WINDOW)
        t &= 31;
#endif
        long long row = (long long)t * tile_rows + r;
        row = row < last_row ? row : last_row;
        return p.X + row * p.row_stride + p.col0 + 4 * h;
    };
    // The first ring loads go out in the middle of the prologue — after the query loads, before their LDS
    // stores.  Loads return in issue order: behind the ring's first touch of HBM (every CU at once) the
    // query fragments, all L2 hits, would wait for it; ahead of it they are stored while it is in flight.
    // MAP: the walk goes over the tile map.  Item g = entries [NW g, NW g + NW).  Iteration 0 of workgroup b takes item b
    // (every wave reads its own entry), iteration 1 item grid + b, every later one the item of a TICKET (item = ticket +
    // 2 grid).  Lanes 0 .. NW-1 of wave 0 run a three-deep pipeline at the TOP of each iteration — post the entries fetched an
    // iteration ago (for iteration it + 1) into LDS, fetch the entries of the ticket drawn an iteration ago (it + 2), draw the
    // next ticket (it + 3; lane 0) — so every returning memory operation has a whole tile time before its result is used
    // (ivf_kernels.hip.h, ivf_batch_scan_kernel: the same deal).  An entry past the end is "no tile".
    uint32_t map_n = 0u, qmask = 0xFFFFFFFFu, map_seen = 0u, map_gpend = 0u;
    TileMapEntry map_pend{(uint32_t)p.n_tiles, 0u};
    volatile uint32_t* mapw = tickw + 4;   // [2][NW][2] entries, then the newest iteration whose item is posted
    const bool dealer = MAP && wave == 0 && lane < kScanWaves;
    auto map_fetch = [&](uint32_t item, int w) -> TileMapEntry {   // entry w of `item`
        const uint32_t v = item * kScanWaves + (uint32_t)w;
        TileMapEntry e{(uint32_t)p.n_tiles, 0u};
        if (item < 0x08000000u && v < map_n) e = p.map[v];
        return e;
    };
    int tile;
    if constexpr (MAP) {
        map_n = *p.map_count;
        const TileMapEntry e0 = map_fetch(blockIdx.x, wave);
        tile = __builtin_amdgcn_readfirstlane((int)e0.tile);
        qmask = __builtin_amdgcn_readfirstlane(e0.mask);
        if (dealer) {
            map_pend = map_fetch(gridDim.x + blockIdx.x, lane);
            if (lane == 0) map_gpend = atomicAdd(p.map_ticket, 1u);
        }
    } else {
        tile = tile_of(0);
    }
    f32x4 xb[D];
    const float* pc = row_ptr(tile < p.n_tiles ? tile : p.n_tiles - 1);
    auto issue_ring = [&]() {
#pragma unroll
        for (int i = 0; i < D; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pc + 8 * i);
    };

    // per-query words of the filter state, requested first (wave 0)
    float pre_t0 = RAGK_SCORE_FLOOR;
    bool pre_seeded = false;
    if (tid < kQT && tid < p.nq) {
        if (p.thr_key) {
            const u64 tk = p.thr_key[tid];
            pre_seeded = tk != 0ull;
            if (pre_seeded) pre_t0 = unord32((uint32_t)(tk >> 32));
        }
        // L2 with ||q||^2 = inf / NaN: every distance is inf or NaN, which faiss's heap never admits
        // (a NaN filter: `score >= NaN` is false for every score, +inf included)
        if (L2 && !(p.qnorm[tid] <= 3.402823466e+38f)) pre_t0 = __builtin_nanf("");
    }

    // ---- prologue: queries -> MFMA B fragments in LDS.  A thread's lane (hence its query row and column
    // half) is the same in every trip, so its loads differ only in the step s: NF float4 loads are issued
    // back to back before the first LDS store (one fragment at a time the twelve dependent L2 round trips
    // of d = 768 were most of the kernel's fixed cost — every CU reads the same 96 KB at the same moment).
    // P == 0: fragment (s, l) = Q[l&31][col0 + 8s + 4(l>>5) .. +3], one float4.
    // P == 1: fragment (s, l) = Q[l&31][16s + 8(l>>5) .. +7] * qscale as fp16, two float4.
    {
        constexpr int FPS = P == 1 ? 2 : 1;         // float4 per fragment
        constexpr int PU = 12 / FPS;                // fragments per trip: 48 VGPRs in flight (d <= 768: one trip)
        constexpr int WPS = kScanWaves;             // steps covered per trip: one per wave
        constexpr int CPS = P == 1 ? 16 : 8;        // columns per step
        const int qrow = lane & 31, hh = lane >> 5;
        const bool have_q = qrow < p.nq;
        const float* qbase = p.Q + (size_t)(have_q ? qrow : 0) * p.d;
        const bool q_vec = (p.d & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Q) & 15) == 0;
        f32x4 v[PU][FPS];
        auto step_of = [&](int s0, int u) -> int {   // -1: past the end
            const int s = s0 + WPS * u;
            return s < S ? s : -1;
        };
        auto load_trip = [&](int s0) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = step_of(s0, u);
#pragma unroll
                for (int f = 0; f < FPS; ++f) {
                    v[u][f] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (s >= 0 && have_q) {
                        const int col = p.col0 + CPS * s + (CPS / 2) * hh + 4 * f;
                        if (q_vec && col + 3 < p.d) {  // rows 16-byte aligned: one load
                            v[u][f] = *reinterpret_cast<const f32x4*>(qbase + col);
                        } else {
                            f32x4 t = {0.f, 0.f, 0.f, 0.f};
                            if (col + 0 < p.d) t[0] = qbase[col + 0];
                            if (col + 1 < p.d) t[1] = qbase[col + 1];
                            if (col + 2 < p.d) t[2] = qbase[col + 2];
                            if (col + 3 < p.d) t[3] = qbase[col + 3];
                            v[u][f] = t;
                        }
                    }
                }
            }
        };
        auto store_trip = [&](int s0, float qsc) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = step_of(s0, u);
                if (s >= 0) {
                    if constexpr (P == 1) {
                        f16x8 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o[j] = (_Float16)(v[u][0][j] * qsc);
                            o[4 + j] = (_Float16)(v[u][FPS - 1][j] * qsc);
                        }
                        qf[s * 64 + lane] = __builtin_bit_cast(f32x4, o);
                    } else {
                        qf[s * 64 + lane] = v[u][0];
                    }
                }
            }
        };
        load_trip(wave);
        issue_ring();
        if constexpr (P == 1) {
            // ---- what a separate prep launch used to do (4 us + a dependent-launch boundary): per query
            // ||q||^2 and max |q_i| -> the power-of-two scale that puts the query high in the fp16 range,
            // and the worst-case error bound eps of an approximate score (flat_kernels "two-stage exact
            // search") -> margin = 2 eps.  Every workgroup derives them from the fragments it is loading
            // anyway — same instructions, same order, same bits in every workgroup — and workgroup 0
            // publishes the margins for the resolve kernel.  A thread holds its share of its query's row
            // in registers; the 16 threads of a row meet through 4 KB of the (still empty) key buffers.
            const bool one_trip = S <= WPS * PU;
            float ss = 0.f, am = 0.f;
            auto accumulate = [&]() {
#pragma unroll
                for (int u = 0; u < PU; ++u)
#pragma unroll
                    for (int f = 0; f < FPS; ++f)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = fabsf(v[u][f][j]);
                            am = (a <= 3.0e38f) ? fmaxf(am, a) : __builtin_inff();  // NaN / inf poison the maximum
                            ss = __builtin_fmaf(a, a, ss);
                        }
            };
            accumulate();
            for (int s0 = wave + WPS * PU; s0 < S; s0 += WPS * PU) {  // d > 768: the later trips are loaded twice
                load_trip(s0);
                accumulate();
            }
            float* part = reinterpret_cast<float*>(keys);
            part[(qrow * 16 + wave * 2 + hh) * 2 + 0] = ss;
            part[(qrow * 16 + wave * 2 + hh) * 2 + 1] = am;
            __syncthreads();
            if (tid < kQT) {
                float tss = 0.f, tam = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    tss += part[(tid * 16 + i) * 2 + 0];
                    tam = fmaxf(tam, part[(tid * 16 + i) * 2 + 1]);
                }
                float qscale = 0.f, unscale = 0.f, margin = 0.f;
                bool parked = false;
                if (tid < p.nq) {
                    // usable range: the scales and their product must stay ordinary fp32 numbers
                    const bool ok = tss <= 1.0e24f && (tam == 0.f || (tam >= 1.0e-12f && tam <= 1.0e12f));
                    if (!ok) {
                        parked = true;  // the fp32 fallback answers this query
                        if (blockIdx.x == 0) p.lossy[tid] = p.epoch;
                    } else {
                        int e = 0;
                        if (tam > 0.f) (void)frexpf(tam, &e);         // tam = m 2^e, m in [0.5, 1)
                        qscale = ldexpf(1.f, 14 - e);                   // |q| qscale < 2^14
                        unscale = 1.f / (qscale * p.x_scale);           // powers of two: exact
                        const float d64f = (float)(2 * p.dc8);
                        const float qn = sqrtf(tss) * 1.0001f;
                        const float rel = 0.0009775f + 6.f * d64f * 5.9604645e-8f;  // 2^-10 (1 + 2^-11) rounded up; 6 d 2^-24
                        const float eps = qn * p.x_normmax * rel +
                                          sqrtf(d64f) * (qn * p.x_absmax + p.x_normmax * tam) * 7.4505806e-9f;  // 2^-27
                        margin = 2.f * eps * 1.01f * (L2 ? 2.f : 1.f);
                    }
                    if (blockIdx.x == 0) p.margin_out[tid] = margin;
                }
                qsc_s[tid] = qscale;
                uns_s[tid] = unscale;
                mrg[tid] = margin;
                cnt[tid] = 0;
                // unused query columns (nq < 32) score 0 against every row: park their threshold at +inf so
                // they never enter the slow path (left at -inf they tie forever and double the scan time)
                float t0 = pre_t0;
                if (pre_seeded) t0 -= margin;  // the band below an approximate score that k rows reach
                thr[tid] = tid < p.nq ? (parked ? __builtin_inff() : t0) : __builtin_inff();
            }
            __syncthreads();  // scales visible; `part` (the key buffers) free again
            const float qsc = qsc_s[qrow];
            if (!one_trip) load_trip(wave);
            store_trip(wave, qsc);
            for (int s0 = wave + WPS * PU; s0 < S; s0 += WPS * PU) {
                load_trip(s0);
                store_trip(s0, qsc);
            }
        } else {
            store_trip(wave, 0.f);
            for (int s0 = wave + WPS * PU; s0 < S; s0 += WPS * PU) {
                load_trip(s0);
                store_trip(s0, 0.f);
            }
            if (tid < kQT) {
                cnt[tid] = 0;
                thr[tid] = tid < p.nq ? pre_t0 : __builtin_inff();  // unused columns parked (see above)
            }
        }
    }
    if (tid < 4) flag[tid] = 0;
    if (tid == 0) tickw[1] = 0u;  // no iteration has this number: tickets are for iterations >= n_full >= 4
    if (MAP && tid == 0) mapw[4 * kScanWaves] = 0u;   // (items are posted for iterations >= 1)
    __syncthreads();
    RAGK_STAMP(1);

    const u64 key_ceil = p.ceil ? p.ceil[r] : ~0ull;
    // Warm start (k <= 16, 8-wave build, no sampled threshold): see the filter below.
    const bool warm = kScanWaves == 8 && p.k <= 16 && !p.acc_out && p.thr_key == nullptr;

    // Overflow flags: pass number seq (one pass = one trip through the barrier loop below) owns
    // flag[seq & 3]; appends raise the flag of the pass that will check them, and pass seq clears
    // the slot of pass seq + 2, so a wave that has already left a pass never races a wave that is
    // still reading that pass's flag.
    uint32_t seq = 0;
    const bool dyn = p.dyn_ctr != nullptr;  // kernel-uniform; then n_iters == n_full: the static rounds
    bool more = MAP ? (uint32_t)blockIdx.x * kScanWaves < map_n : p.n_iters > 0;   // workgroup-uniform: iteration `it` exists
    for (int it = 0; more; ++it) {
        const bool active = tile < p.n_tiles;  // wave-uniform
        uint32_t qmask_n = 0u;
        if constexpr (MAP) {
            if (dealer) {   // the deal (see the walk above)
                volatile uint32_t* w = mapw + 2 * kScanWaves * ((it + 1) & 1);
                w[2 * lane] = map_pend.tile;
                w[2 * lane + 1] = map_pend.mask;
                if (lane == 0) mapw[4 * kScanWaves] = (uint32_t)(it + 1);   // (a wave's LDS writes complete in order)
                const uint32_t g = __builtin_amdgcn_readfirstlane(map_gpend);
                map_pend = map_fetch(g + 2u * gridDim.x, lane);
                if (lane == 0) map_gpend = atomicAdd(p.map_ticket, 1u);
            }
            // Shared band edges.  A workgroup meets a given query in one or two of its items, so a filter that learns only
            // from its own rows stays cold (sort after sort).  The lower edge of a band — a k-th best approximate score
            // minus the margin — established by ANY workgroup bounds the final band from below (that workgroup's k-th
            // best is at most the global one), so it is published (atomic max, one line per query) and wave 0 refreshes
            // the local filters from the published values once per iteration, the load consumed an iteration later.
            if (P == 1 && p.map_thr && wave == 0 && lane < kQT) {
                if (map_seen) {
                    const float tg = unord32(map_seen);
                    if (tg > thr[lane]) thr[lane] = tg;   // parked (+inf) and NaN filters stay
                }
                map_seen = __hip_atomic_load(p.map_thr + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (dyn && it + 1 >= p.n_iters && tid == 0) {  // the ticket of iteration it + 1
            const uint32_t g = atomicAdd(p.dyn_ctr, 1u);
            tickw[0] = g;
            tickw[1] = (uint32_t)(it + 1);
        }

        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (p.acc_in && active) {  // continue the fmaf chains of the previous column chunk
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = p.acc_io[((size_t)tile * kTileRows + (i & 3) + 8 * (i >> 2) + 4 * h) * kQT + r];
        }

        // One step = 8 columns = 4 MFMAs on ring slot i.  The ring is refilled G slots at a time,
        // D steps ahead: with G = 4 the four loads that make up one 128-byte line of each of the
        // wave's 32 rows are issued back to back, which is what lets the vector L1 fetch each
        // line once (measured: spread one per step the same stream runs at half the rate).
        // sched_group_barrier pins {next query fragment, 4 MFMA} per step and the refills at the
        // end of their group; left alone hipcc sinks all refills to the end of the unrolled body.
        constexpr int G = D >= 4 ? 4 : D;
        f32x4 qcur = qf[lane];
        int s0 = 0;
        for (; s0 < S - D; s0 += D) {
            const float* src = pc + 8 * (s0 + D);
            const f32x4* qs = qf + (size_t)(s0 + 1) * 64 + lane;
#pragma unroll
            for (int g = 0; g < D; g += G) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int i = g + j;
                    const f32x4 qn = qs[i * 64];
                    acc = scan_step<P>(xb[i], qcur, acc);
                    qcur = qn;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // next query fragment (DS read)
                    __builtin_amdgcn_sched_group_barrier(0x008, P == 1 ? 1 : 4, 0);  // this step's MFMA
                }
#pragma unroll
                for (int j = 0; j < G; ++j) xb[g + j] = *reinterpret_cast<const f32x4*>(src + 8 * (g + j));
                __builtin_amdgcn_sched_group_barrier(0x020, G, 0);  // ring refill (VMEM reads)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile: static rounds by formula, later ones by the ticket drawn at the top of this iteration
        int tnext = p.n_tiles;
        bool more_next;
        if constexpr (MAP) {
            while (mapw[4 * kScanWaves] != (uint32_t)(it + 1)) {
            }
            volatile uint32_t* w = mapw + 2 * kScanWaves * ((it + 1) & 1);
            tnext = __builtin_amdgcn_readfirstlane((int)w[2 * wave]);
            qmask_n = __builtin_amdgcn_readfirstlane(w[2 * wave + 1]);
            more_next = __builtin_amdgcn_readfirstlane((int)w[0]) < p.n_tiles;
        } else if (!dyn || it + 1 < p.n_iters) {
            more_next = it + 1 < p.n_iters;
            if (more_next) tnext = tile_of(it + 1);
        } else {
            while (tickw[1] != (uint32_t)(it + 1)) {
            }
            const int g = __builtin_amdgcn_readfirstlane((int)tickw[0]);
            more_next = g < p.n_dyn_groups + p.n_singles;
            if (g < p.n_dyn_groups)
                tnext = p.dyn_tile0 + g * kScanWaves + wave;
            else if (more_next && wave == 0)
                tnext = p.dyn_tile0 + p.n_dyn_groups * kScanWaves + (g - p.n_dyn_groups);
        }
        const float* pn = row_ptr(tnext < p.n_tiles ? tnext : p.n_tiles - 1);
        // last D steps: refill the ring from the next tile
        {
            const f32x4* qs = qf + (size_t)(s0 + 1) * 64 + lane;
#pragma unroll
            for (int g = 0; g < D; g += G) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int i = g + j;
                    // the fragment after the last step is never used; stay inside the Q image
                    const f32x4 qn = qs[(i + 1 < D ? i : -1 - s0) * 64];
                    acc = scan_step<P>(xb[i], qcur, acc);
                    qcur = qn;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, P == 1 ? 1 : 4, 0);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) xb[g + j] = *reinterpret_cast<const f32x4*>(pn + 8 * (g + j));
                __builtin_amdgcn_sched_group_barrier(0x020, G, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pc = pn;

        if (p.acc_out) {  // not the last column chunk: park the partial sums (kernel-uniform branch)
            if (active) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    p.acc_io[((size_t)tile * kTileRows + (i & 3) + 8 * (i >> 2) + 4 * h) * kQT + r] = acc[i];
            }
            tile = tnext;
            more = more_next;
            continue;
        }

        // ---- ranking scores.  Lane (r, h): query r, tile rows (i&3) + 8(i>>2) + 4h.
        const long long row0 = (long long)tile * tile_rows;
        float sc[16];
        if (P == 1) {  // undo the power-of-two scales (exact)
            const float uns = uns_s[r];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= uns;
        }
        if (L2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 xn = {0.f, 0.f, 0.f, 0.f};
                const long long rb = row0 + 8 * g + 4 * h;
                if (active) {
                    if (rb + 3 <= last_row) {
                        xn = *reinterpret_cast<const f32x4*>(p.xnorm + rb);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (rb + j <= last_row) xn[j] = p.xnorm[rb + j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[4 * g + j] = __builtin_fmaf(2.0f, acc[4 * g + j], -xn[j]) + 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = acc[i] + 0.0f;
        }

        // ---- warm start.  A cold filter lets the first round's 256 rows per query into a 64-slot buffer:
        // overflow, sort, retry — and again for the next three or four rounds, with the whole workgroup
        // (and, since every workgroup starts together, the whole chip) parked at each sort: 40 us of a
        // 0.71 ms scan of 1.25M x 768.  Instead, each lane posts the best eligible score of its 16 rows;
        // the 16 half-tile maxima of a query come from 16 disjoint row sets, so at least k distinct rows
        // reach the k-th largest of them and the filter may start there.  About 15 of the first 256 rows
        // then pass and the first sort comes four rounds later, with a threshold from ~1300 rows.
        if (warm && it == 0) {  // kernel-uniform
            float* sub = reinterpret_cast<float*>(keys);  // the candidate buffers are still empty
            float m = -__builtin_inff();
            if (active && (!MAP || ((qmask >> r) & 1u))) {   // (MAP: only tiles of lists the query probes ...)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    bool real = row <= last_row && make_key(sc[i], (uint32_t)row) < key_ceil;
                    // (... and never a list layout's padding row: the zero vector has the best possible L2 score)
                    if constexpr (MAP) real = real && p.map_ids[row] != 0xFFFFFFFFu;
                    if (real) m = fmaxf(m, sc[i]);  // NaN never wins
                }
            }
            sub[r * 16 + wave * 2 + h] = m;
            __syncthreads();
            {
                const int q = tid >> 4, j = tid & 15;
                const float v = sub[q * 16 + j];
                int rank = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float o = sub[q * 16 + i];
                    rank += (o > v || (o == v && i < j)) ? 1 : 0;
                }
                if (rank == p.k - 1) {
                    const float cand = P == 1 ? v - mrg[q] : v;
                    if (cand > thr[q]) thr[q] = cand;  // parked (+inf) and lossy queries stay parked
                }
            }
            __syncthreads();
        }

        // ---- filter + append
        uint32_t pending = 0;
        if (active && (!MAP || ((qmask >> r) & 1u))) {
            float t = thr[r];
            float m = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, sc[i]);
            if (m >= t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    bool take = sc[i] >= t && row <= last_row && make_key(sc[i], (uint32_t)row) < key_ceil;
                    if constexpr (MAP) take = take && p.map_ids[row] != 0xFFFFFFFFu;
                    if (take) {
                        const uint32_t slot = atomicAdd(&cnt[r], 1u);
                        if (slot < (uint32_t)C) {
                            keys[(size_t)r * C + slot] = make_key(sc[i], (uint32_t)row);
                        } else {
                            pending |= 1u << i;
                            flag[seq & 3] = 1;
                        }
                    }
                }
            }
        }

        // ---- overflow handling: synchronised compaction of every query, then retry
        for (;;) {
            __syncthreads();
            const bool need = flag[seq & 3] != 0;  // workgroup-uniform
            if (tid == 0) flag[(seq + 2) & 3] = 0;
            ++seq;
            if (!need) break;
            {   // this wave's queries (wave, wave + NW, ...) go through the sort network together
                constexpr int NQW = (kQT + kScanWaves - 1) / kScanWaves;
                u64 kk[NQW][E];
                uint32_t nn[NQW];
#pragma unroll
                for (int j = 0; j < NQW; ++j) {
                    const int q = wave + kScanWaves * j;
                    nn[j] = q < kQT ? min(cnt[q], (uint32_t)C) : 0u;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t idx = e * 64 + lane;
                        kk[j][e] = idx < nn[j] ? keys[(size_t)q * C + idx] : 0ull;
                    }
                }
                wave_sort_desc<E, NQW>(kk, lane);
#pragma unroll
                for (int j = 0; j < NQW; ++j) {
                    const int q = wave + kScanWaves * j;
                    if (P == 1) {
                        // keep the band: every key whose score is within mrg[q] of the k-th best so far
                        if (q < kQT && nn[j] > 0) {  // wave-uniform
                            u64 kth = 0ull;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const u64 c = ((u64)__shfl((uint32_t)(kk[j][e] >> 32), (p.k - 1) & 63, 64)) << 32;
                                if (e == ((p.k - 1) >> 6)) kth = c;
                            }
                            const bool full = nn[j] >= (uint32_t)p.k;
                            const float kscore = unord32((uint32_t)(kth >> 32));
                            // never below the current filter (warm start / sampled start / earlier sorts):
                            // every kept key already passed it, and a query sorted along with the others
                            // before it holds k rows must not fall back to an open filter
                            float tnew = full ? fmaxf(kscore - mrg[q], thr[q]) : thr[q];
                            if constexpr (MAP) {   // publish a band edge (never one without its margin: -0.0f marks a given-up band)
                                if (full && lane == 0 && p.map_thr && !__builtin_signbit(mrg[q]) && kscore - mrg[q] > thr[q])
                                    atomicMax(p.map_thr + q * 32, ord32(kscore - mrg[q]));
                            }
                            uint32_t keep = 0;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const uint32_t idx = e * 64 + lane;
                                const bool kp = idx < nn[j] && unord32((uint32_t)(kk[j][e] >> 32)) >= tnew;
                                keep += (uint32_t)__popcll(__ballot(kp));
                            }
                            if (keep > (uint32_t)(C - 16)) {
                                // The band does not fit: this workgroup keeps its best k for this query from here
                                // on and says so.  Every row it drops scores below its k-th best of the moment,
                                // hence at most its FINAL k-th best — entry k-1 of the list it emits — so the
                                // resolve kernel voids the certificate only if that entry lies inside the final
                                // band.  (A cluster of near-duplicates far below the top k overflows the band of
                                // the one workgroup that scans it early, while its running k-th best is still low;
                                // voiding the query for that sent 19 of 32 ordinary queries to the fallback on a
                                // corpus with 400 copies of one row.)
                                if (lane == 0) {
                                    mrg[q] = MAP ? -0.0f : 0.f;   // (the same arithmetic; MAP: "this band was given up")
                                    if (p.wg_lossy) p.wg_lossy[(size_t)q * gridDim.x + blockIdx.x] = p.epoch;
                                }
                                tnew = kscore;  // keep > C - 16 >= k implies full
                                keep = (uint32_t)p.k;
                            }
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const uint32_t idx = e * 64 + lane;
                                if (idx < keep) keys[(size_t)q * C + idx] = kk[j][e];
                            }
                            if (lane == 0) {
                                thr[q] = tnew;
                                cnt[q] = keep;
                            }
                        }
                    } else if (q < kQT && nn[j] > 0) {  // sorting a short buffer is harmless: cnt and thr stay consistent
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int idx = e * 64 + lane;
                            if (idx < p.k) keys[(size_t)q * C + idx] = kk[j][e];
                            if (idx == p.k - 1)
                                if (nn[j] >= (uint32_t)p.k) thr[q] = unord32((uint32_t)(kk[j][e] >> 32));  // else: keep the filter
                        }
                        if (lane == 0) cnt[q] = min(nn[j], (uint32_t)p.k);
                    }
                }
            }
            __syncthreads();
            if (pending) {
                const float t = thr[r];
                uint32_t still = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if ((pending >> i) & 1u) {
                        if (sc[i] >= t) {
                            const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            const uint32_t slot = atomicAdd(&cnt[r], 1u);
                            if (slot < (uint32_t)C) {
                                keys[(size_t)r * C + slot] = make_key(sc[i], (uint32_t)row);
                            } else {
                                still |= 1u << i;
                                flag[seq & 3] = 1;
                            }
                        }
                    }
                }
                pending = still;
            }
        }
        if (it == 0) RAGK_STAMP(2);
        tile = tnext;
        more = more_next;
        if constexpr (MAP) qmask = qmask_n;
    }

    if (p.acc_out) return;
    RAGK_STAMP(3);
    // ---- epilogue: sort every buffer, emit k keys per query for this workgroup
    {
        constexpr int NQW = (kQT + kScanWaves - 1) / kScanWaves;
        u64 kk[NQW][E];
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = wave + kScanWaves * j;
            const uint32_t n = q < kQT ? min(cnt[q], (uint32_t)C) : 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t idx = e * 64 + lane;
                kk[j][e] = idx < n ? keys[(size_t)q * C + idx] : 0ull;
            }
        }
        wave_sort_desc<E, NQW>(kk, lane);
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = wave + kScanWaves * j;
            if (q < kQT) {
                const int kout = P == 1 ? p.kout : p.k;
                u64* out = p.partial + ((size_t)q * gridDim.x + blockIdx.x) * kout;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = e * 64 + lane;
                    if (idx < kout) out[idx] = kk[j][e];
                }
                if (P == 1) {  // kout may exceed the buffer: pad the list with empty keys
                    for (int idx = C + lane; idx < kout; idx += 64) out[idx] = 0ull;
                }
            }
        }
    }
    RAGK_STAMP(4);
    if constexpr (MAP) {   // the last workgroup to leave puts the ticket counter back (launches of an index are stream-ordered)
        __syncthreads();
        if (tid == 0 && atomicAdd(p.map_done, 1u) == gridDim.x - 1) {
            *p.map_ticket = 0u;
            *p.map_done = 0u;
        }
    }
}

// ---- sample pass -> starting thresholds ----------------------------------------------------------
// A cold scan starts every filter at -inf: each workgroup fills and compacts its buffers several times
// before its thresholds bite (measured on 1.25M x 768: 40 us of a 0.71 ms scan at k = 10, 240 us of 0.91
// ms at k = 100).  For larger k a sample pass first runs the scan kernel with k = 1 over 2048 tiles
// spread evenly through the corpus (one per wave); this kernel then takes, per query, the k-th largest
// of the n_lists workgroup maxima.  At least k distinct rows reach that score, so the final k-th best
// does too, and the real scan may start its filter there (ScanParams.thr_key) without losing a row.
__global__ __launch_bounds__(64) void sample_threshold_kernel(const u64* heads, int n_lists, int k, u64* thr_key) {
    const int q = blockIdx.x, lane = threadIdx.x;
    u64 kk[1][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int l = e * 64 + lane;
        kk[0][e] = l < n_lists ? heads[(size_t)q * n_lists + l] : 0ull;
    }
    wave_sort_desc<4, 1>(kk, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (e * 64 + lane == k - 1) thr_key[q] = kk[0][e];  // 0 when fewer than k lists hold a row
}

// ---- K2: merge ------------------------------------------------------------------------------
// Top-k of n_lists lists that are each sorted best-first (the per-workgroup lists K1 emits, or the
// per-shard lists after the all-gather).  One 256-thread workgroup per query plays a k-round
// tournament: every thread holds the head of the list(s) it owns, a round is one workgroup-wide
// 64-bit max (DPP reduction + 4 LDS words + one barrier), the winner advances its list.  A round is a
// serial chain of wave-wide instructions, ~0.4 us: k = 10 over 256 lists takes ~9 us with the staging
// and the launch — the bitonic merge this replaces sorted 4096 keys to keep 10 and took 90.
constexpr int kMergeMaxOwned = 8;  // lists per thread: n_lists <= 2048

struct KeyListSrc {  // lists of ranking keys: keys[(q * n_lists + l) * k + pos]
    const u64* keys;
    int n_lists, k;
    __device__ __forceinline__ u64 get(int q, int l, int pos) const {
        return keys[((size_t)q * n_lists + l) * k + pos];
    }
};

struct ShardListSrc {  // per-shard (score, id) results: shard l's element (q * k + pos) of two arrays
    const float* scores;
    const long long* ids;
    long long score_stride;  // floats between consecutive shards' score blocks
    long long id_stride;     // int64s between consecutive shards' id blocks
    int k, metric;
    __device__ __forceinline__ u64 get(int q, int l, int pos) const {
        const size_t e = (size_t)q * k + pos;
        const long long id = ids[(size_t)l * id_stride + e];
        const float s = scores[(size_t)l * score_stride + e];  // both loads in flight together: the buffer has just
                                                               // arrived from other GPUs and is cold in this L2
        if (id < 0 || id > 0xFFFFFFFEll) return 0ull;  // padding; ids beyond 32 bits cannot come from rag_index (set_id_offset refuses them)
        return make_key(metric ? -s : s, (uint32_t)id);  // L2 lists carry ascending distances
    }
};

struct MergeOut {
    float* scores;       // [nq][out_stride], this call fills columns 0..k-1
    long long* ids;      // [nq][out_stride]
    u64* last_key;       // [nq] or null: the k-th key of this call (0 if the lists ran dry)
    int out_stride;
    const float* qnorm;  // canonical ||q||^2 (index-side L2 only)
    long long id_offset;
    int metric;          // 0 IP, 1 L2
    int from_shards;     // keys carry (-distance, id) of already finished results
    const uint32_t* qmask;  // null, or [nq] words: only queries whose word equals `epoch` are written (fallback path)
    uint32_t epoch;
    // shard merge only: every shard's block of the gathered buffer carries one "my result is not final" word
    // (RAG_SEARCH_DEFER_FALLBACK); query 0's workgroup ORs them into *any_flag (null: no flags)
    const uint32_t* flags;   // word of shard 0
    long long flag_stride;   // uint32 words between consecutive shards' flag words
    int n_flags;
    uint32_t* any_flag;
    // optional mirror of the result in pinned HOST memory (same [nq][out_stride] layout + the OR-ed flag word): the
    // merge writes it itself — posted stores, a few KB — so the batch needs no read-back copy behind the kernel
    float* scores_host;
    long long* ids_host;
    uint32_t* any_flag_host;
};

// Wave-wide 64-bit max on the DPP cross-lane paths (quad permutes, row mirrors, row broadcasts): a
// few cycles per step where a ds_bpermute shuffle costs an LDS round trip — the tournament below runs
// one of these per round, and with shuffles that was most of a round's 0.8 us.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {  // the result is wave-uniform
    v = umax64(v, dpp_u64<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
    v = umax64(v, dpp_u64<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
    v = umax64(v, dpp_u64<0x141, 0xF>(v));  // row_half_mirror
    v = umax64(v, dpp_u64<0x140, 0xF>(v));  // row_mirror: every lane of a row holds the row's max
    v = umax64(v, dpp_u64<0x142, 0xA>(v));  // row_bcast15 into rows 1 and 3
    v = umax64(v, dpp_u64<0x143, 0xC>(v));  // row_bcast31 into rows 2 and 3: lane 63 holds the wave's max
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
}

// The first `look` keys of every list are staged in LDS up front (coalesced), so a round never waits
// for global memory: the winner's list advances with one LDS read.  (With the look-ahead in
// registers, loaded when a list advanced, the compiler put the load's full latency on every round:
// 0.7 us per round, 74 us for k = 100.)  Lists that contribute more than `look` keys read on from
// global memory.
__host__ __device__ inline int merge_look(int n_lists, int list_len, int k) {
    const int nl = n_lists > 0 ? n_lists : 1;
    const int fit = 6144 / nl;                 // 48 KiB of keys
    int want = 2 * ((k + nl - 1) / nl) + 4;    // a list's expected share of the k winners, with slack
    if (want > fit) want = fit < 2 ? 2 : fit;
    return want < list_len ? want : list_len;
}

// Staging of the look-ahead: NT threads, eight loads in flight each (one at a time, the six dependent
// round trips of 256 lists x 6 keys were a third of the kernel).
template <class Src, int NT>
__device__ __forceinline__ void stage_heads(const Src& src, int q, int n_lists, int look, u64* ahead, int t) {
    const int total = n_lists * look;
    for (int base = 0; base < total; base += NT * 8) {
        u64 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + t + NT * u;
            const int l = idx / look;
            v[u] = idx < total ? src.get(q, l, idx - l * look) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + t + NT * u;
            if (idx < total) ahead[idx] = v[u];
        }
    }
}

// The rounds.  WAVES x 64 threads take part, thread t owns lists t, t + 64 WAVES, ... (OWN of them:
// n_lists <= 64 WAVES OWN).  A round is a serial chain — local max, wave-wide max, (WAVES > 1: four LDS
// words and a barrier,) the winner advances — so everything a round does not need is kept out of it:
// with up to 256 lists ONE wave plays (four lists per lane, no LDS, no barrier: 0.13 us per round against
// 0.45 with four waves).  on_round(round, best key of the round) runs on every participating thread;
// STOP_EMPTY ends the tournament when the lists run dry.  Returns the last round's key.
template <class Src, int OWN, int WAVES, bool STOP_EMPTY, class OnRound>
__device__ __forceinline__ u64 tournament_rounds(const Src& src, int q, int n_lists, int rounds, int list_len, int look,
                                                 const u64* ahead, u64 (*wmax)[4], int t, OnRound on_round) {
    constexpr int NT = 64 * WAVES;
    u64 head[OWN];
    int pos[OWN];
#pragma unroll
    for (int j = 0; j < OWN; ++j) {
        const int l = t + NT * j;
        pos[j] = 0;
        head[j] = (l < n_lists && rounds > 0) ? ahead[(size_t)l * look] : 0ull;
    }
    u64 bm = 0ull;
    for (int round = 0; round < rounds; ++round) {
        u64 best = head[0];
#pragma unroll
        for (int j = 1; j < OWN; ++j) best = umax64(best, head[j]);
        const u64 wm = wave_max_u64(best);
        if constexpr (WAVES == 1) {
            bm = wm;
        } else {
            if ((t & 63) == 0) wmax[round & 1][t >> 6] = wm;
            __syncthreads();  // the other parity is free again: its readers passed this barrier's predecessor
            bm = umax64(umax64(wmax[round & 1][0], wmax[round & 1][1]), umax64(wmax[round & 1][2], wmax[round & 1][3]));
        }
        on_round(round, bm);
        if (bm == 0ull) {
            if (STOP_EMPTY) break;  // uniform over the participants
            continue;
        }
        if (best == bm) {  // keys are unique, so exactly one thread advances one list
#pragma unroll
            for (int j = 0; j < OWN; ++j) {
                if (head[j] == bm) {
                    const int l = t + NT * j;
                    const int np = ++pos[j];
                    head[j] = np < look ? ahead[(size_t)l * look + np] : (np < list_len ? src.get(q, l, np) : 0ull);
                }
            }
        }
    }
    return bm;
}

template <class Src, int OWN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void tournament_merge_kernel(const Src src, const int n_lists, const int k,
                                                                      const int look, const MergeOut out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* ahead = reinterpret_cast<u64*>(smem);  // [n_lists][look]
    __shared__ u64 wmax[2][4];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (out.qmask && out.qmask[q] != out.epoch) return;  // workgroup-uniform
    if (out.any_flag && q == 0 && tid == 0) {
        uint32_t any = 0u;
        for (int g = 0; g < out.n_flags; ++g) any |= out.flags[(size_t)g * out.flag_stride];
        *out.any_flag = any;
        if (out.any_flag_host) *out.any_flag_host = any;
    }
    stage_heads<Src, 64 * WAVES>(src, q, n_lists, look, ahead, tid);
    __syncthreads();
    tournament_rounds<Src, OWN, WAVES, false>(src, q, n_lists, k, k, look, ahead, wmax, tid, [&](int round, u64 bm) {
        if (tid != 0) return;
        float s;
        long long id;
        if (bm == 0ull) {
            s = out.metric ? 3.402823466e+38f : -3.402823466e+38f;
            id = -1;
        } else {
            const float rs = unord32((uint32_t)(bm >> 32));
            if (out.from_shards) {
                s = out.metric ? -rs : rs;
            } else if (out.metric) {
                const float dist = out.qnorm[q] - rs;
                s = dist < 0.f ? 0.f : dist;
            } else {
                s = rs;
            }
            id = (long long)(0xFFFFFFFFu - (uint32_t)(bm & 0xFFFFFFFFull)) + out.id_offset;
        }
        out.scores[(size_t)q * out.out_stride + round] = s;
        out.ids[(size_t)q * out.out_stride + round] = id;
        if (out.scores_host) {
            out.scores_host[(size_t)q * out.out_stride + round] = s;
            out.ids_host[(size_t)q * out.out_stride + round] = id;
        }
        if (out.last_key && round == k - 1) out.last_key[q] = bm;
    });
}

// ---- two-stage exact search: screening copy, certificate, exact second stage ---------------------
//
// Stage 1 (scan_topk_kernel<P = 1>) reads a scaled fp16 copy of the corpus — half the bytes of the
// fp32 scan, and the matrix work drops to 1/16 — and keeps, per query and workgroup, every row whose
// *approximate* score lies within 2 eps of the workgroup's running k-th best.  screen_resolve_kernel
// then gathers the rows within 2 eps of the global k-th best approximate score (at most 239 of them),
// recomputes their canonical fp32 scores and ranks them.  With |approx - exact| <= eps
// for every row, the k rows with the best approximate scores have exact scores >= a_k - eps, so the
// k-th exact score is >= a_k - eps, so every true top-k row has approx >= a_k - 2 eps: it is in the
// band.  The result is therefore the exact top-k (bit-identical to the one-pass fp32 search) whenever
// the whole band was kept — the certificate.  eps is a worst-case bound, not an estimate:
//   fp16 rounding of both operands (values pre-scaled by powers of two so that neither overflow nor
//   flushed subnormals matter)          (2^-10 + 2^-22) sum|q_i x_i|  <=  ... ||q|| ||x||
//   anything below the fp16 normal range treated as lost   sqrt(d) (||q|| amax_x + ||x|| amax_q) 2^-27
//   fp32 accumulation in the matrix core, in any order, and the canonical chain itself
//                                        6 d 2^-24 ||q|| ||x||
// Queries whose certificate fails (dense near-ties: a band wider than the buffers; non-finite or
// out-of-range values) are re-run through the fp32 scan: the fallback launches are always enqueued
// and switch themselves off on the device when no query needs them.

struct ScreenCorpusStats {  // device words updated with atomics while rows are added
    uint32_t absmax_bits;   // max |x_ij| as float bits
    uint32_t sqnorm_bits;   // max ||x_i||^2 as float bits
    uint32_t bad;           // a non-finite element was seen
    uint32_t pad;
};

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// One wave per row, grid-stride; the running maxima stay in registers until the end.
__global__ __launch_bounds__(256) void screen_stats_kernel(const float* X, long long row_stride, int d8, long long row0,
                                                           long long n, ScreenCorpusStats* st) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    float amax = 0.f, nmax = 0.f;
    bool bad = false;
    for (long long i = wave; i < n; i += n_waves) {
        const f32x4* src = reinterpret_cast<const f32x4*>(X + (row0 + i) * row_stride);
        float ss = 0.f;
        for (int c = lane; c < d8 / 4; c += 64) {
            const f32x4 v = src[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = fabsf(v[j]);
                bad |= !(a <= 3.0e38f);
                amax = fmaxf(amax, a);
                ss = __builtin_fmaf(a, a, ss);
            }
        }
        ss = wave_sum_f32(ss);
        bad |= !(ss <= 3.0e38f);
        nmax = fmaxf(nmax, ss);
    }
    amax = wave_max_f32(amax);
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) {
        atomicMax(&st->absmax_bits, __float_as_uint(amax));  // non-negative floats order like their bits
        atomicMax(&st->sqnorm_bits, __float_as_uint(nmax));
        if (any_bad) atomicOr(&st->bad, 1u);
    }
}

// X (fp32, stride d8) -> X16 (fp16, stride d64 halves, zero padded), values multiplied by `scale`.
__global__ __launch_bounds__(256) void screen_convert_kernel(const float* X, long long row_stride, int d8, long long row0,
                                                             long long n, float scale, _Float16* X16, int d64) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long i = wave; i < n; i += n_waves) {
        const f32x4* src = reinterpret_cast<const f32x4*>(X + (row0 + i) * row_stride);
        f16x8* dst = reinterpret_cast<f16x8*>(X16 + (row0 + i) * (long long)d64);
        for (int c = lane; c < d64 / 8; c += 64) {
            f16x8 o;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (8 * c + 4 * hh < d8) v = src[2 * c + hh];  // d8 is a multiple of 4
#pragma unroll
                for (int j = 0; j < 4; ++j) o[4 * hh + j] = (_Float16)(v[j] * scale);
            }
            dst[c] = o;
        }
    }
}

struct ScreenQueryState {   // per pass of <= 32 queries (device memory).  The flag words are EPOCH-valued: a flag
                            // is set for the search whose epoch it holds, so no launch has to clear them.
    float margin[kQT];      // 2 eps (4 eps for L2 ranking scores); written by workgroup 0 of the screening scan
    uint32_t lossy[kQT];    // certificate void: out-of-range query, or a workgroup dropped part of the band
    uint32_t sample_lossy[kQT];  // scratch for the sample pass (its lists only seed thresholds)
    uint32_t fallback[kQT]; // set by the resolve kernel: this query goes through the fp32 scan
    uint32_t any_fallback;  // the fallback launches read this word
    uint32_t pad[3];
};

struct ScreenCounters {     // cumulative, read back by rag_index_screen_stats
    unsigned long long queries;
    unsigned long long fallbacks;
    uint32_t max_err_ratio_bits;  // max observed |approx - exact| / eps over all verified candidates
    uint32_t pad;
};

// Resolve: what turns the screening pass's lists into results — ONE launch, one 256-thread workgroup per
// query (it was three: collect, verify, finalize, 27 us of kernels plus two dependent-launch boundaries
// of a 0.44 ms shard step; everything between the phases now stays in LDS).
//   (A) the k-th best approximate key over all workgroup lists, by the same tournament as the merge
//       kernel, k rounds;
//   (B) every listed row whose approximate score is within the query's margin of it — the band — is a
//       candidate.  The certificate needs the WHOLE band: more than kp - 1 band rows, a workgroup list that
//       lies entirely inside the band (it may have been cut at kout), or a workgroup that gave up its band
//       for this query AND still has k rows inside the final band, void it;
//   (C) canonical fp32 score of every candidate (the rago_dot chain of oracle/flat_oracle.c).  Each wave
//       stages RPW candidate rows in LDS with coalesced 16-byte loads, all in flight together, then RPW
//       lanes run the RPW fmaf chains side by side out of LDS; four waves cover 32 (16) candidates per
//       pass, which is one pass for the usual dozen or two;
//   (D) wave 0 checks the certificate, ranks the exact keys and writes (score, id); a query whose
//       certificate failed raises the fallback words the fp32 scan launches read.
struct ResolveParams {
    KeyListSrc src;
    int n_lists, k, kp, look;
    const float* X;          // fp32 corpus
    long long row_stride;
    const float* xnorm;      // canonical squared norms (L2)
    const float* Q;          // this pass's queries, [nq][d]
    int d, d8, l2;
    const float* qnorm;      // canonical ||q||^2 (L2)
    long long id_offset;
    ScreenQueryState* qs;
    const uint32_t* wg_lossy;  // [nq][n_lists] = epoch where a workgroup gave up part of a query's band
    uint32_t epoch;          // this search's id: the value "set" flags hold
    ScreenCounters* ctr;
    float* out_s;            // [nq][k]
    long long* out_i;
    uint32_t* flag_out;      // null, or the caller's word (RAG_SEARCH_DEFER_FALLBACK): set to 1 when a certificate failed
    const uint32_t* id_map;  // null, or [rows]: the id a row is ranked and reported by (the IVF list layout's stored ids)
#ifdef RAGK_STAMPS
    unsigned long long* stamps;  // experiment build: [nq][8] phase stamps
#endif
};
#ifdef RAGK_STAMPS
#define RAGK_RSTAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RAGK_RSTAMP(i) do { } while (0)
#endif

__host__ __device__ inline int resolve_rows_per_wave(int d8) { return d8 <= 1024 ? 8 : 4; }
__host__ __device__ inline size_t resolve_lds_bytes(int d8, int n_lists, int look, int kp) {
    const size_t stage = (size_t)4 * resolve_rows_per_wave(d8) * (d8 + 4) * sizeof(float);
    const size_t ahead = (size_t)n_lists * look * 8;
    return (size_t)kp * 16 + (size_t)d8 * sizeof(float) + (stage > ahead ? stage : ahead);
}

template <int OWN, int TW, int RPW>
__global__ __launch_bounds__(256) void screen_resolve_kernel(const ResolveParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* c_keys = reinterpret_cast<u64*>(smem);                   // [kp] exact ranking keys
    uint32_t* c_rows = reinterpret_cast<uint32_t*>(c_keys + p.kp);  // [kp] local row numbers
    float* c_approx = reinterpret_cast<float*>(c_rows + p.kp);      // [kp] approximate ranking scores
    float* qv = c_approx + p.kp;                                    // [d8] the query, zero padded (kp * 16 bytes in: aligned)
    u64* ahead = reinterpret_cast<u64*>(qv + p.d8);                 // [n_lists][look], phases A and B ...
    float* rows = reinterpret_cast<float*>(qv + p.d8);              // ... then [4 waves][RPW][d8 + 4], phase C
    __shared__ u64 wmax[2][4];
    __shared__ u64 s_kth;
    __shared__ uint32_t s_cnt, s_cut;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_lists = p.n_lists, look = p.look, kp = p.kp, kout = p.src.k;
    RAGK_RSTAMP(0);

    stage_heads<KeyListSrc, 256>(p.src, q, n_lists, look, ahead, tid);
    for (int c = tid; c < p.d8; c += 256) qv[c] = c < p.d ? p.Q[(size_t)q * p.d + c] : 0.f;
    if (tid == 0) {
        s_cnt = 0;
        s_cut = 0;
    }
    __syncthreads();

    RAGK_RSTAMP(1);
    // ---- (A) k-th best approximate key (TW == 1: wave 0 alone plays the tournament)
    if (TW == 4 || wave == 0) {
        const u64 last = tournament_rounds<KeyListSrc, OWN, TW, true>(p.src, q, n_lists, p.k, kout, look, ahead, wmax, tid,
                                                                       [](int, u64) {});
        if (tid == 0) s_kth = last;  // 0: fewer than k rows in all, everything listed is a candidate
    }
    __syncthreads();
    const u64 kth = s_kth;

    RAGK_RSTAMP(2);
    // ---- (B) the band
    const float margin = p.qs->margin[q];
    const float thr = kth != 0ull ? unord32((uint32_t)(kth >> 32)) - margin : -__builtin_inff();
    for (int l = tid; l < n_lists; l += 256) {
        int pp = 0;
        for (; pp < kout; ++pp) {
            const u64 key = pp < look ? ahead[(size_t)l * look + pp] : p.src.get(q, l, pp);
            if (key == 0ull) break;
            const float a = unord32((uint32_t)(key >> 32));
            if (a < thr) break;
            const uint32_t slot = atomicAdd(&s_cnt, 1u);
            if (slot < (uint32_t)kp) {
                c_approx[slot] = a;
                c_rows[slot] = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
            }
        }
        if (pp == kout) s_cut = 1;  // the whole list is in the band: rows below its cut may be too
        // a workgroup that gave up its band dropped nothing above its final k-th best (entry k-1): the
        // certificate survives unless that entry is itself inside the band
        else if (pp >= p.k && p.wg_lossy[(size_t)q * n_lists + l] == p.epoch) s_cut = 1;
    }
    __syncthreads();
    const int n = (int)min(s_cnt, (uint32_t)kp);                 // workgroup-uniform
    const bool overflow = s_cnt >= (uint32_t)kp || s_cut != 0;

    RAGK_RSTAMP(3);
    // ---- (C) canonical scores.  d8 / 4 <= 32 * 64 / RPW float4 per row: a lane holds NU of each of its wave's rows.
    constexpr int NU = 32 / RPW;
    const int d4 = p.d8 / 4;
    float* wrows = rows + (size_t)wave * RPW * (p.d8 + 4);
    for (int base = 0; base < n; base += 4 * RPW) {
        __syncthreads();  // the staging area is free: phase B (first pass) or the previous pass's chains are done
        const int g0 = base + wave * RPW;
        {
            f32x4 v[RPW][NU];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const bool have = g0 + j < n;  // wave-uniform
                const f32x4* src = reinterpret_cast<const f32x4*>(p.X + (long long)(have ? c_rows[g0 + j] : 0u) * p.row_stride);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    v[j][u] = (have && c < d4) ? src[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                f32x4* dst = reinterpret_cast<f32x4*>(wrows + (size_t)j * (p.d8 + 4));
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    if (c < d4) dst[c] = v[j][u];
                }
            }
        }
        __syncthreads();
        if (lane < RPW && g0 + lane < n) {
            const uint32_t my_row = c_rows[g0 + lane];
            const f32x4* x4 = reinterpret_cast<const f32x4*>(wrows + (size_t)lane * (p.d8 + 4));
            const f32x4* q4 = reinterpret_cast<const f32x4*>(qv);
            float acc = 0.f;
            f32x4 xa = x4[0], xb = x4[1], qa = q4[0], qb = q4[1];
            for (int s = 2; s <= d4; s += 2) {  // operands of the next group are read under this group's chain
                const int sn = s < d4 ? s : 0;
                const f32x4 xa_n = x4[sn], xb_n = x4[sn + 1], qa_n = q4[sn], qb_n = q4[sn + 1];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_fmaf(xa[t], qa[t], acc);
                    acc = __builtin_fmaf(xb[t], qb[t], acc);
                }
                xa = xa_n;
                xb = xb_n;
                qa = qa_n;
                qb = qb_n;
            }
            float score = acc + 0.0f;
            if (p.l2) score = __builtin_fmaf(2.0f, acc, -p.xnorm[my_row]) + 0.0f;
            c_keys[g0 + lane] = make_key(score, p.id_map ? p.id_map[my_row] : my_row);
        }
    }
    __syncthreads();
    RAGK_RSTAMP(4);
    if (wave != 0) return;

    // ---- (D) certificate, ranking, results: one wave, kp <= 256 keys (the usual few dozen take the
    // one-key-per-lane sort)
    constexpr int E = 4;
    u64 kk[1][E];
    float worst = 0.f;
    const float eps = margin * (p.l2 ? 0.25f : 0.5f);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = e * 64 + lane;
        u64 key = 0ull;
        if (idx < n) {
            key = c_keys[idx];
            const float ex = unord32((uint32_t)(key >> 32));
            const float err = fabsf(c_approx[idx] - ex) * (p.l2 ? 0.5f : 1.f);
            worst = fmaxf(worst, err);
        }
        kk[0][e] = key;
    }
    worst = wave_max_f32(worst);
    // the scan kept the whole band (not lossy), this kernel held it (no overflow), and — belt and braces —
    // no verified candidate shows an error beyond the bound the band was built from
    bool ok = p.qs->lossy[q] != p.epoch && !overflow;
    if (ok && !(worst <= eps)) ok = false;  // also catches eps == 0 with any error, and NaN
    if (n <= 64) {
        u64 k1[1][1] = {{kk[0][0]}};
        wave_sort_desc<1, 1>(k1, lane);
        kk[0][0] = k1[0][0];
    } else {
        wave_sort_desc<E, 1>(kk, lane);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = e * 64 + lane;
        if (idx < p.k) {
            const u64 key = kk[0][e];
            float sc;
            long long id;
            if (key == 0ull) {
                sc = p.l2 ? 3.402823466e+38f : -3.402823466e+38f;
                id = -1;
            } else {
                const float rs = unord32((uint32_t)(key >> 32));
                if (p.l2) {
                    const float dist = p.qnorm[q] - rs;
                    sc = dist < 0.f ? 0.f : dist;
                } else {
                    sc = rs;
                }
                id = (long long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) + p.id_offset;
            }
            p.out_s[(size_t)q * p.k + idx] = sc;
            p.out_i[(size_t)q * p.k + idx] = id;
        }
    }
    if (lane == 0) {
        atomicAdd(&p.ctr->queries, 1ull);
        if (!ok) {
            p.qs->fallback[q] = p.epoch;
            p.qs->any_fallback = p.epoch;
            if (p.flag_out) *p.flag_out = 1u;
            atomicAdd(&p.ctr->fallbacks, 1ull);
        } else if (eps > 0.f) {
            atomicMax(&p.ctr->max_err_ratio_bits, __float_as_uint(worst / eps));
        }
    }
    RAGK_RSTAMP(5);
}

// ---- helpers ----------------------------------------------------------------------------------

// Canonical fp32 dot (oracle/flat_oracle.c:rago_dot): zero-padded to a multiple of 8, each group
// of 8 visited 0,4,1,5,2,6,3,7, one fmaf per term.
__device__ __forceinline__ float canonical_sqnorm(const float* x, int d) {
    float acc = 0.f;
    const int d8 = (d + 7) & ~7;
    for (int s = 0; s < d8; s += 8) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = s + t, k1 = s + 4 + t;
            const float a = k0 < d ? x[k0] : 0.f;
            const float b = k1 < d ? x[k1] : 0.f;
            acc = __builtin_fmaf(a, a, acc);
            acc = __builtin_fmaf(b, b, acc);
        }
    }
    return acc;
}

__global__ void row_sqnorm_kernel(const float* X, long long row_stride, int d, long long row0, long long n,
                                  float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[row0 + i] = canonical_sqnorm(X + (row0 + i) * row_stride, d);
}

// The same per QUERY (a few rows, on the critical path of every L2 search): one wave per row stages it in LDS with
// coalesced loads, then one lane runs the canonical chain from there (a thread per row waits for a dependent global
// load every few terms: 17 us for d = 768; this is 4).  Same bits as row_sqnorm_kernel.
__global__ __launch_bounds__(64) void query_sqnorm_kernel(const float* Q, int d, float* out) {
    extern __shared__ __attribute__((aligned(16))) float qrow[];
    const int q = blockIdx.x, lane = threadIdx.x;
    const int d8 = (d + 7) & ~7;
    for (int c = lane; c < d8; c += 64) qrow[c] = c < d ? Q[(size_t)q * d + c] : 0.f;
    __syncthreads();
    if (lane == 0) {
        float acc = 0.f;
        for (int s = 0; s < d8; s += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + s), b = *reinterpret_cast<const f32x4*>(qrow + s + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = __builtin_fmaf(a[t], a[t], acc);
                acc = __builtin_fmaf(b[t], b[t], acc);
            }
        }
        out[q] = acc;
    }
}

__global__ void fill_neutral_kernel(float* scores, long long* ids, int total, int metric) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    scores[i] = metric ? 3.402823466e+38f : -3.402823466e+38f;
    ids[i] = -1;
}

// Synthetic corpus, bit-exact with oracle/flat_oracle.c:rago_synth_rows.
__device__ __forceinline__ u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float synth_elem(u64 seed, u64 row, uint32_t col, uint32_t d) {
    const u64 rr = mix64(seed ^ mix64(row * (u64)d + col));
    const int s = (int)(rr & 0xFFFF) + (int)((rr >> 16) & 0xFFFF) + (int)((rr >> 32) & 0xFFFF) +
                  (int)((rr >> 48) & 0xFFFF);
    return (float)(s - 131070) * (1.0f / 65536.0f);
}
// 1/sqrt from multiplies and fmas only, bit-exact with oracle/flat_oracle.c:rago_rsqrt
__device__ __forceinline__ float synth_rsqrt(float v) {
    if (!(v > 0.f)) return 0.f;
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(v) >> 1));
    const float hh = __fmul_rn(0.5f, v);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float t = __fmul_rn(y, y);
        const float w = __builtin_fmaf(-hh, t, 1.5f);
        y = __fmul_rn(y, w);
    }
    return y;
}
// pass 1: inv[i] = rsqrt(sum_j elem^2), the sum a sequential fmaf chain in index order
__global__ void synth_inv_kernel(u64 seed, long long row_number0, long long n, int d, float* inv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ss = 0.f;
    for (int j = 0; j < d; ++j) {
        const float v = synth_elem(seed, (u64)(row_number0 + i), (uint32_t)j, (uint32_t)d);
        ss = __builtin_fmaf(v, v, ss);
    }
    inv[i] = synth_rsqrt(ss);
}
// pass 2: coalesced element fill, X[(dst_row0+i)*stride + j] = elem * inv[i] (pad columns = 0)
__global__ void synth_fill_kernel(u64 seed, long long row_number0, long long n, int d, int d8,
                                  long long row_stride, const float* inv, float* X, long long dst_row0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = n * (long long)d8;
    if (idx >= total) return;
    const long long i = idx / d8;
    const int j = (int)(idx - i * d8);
    float v = 0.f;
    if (j < d) v = __fmul_rn(synth_elem(seed, (u64)(row_number0 + i), (uint32_t)j, (uint32_t)d), inv[i]);
    X[(dst_row0 + i) * row_stride + j] = v;
}

}  // namespace ragk
