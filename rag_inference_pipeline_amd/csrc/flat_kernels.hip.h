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
// plus small helpers (row norms, query norms, synthetic corpus, neutral fill).
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
typedef unsigned long long u64;

constexpr int kQT = 32;        // queries per scan pass (MFMA N)
constexpr int kTileRows = 32;  // corpus rows per wave tile (MFMA M)
constexpr int kMaxScanWaves = 16;

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

struct ScanParams {
    const float* X;        // corpus, row-major, row_stride floats per row (zero padded to d8)
    const float* xnorm;    // canonical squared norms (L2 metric only)
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
    int n_iters;           // tiles per wave
};

// LDS layout: [Q fragments d8*128 B][keys 32*C*8 B][cnt 32 u32][thr 32 f32][flag 4 u32]
__host__ __device__ inline size_t scan_lds_bytes(int d8, int C) {  // d8 = columns held in LDS (one chunk)
    return (size_t)d8 * 128 + (size_t)kQT * C * 8 + kQT * 4 + kQT * 4 + 16;
}

// NW = waves per workgroup (NW/4 per SIMD), E = buffer capacity / 64, D = register ring depth
// (steps of 8 columns in flight per wave), L2 = metric.
template <int NW, int E, int D, bool L2>
__global__ __launch_bounds__(NW * 64) void scan_topk_kernel(const ScanParams p) {
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

    // ---- prologue: queries -> MFMA B fragments.  Fragment (s, l) = Q[l&31][8s + 4(l>>5) .. +3].
    const bool q_vec = (p.d & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Q) & 15) == 0;
    for (int idx = tid; idx < S * 64; idx += kScanWaves * 64) {
        const int s = idx >> 6, l = idx & 63;
        const int qrow = l & 31, col = p.col0 + 8 * s + 4 * (l >> 5);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qrow < p.nq) {
            const float* src = p.Q + (size_t)qrow * p.d + col;
            if (q_vec && col + 3 < p.d) {  // rows 16-byte aligned: one load per fragment
                v = *reinterpret_cast<const f32x4*>(src);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col + j < p.d) v[j] = src[j];
            }
        }
        qf[idx] = v;
    }
    if (tid < kQT) {
        cnt[tid] = 0;
        // unused query columns (nq < 32) score 0 against every row: park their threshold at +inf so
        // they never enter the slow path (left at -inf they tie forever and double the scan time)
        thr[tid] = tid < p.nq ? -__builtin_inff() : __builtin_inff();
    }
    if (tid < 4) flag[tid] = 0;
    __syncthreads();

    const long long last_row = p.n_rows - 1;
    const int tiles_per_iter = gridDim.x * kScanWaves;
    int tile = blockIdx.x * kScanWaves + wave;
    auto row_ptr = [&](int t) -> const float* {
#if defined(RAGK_ABLATE_L2_WINDOW)
        t &= 31;
#endif
        long long row = (long long)t * kTileRows + r;
        row = row < last_row ? row : last_row;
        return p.X + row * p.row_stride + p.col0 + 4 * h;
    };

    const u64 key_ceil = p.ceil ? p.ceil[r] : ~0ull;
    f32x4 xb[D];
    const float* pc = row_ptr(tile < p.n_tiles ? tile : p.n_tiles - 1);
#pragma unroll
    for (int i = 0; i < D; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pc + 8 * i);

    // Overflow flags: pass number seq (one pass = one trip through the barrier loop below) owns
    // flag[seq & 3]; appends raise the flag of the pass that will check them, and pass seq clears
    // the slot of pass seq + 2, so a wave that has already left a pass never races a wave that is
    // still reading that pass's flag.
    uint32_t seq = 0;
    for (int it = 0; it < p.n_iters; ++it, tile += tiles_per_iter) {
        const bool active = tile < p.n_tiles;  // wave-uniform
        const int tnext = tile + tiles_per_iter;
        const float* pn = row_ptr(tnext < p.n_tiles ? tnext : p.n_tiles - 1);

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
                    acc = RAGK_MFMA(xb[i][0], qcur[0], acc);
                    acc = RAGK_MFMA(xb[i][1], qcur[1], acc);
                    acc = RAGK_MFMA(xb[i][2], qcur[2], acc);
                    acc = RAGK_MFMA(xb[i][3], qcur[3], acc);
                    qcur = qn;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // next query fragment (DS read)
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
                }
#pragma unroll
                for (int j = 0; j < G; ++j) xb[g + j] = *reinterpret_cast<const f32x4*>(src + 8 * (g + j));
                __builtin_amdgcn_sched_group_barrier(0x020, G, 0);  // ring refill (VMEM reads)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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
                    acc = RAGK_MFMA(xb[i][0], qcur[0], acc);
                    acc = RAGK_MFMA(xb[i][1], qcur[1], acc);
                    acc = RAGK_MFMA(xb[i][2], qcur[2], acc);
                    acc = RAGK_MFMA(xb[i][3], qcur[3], acc);
                    qcur = qn;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
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
            continue;
        }

        // ---- ranking scores.  Lane (r, h): query r, tile rows (i&3) + 8(i>>2) + 4h.
        const long long row0 = (long long)tile * kTileRows;
        float sc[16];
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

        // ---- filter + append
        uint32_t pending = 0;
        if (active) {
            float t = thr[r];
            float m = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, sc[i]);
            if (m >= t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (sc[i] >= t && row <= last_row && make_key(sc[i], (uint32_t)row) < key_ceil) {
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
                    if (q < kQT && nn[j] > 0) {  // sorting a short buffer is harmless: cnt and thr stay consistent
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int idx = e * 64 + lane;
                            if (idx < p.k) keys[(size_t)q * C + idx] = kk[j][e];
                            if (idx == p.k - 1)
                                thr[q] = nn[j] >= (uint32_t)p.k ? unord32((uint32_t)(kk[j][e] >> 32)) : -__builtin_inff();
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
    }

    if (p.acc_out) return;
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
                u64* out = p.partial + ((size_t)q * gridDim.x + blockIdx.x) * p.k;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = e * 64 + lane;
                    if (idx < p.k) out[idx] = kk[j][e];
                }
            }
        }
    }
}

// ---- K2: merge ------------------------------------------------------------------------------
// Top-k of n_lists lists that are each sorted best-first (the per-workgroup lists K1 emits, or the
// per-shard lists after the all-gather).  One 256-thread workgroup per query plays a k-round
// tournament: every thread holds the head of the list(s) it owns, a round is one workgroup-wide
// 64-bit max (shuffles + 4 LDS words), the winner advances its list.  ~150 cycles per round, so
// k = 10 over 256 lists costs about a microsecond of device time — the bitonic merge this replaces
// sorted 4096 keys to keep 10 and took 90.
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
        if (id < 0) return 0ull;
        const float s = scores[(size_t)l * score_stride + e];
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
};

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = umax64(v, shfl_xor_u64(v, m));
    return v;
}

template <class Src>
__global__ __launch_bounds__(256) void tournament_merge_kernel(const Src src, const int n_lists, const int k,
                                                               const MergeOut out) {
    __shared__ u64 wmax[2][4];
    const int q = blockIdx.x, tid = threadIdx.x;
    u64 head[kMergeMaxOwned], next[kMergeMaxOwned];
    int pos[kMergeMaxOwned];
#pragma unroll
    for (int j = 0; j < kMergeMaxOwned; ++j) {
        const int l = tid + 256 * j;
        pos[j] = 0;
        head[j] = (l < n_lists && k > 0) ? src.get(q, l, 0) : 0ull;
        next[j] = (l < n_lists && k > 1) ? src.get(q, l, 1) : 0ull;  // one key of look-ahead per list
    }
    for (int round = 0; round < k; ++round) {
        u64 best = head[0];
#pragma unroll
        for (int j = 1; j < kMergeMaxOwned; ++j) best = umax64(best, head[j]);
        const u64 wm = wave_max_u64(best);
        if ((tid & 63) == 0) wmax[round & 1][tid >> 6] = wm;
        __syncthreads();  // the other parity is free again: its readers passed this barrier's predecessor
        const u64 bm = umax64(umax64(wmax[round & 1][0], wmax[round & 1][1]),
                              umax64(wmax[round & 1][2], wmax[round & 1][3]));
        if (tid == 0) {
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
            if (out.last_key && round == k - 1) out.last_key[q] = bm;
        }
        if (bm != 0ull && best == bm) {  // keys are unique, so exactly one thread advances one list
#pragma unroll
            for (int j = 0; j < kMergeMaxOwned; ++j) {
                if (head[j] == bm) {
                    const int l = tid + 256 * j;
                    pos[j] += 1;
                    head[j] = next[j];
                    next[j] = pos[j] + 1 < k ? src.get(q, l, pos[j] + 1) : 0ull;
                }
            }
        }
    }
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
