// ivf_kernels.hip.h — the IVFFlat `nprobe` mode (rag_ivf_* in include/rag_amd.h): plan + batch list scan.
//
// The reference's own data generator writes a faiss IndexIVFFlat (scripts/create_test_docs.py:83-104: 4.5M x 768, L2,
// nlist 4096, nprobe 64) and FAISSStore.load sets `index.nprobe` (faiss_store.py:84-92): on such a file the reference
// does NOT search exhaustively — a query only sees the rows of the `nprobe` inverted lists whose centroids are nearest
// to it.  Step 1, the coarse quantizer, is the flat scan over the nlist centroids.  Step 2 is here and is built the way
// the flat scan is, because it is the same HBM-bound job on a subset of the rows:
//
//   * rows live in LIST order, every list padded to whole 32-row tiles (padding rows are zeros with id 0xFFFFFFFF), so a
//     wave's tile never straddles two lists;
//   * `ivf_plan_kernel` (one workgroup) turns a pass's probe table [nq <= 32][nprobe] into per-list query MASKS and the
//     packed sequence of (tile, mask) entries of every list that at least one query probes; a work ITEM is 8 consecutive
//     entries — whole items whatever the lists' lengths (lists of 9 tiles would otherwise run as 8 + 1).  A list probed
//     by several queries of the batch is read ONCE;
//   * `ivf_batch_scan_kernel` (one 8-wave workgroup per CU, items dealt by ticket): a wave multiplies its 32-row tile
//     with ALL 32 queries of the pass on v_mfma_f32_32x32x2_f32 — the flat scan's inner loop, hence the flat search's
//     canonical summation order and score bits — and lanes of queries that do not probe the item's list drop their
//     scores.  The matrix pipe does up to 32x the needed products and does not care: the kernel moves 3 KB per row at
//     HBM rate and the pipe runs at about half of its peak beside it, as in the flat scan.  Selection is the flat scan's
//     (threshold filter -> LDS candidate buffers -> wave bitonic sort on overflow); a workgroup emits its best k keys per
//     query and the flat search's tournament merge ranks the workgroups' lists.
//
// The default search (k <= 100) is the TWO-STAGE form of this: the screening pass is the flat scan kernel itself, in its MAP
// mode, over a scaled fp16 copy of the rows (flat_kernels.hip.h scan_topk_kernel<.., P = 1, MAP = true>: the plan's packed
// entries as the tile walk, the masks, padding rows excluded), then screen_resolve_kernel (canonical fp32 scores of the
// band, certificate, ranking by stored id); `ivf_batch_scan_kernel` below is then the self-disabling fallback — and the
// whole search when k > 100 or RAG_AMD_IVF_TWO_STAGE=0 (rag_ivf_host.hip.h).
//
// Ranking is (score, ascending stored id) — faiss leaves ties to its heap's visiting order; the candidate SET (the union
// of the probed lists) is what the mode is about.  With nprobe >= nlist the result equals rag_index_search's bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragk {

constexpr uint32_t kIvfPadId = 0xFFFFFFFFu;   // id of a padding row
constexpr int kIvfItemTiles = 8;                // tiles per work item = waves per workgroup
constexpr int kIvfMaxLists = 32768;             // the plan kernel keeps one mask word per list in LDS
constexpr int kIvfThrStride = 32;               // words between two queries' shared thresholds: a 128-byte line each

using IvfTile = TileMapEntry;   // 8 bytes, one wave's work in an item: {tile number, bit q = query q of the pass probes its list}
constexpr uint32_t kIvfNoTile = 0xFFFFFFFFu;

struct IvfPlanParams {
    const long long* probe;     // [nq][nprobe] list numbers from the coarse search, -1 = none
    const uint32_t* tile_off;   // [nlist + 1] first tile of each list (padded layout)
    IvfTile* tiles;             // out: the probed lists' tiles, packed ([<= all tiles])
    uint32_t* n_tiles;          // out
    uint32_t* gthr;             // [2 * kQT * kIvfThrStride] the scans' shared thresholds / band edges: zeroed here
    int nq, nprobe, nlist;
};

// One workgroup of 1024 threads: masks in LDS, then lists 1024 at a time: tiles per probed list -> running prefix -> entries.
__global__ __launch_bounds__(1024) void ivf_plan_kernel(const IvfPlanParams p) {
    extern __shared__ __attribute__((aligned(16))) char ivf_smem[];
    uint32_t* mask = reinterpret_cast<uint32_t*>(ivf_smem);   // [nlist]
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t base_s, n_big;
    __shared__ uint32_t big[256][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int l = t; l < p.nlist; l += 1024) mask[l] = 0u;
    if (t == 0) {
        base_s = 0u;
        n_big = 0u;
    }
    if (t < 2 * kQT) p.gthr[t * kIvfThrStride] = 0u;   // the exact scan's thresholds and the screening pass's band edges
    __syncthreads();
    const int np = p.nq * p.nprobe;
    for (int i = t; i < np; i += 1024) {
        const long long l = p.probe[i];
        if (l >= 0 && l < p.nlist) atomicOr(&mask[l], 1u << (i / p.nprobe));
    }
    __syncthreads();
    for (int l0 = 0; l0 < p.nlist; l0 += 1024) {
        const int l = l0 + t;
        uint32_t m = 0u, ft = 0u, nt = 0u;
        if (l < p.nlist) {
            m = mask[l];
            ft = p.tile_off[l];
            nt = m ? p.tile_off[l + 1] - ft : 0u;
        }
        uint32_t incl = nt;   // inclusive prefix within the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const uint32_t at = before + incl - nt;
        // the entries of a list are written by the whole wave, those of a long list (the generator's Gaussian rows put
        // 800 000 of 4.5M rows — 25 000 tiles — into one) by the whole workgroup
        const bool is_big = nt > 512u;
        if (is_big) {
            const uint32_t slot = atomicAdd(&n_big, 1u);
            if (slot < 256u) {
                big[slot][0] = ft;
                big[slot][1] = nt;
                big[slot][2] = m;
                big[slot][3] = at;
            }
        }
        __syncthreads();
        const bool big_fits = n_big <= 256u;   // (else every list goes the wave's way)
        unsigned long long todo = __ballot(nt != 0u && !(is_big && big_fits));
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t l_ft = __shfl(ft, src, 64), l_nt = __shfl(nt, src, 64), l_m = __shfl(m, src, 64);
            const uint32_t l_at = __shfl(at, src, 64);
            for (uint32_t j = lane; j < l_nt; j += 64) p.tiles[l_at + j] = IvfTile{l_ft + j, l_m};
        }
        if (big_fits) {
            for (uint32_t b = 0; b < n_big; ++b) {
                const uint32_t l_ft = big[b][0], l_nt = big[b][1], l_m = big[b][2], l_at = big[b][3];
                for (uint32_t j = t; j < l_nt; j += 1024) p.tiles[l_at + j] = IvfTile{l_ft + j, l_m};
            }
        }
        __syncthreads();
        if (t == 0) n_big = 0u;
        __syncthreads();
        if (t == 1023) base_s = before + incl;
        __syncthreads();
    }
    if (t == 0) *p.n_tiles = base_s;
}

// ---- coarse quantizer: selection ------------------------------------------------------------------------------
// The flat search over a few thousand centroids with k = nprobe = 64 is all fixed cost: 128 workgroups each emit a
// 64-key list of their one tile and the tournament merge takes 64 rounds over 128 lists (45 + 35 us — more than a
// batch-1 list scan).  Instead the flat scan kernel runs in its "park the inner products" mode (ScanParams.acc_out: the
// canonical sums of every (centroid, query) pair go to acc[row][32], no selection) and this kernel picks the nprobe best
// per query.  One workgroup per query; a wave sorts 8 chunks of 64 keys at once (the network's shuffle latencies overlap
// across the sets), folds them pairwise — elementwise max against the partner list REVERSED leaves the 64 largest of the
// union as a bitonic sequence, which takes the 6-stage merge network, not a 21-stage sort — and folds the survivor into
// its running best; wave 0 folds the four waves' lists.  Same keys as the flat search: (score, ascending list number).

// the last pass of the bitonic network: every one of the first `nsets` sets holds a bitonic sequence -> descending
template <int E, int NQ>
__device__ __forceinline__ void wave_merge_desc(u64 (&key)[NQ][E], int lane, int nsets) {
#pragma unroll
    for (int stride = 32 * E; stride > 0; stride >>= 1) {
        if (stride >= 64) {
            const int se = stride >> 6;
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                if (s < nsets) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & se) == 0) {
                            const u64 a = key[s][e], b = key[s][e | se];
                            key[s][e] = umax64(a, b);
                            key[s][e | se] = umin64(a, b);
                        }
                    }
                }
            }
        } else {
            u64 other[NQ][E];
#pragma unroll
            for (int s = 0; s < NQ; ++s)
                if (s < nsets)
#pragma unroll
                    for (int e = 0; e < E; ++e) other[s][e] = shfl_xor_u64(key[s][e], stride);
            const bool lower = (lane & stride) == 0;
#pragma unroll
            for (int s = 0; s < NQ; ++s)
                if (s < nsets)
#pragma unroll
                    for (int e = 0; e < E; ++e) key[s][e] = lower ? umax64(key[s][e], other[s][e]) : umin64(key[s][e], other[s][e]);
        }
    }
}

// a <- the 64 E largest of (a U b), both sorted descending, as a bitonic sequence (index i of a meets index 64 E - 1 - i of b)
template <int E>
__device__ __forceinline__ void fold_reversed(u64 (&a)[E], const u64 (&b)[E], int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u64 v = b[E - 1 - e];
        const uint32_t lo = __shfl((uint32_t)v, 63 - lane, 64), hi = __shfl((uint32_t)(v >> 32), 63 - lane, 64);
        a[e] = umax64(a[e], ((u64)hi << 32) | lo);
    }
}

// NWV waves per workgroup (8 from 2048 keys up).  The kernel also computes the query's canonical ||q||^2 when asked to
// (Q, qnorm_out != null: one lane of wave 1, while wave 0 folds the waves' lists — a separate 9 us launch otherwise) and
// applies the dead-query rule with it at the end (every distance inf or NaN: no list is "nearest").
template <int E, int NWV>
__global__ __launch_bounds__(64 * NWV) void ivf_coarse_select_kernel(const float* acc, const float* cnorm, const float* qnorm_in,
                                                                    const float* Q, int d, float* qnorm_out, int nlist, int np,
                                                                    int l2, long long* probe) {
    constexpr int NB = 8 / E;      // chunks in flight per wave
    constexpr int CH = 64 * E;     // keys per chunk
    __shared__ u64 tops[NWV][CH];
    __shared__ float s_qn;
    extern __shared__ __attribute__((aligned(16))) char ivf_smem[];
    float* qrow = reinterpret_cast<float*>(ivf_smem);   // [d8] (only when the norm is computed here)
    const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d8 = (d + 7) & ~7;
    if (qnorm_out)
        for (int c = threadIdx.x; c < d8; c += 64 * NWV) qrow[c] = c < d ? Q[(size_t)q * d + c] : 0.f;
    u64 top[1][E];
#pragma unroll
    for (int e = 0; e < E; ++e) top[0][e] = 0ull;
    for (int base = wave * NB * CH; base < nlist; base += NWV * NB * CH) {
        u64 c[NB][E];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int row = base + s * CH + e * 64 + lane;
                u64 key = 0ull;
                if (row < nlist) {
                    const float a = acc[(size_t)row * kQT + q];
                    const float sc = l2 ? __builtin_fmaf(2.0f, a, -cnorm[row]) + 0.0f : a + 0.0f;
                    if (sc >= RAGK_SCORE_FLOOR) key = make_key(sc, (uint32_t)row);
                }
                c[s][e] = key;
            }
        }
        wave_sort_desc<E, NB>(c, lane);
#pragma unroll
        for (int w = NB / 2; w >= 1; w >>= 1) {
#pragma unroll
            for (int s = 0; s < NB / 2; ++s)
                if (s < w) fold_reversed<E>(c[s], c[s + w], lane);
            wave_merge_desc<E, NB>(c, lane, w);
        }
        fold_reversed<E>(top[0], c[0], lane);
        wave_merge_desc<E, 1>(top, lane, 1);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) tops[wave][e * 64 + lane] = top[0][e];
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < NWV; ++w) {
#pragma unroll
            for (int e = 0; e < E; ++e) top[0][e] = umax64(top[0][e], tops[w][(E - 1 - e) * 64 + 63 - lane]);
            wave_merge_desc<E, 1>(top, lane, 1);
        }
    } else if (wave == 1 && lane == 0) {
        float qn = qnorm_in ? qnorm_in[q] : 0.f;
        if (qnorm_out) {   // the canonical chain (flat_kernels.hip.h query_sqnorm_kernel: the same bits)
            float a = 0.f;
            for (int s = 0; s < d8; s += 8) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(qrow + s), x1 = *reinterpret_cast<const f32x4*>(qrow + s + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a = __builtin_fmaf(x0[t], x0[t], a);
                    a = __builtin_fmaf(x1[t], x1[t], a);
                }
            }
            qn = a;
            qnorm_out[q] = a;
        }
        s_qn = qn;
    }
    __syncthreads();
    if (wave != 0) return;
    const bool dead = l2 && !(s_qn <= 3.402823466e+38f);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = e * 64 + lane;
        if (j < np) probe[(size_t)q * np + j] = (top[0][e] && !dead) ? (long long)(0xFFFFFFFFu - (uint32_t)top[0][e]) : -1ll;
    }
}

struct IvfBatchParams {
    const float* X;            // rows in padded list order, row_stride floats per row (columns zero-padded to d8)
    long long row_stride;
    const float* xnorm;        // [rows] canonical squared norms (L2)
    const uint32_t* ids;       // [rows] stored id, kIvfPadId for padding rows
    const float* Q;            // [nq][d]
    const float* qnorm;        // [nq] (L2): a non-finite one means no results
    const IvfTile* tiles;      // the pass's packed (tile, mask) entries: item g = entries [8 g, 8 g + 8)
    const uint32_t* n_tiles;
    uint32_t* ticket;          // 0 at launch; the last workgroup to leave zeroes it again
    uint32_t* done;            // workgroups that have left (same)
    u64* partial;              // out: [kQT][grid][k] keys, sorted descending per (query, workgroup)
    const u64* ceil;           // [kQT] or null: only keys strictly below ceil[q] are candidates (k > max_k rounds)
    const uint32_t* enable;    // null, or a device word: unless it equals `epoch` the launch is a no-op (the two-stage
    uint32_t epoch;            // search's fallback: flat_kernels.hip.h ScanParams.enable)
    uint32_t* gthr;            // [kQT * kIvfThrStride] thresholds shared by the workgroups (ord32 of a score, 0 = none
                               // yet; one 128-byte line per query): see the filter
    int d, d8, nq, k, l2;
};

// LDS: [Q fragments d8 * 128 B][keys 32 * C * 8 B][cnt 32][thr 32][flag 4][item records 2 x 8 entries][sequence word + pad]
__host__ __device__ inline size_t ivf_scan_lds_bytes(int d8, int C) {
    return (size_t)d8 * 128 + (size_t)kQT * C * 8 + kQT * 4 + kQT * 4 + 16 + 128 + 16;
}

// E = candidate buffer capacity / 64, D = register ring depth (steps of 32 bytes per row in flight per wave).
template <int E, int D>
__global__ __launch_bounds__(512) void ivf_batch_scan_kernel(const IvfBatchParams p) {
    constexpr int NW = kIvfItemTiles;
    constexpr int C = 64 * E;
    if (p.enable && *p.enable != p.epoch) return;   // launch-uniform, before any ticket is drawn
    extern __shared__ __attribute__((aligned(16))) char ivf_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int S = p.d8 >> 3;

    f32x4* qf = reinterpret_cast<f32x4*>(ivf_smem);
    u64* keys = reinterpret_cast<u64*>(ivf_smem + (size_t)S * 1024);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(keys + (size_t)kQT * C);
    float* thr = reinterpret_cast<float*>(cnt + kQT);
    uint32_t* flag = reinterpret_cast<uint32_t*>(thr + kQT);
    // the item of iteration it sits in itemw[16 (it & 1) ..] (8 entries of two words); itemw[32] = the newest iteration
    // whose item is posted.  Lanes 0-7 of wave 0 write entries then number, readers read number then entries (LDS
    // operations of a wave execute in order)
    volatile uint32_t* itemw = reinterpret_cast<volatile uint32_t*>(flag + 4);

    // ---- the deal.  Item g = entries [8 g, 8 g + 8) of the packed tile sequence, entry w for wave w.  Iteration 0 of
    // workgroup b takes item b, iteration 1 item grid + b, every later one the item of a TICKET (item = ticket + 2 grid)
    // drawn from one global counter.  Lanes 0-7 of wave 0 run a three-deep pipeline at the TOP of each iteration — post
    // the entries fetched an iteration ago (for iteration it + 1), fetch the entries of the ticket drawn an iteration ago
    // (it + 2), draw the next ticket (it + 3; lane 0) — so every returning memory operation has a whole tile time before
    // its result is used, and the use sits right behind the iteration's barrier, where the wave's load queue is drained
    // anyway (vmcnt counts in order: a wait for the ticket in the middle of the K loop would be a wait for the whole
    // register ring).
    const bool dealer = wave == 0 && lane < NW;
    uint32_t n_tiles = 0u, g_pend = 0u;
    IvfTile rec_pend{kIvfNoTile, 0u};
    auto fetch = [&](uint32_t item) -> IvfTile {   // dealer lanes: this lane's entry of `item`
        const uint32_t v = item * NW + lane;
        IvfTile e{kIvfNoTile, 0u};
        if (item < 0x10000000u && v < n_tiles) e = p.tiles[v];
        return e;
    };
    if (dealer) {
        n_tiles = *p.n_tiles;
        const IvfTile rec0 = fetch(blockIdx.x);
        rec_pend = fetch(gridDim.x + blockIdx.x);
        if (lane == 0) g_pend = atomicAdd(p.ticket, 1u);
        itemw[2 * lane] = rec0.tile;
        itemw[2 * lane + 1] = rec0.mask;
        if (lane == 0) itemw[32] = 0u;
    }

    // ---- prologue: queries -> MFMA B fragments in LDS: fragment (s, l) = Q[l & 31][8 s + 4 (l >> 5) .. + 3]
    {
        constexpr int PU = 12;
        const bool have_q = r < p.nq;
        const float* qbase = p.Q + (size_t)(have_q ? r : 0) * p.d;
        const bool q_vec = (p.d & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Q) & 15) == 0;
        for (int s0 = wave; s0 < S; s0 += NW * PU) {
            f32x4 v[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = s0 + NW * u;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < S && have_q) {
                    const int col = 8 * s + 4 * h;
                    if (q_vec && col + 3 < p.d) {
                        v[u] = *reinterpret_cast<const f32x4*>(qbase + col);
                    } else {
                        f32x4 t = {0.f, 0.f, 0.f, 0.f};
                        if (col + 0 < p.d) t[0] = qbase[col + 0];
                        if (col + 1 < p.d) t[1] = qbase[col + 1];
                        if (col + 2 < p.d) t[2] = qbase[col + 2];
                        if (col + 3 < p.d) t[3] = qbase[col + 3];
                        v[u] = t;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = s0 + NW * u;
                if (s < S) qf[s * 64 + lane] = v[u];
            }
        }
        if (tid < kQT) {
            float t0 = RAGK_SCORE_FLOOR;
            // L2 with ||q||^2 = inf / NaN: every distance is inf or NaN, which faiss's heap never admits
            if (tid < p.nq && p.l2 && !(p.qnorm[tid] <= 3.402823466e+38f)) t0 = __builtin_nanf("");
            cnt[tid] = 0;
            thr[tid] = tid < p.nq ? t0 : __builtin_inff();   // unused query columns parked
        }
        if (tid < 4) flag[tid] = 0;
    }
    __syncthreads();

    const u64 key_ceil = p.ceil ? p.ceil[r] : ~0ull;
    auto row_ptr = [&](uint32_t tile) -> const float* { return p.X + ((long long)tile * kTileRows + r) * p.row_stride + 4 * h; };

    // iteration 0's item (posted before the barrier): this wave's entry, and entry 0 (none = no more items)
    uint32_t tile = __builtin_amdgcn_readfirstlane(itemw[2 * wave]), imask = __builtin_amdgcn_readfirstlane(itemw[2 * wave + 1]);
    bool more = __builtin_amdgcn_readfirstlane(itemw[0]) != kIvfNoTile;
    f32x4 xb[D];
    const float* pc = p.X;
    bool active = tile != kIvfNoTile;
    if (active) {
        pc = row_ptr(tile);
#pragma unroll
        for (int i = 0; i < D; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pc + 8 * i);
    }

    uint32_t seq = 0, g_seen = 0u;
    for (int it = 0; more; ++it) {   // workgroup-uniform
        if (dealer) {
            volatile uint32_t* w = itemw + 16 * ((it + 1) & 1);
            w[2 * lane] = rec_pend.tile;
            w[2 * lane + 1] = rec_pend.mask;
            if (lane == 0) itemw[32] = (uint32_t)(it + 1);   // (a wave's LDS writes complete in order)
            const uint32_t g = __builtin_amdgcn_readfirstlane(g_pend);
            rec_pend = fetch(g + 2u * gridDim.x);
            if (lane == 0) g_pend = atomicAdd(p.ticket, 1u);
        }
        // Shared thresholds.  A workgroup meets a given query in one or two of its ~30 items (a query probes 64 of the
        // batch's ~1600 lists), so a filter that learns only from the workgroup's own rows stays cold: every first meeting
        // puts 256 rows into a 64-slot buffer — overflow, workgroup-wide sort, retry.  So a k-th best established in a sort
        // is published (atomic max on the ordered score, one line per query) and wave 0 refreshes the local filters from the
        // published values once per iteration — the load of one iteration is consumed at the top of the next, like the
        // ticket.  k rows with at least the published score exist somewhere, hence a row below it is not in the query's
        // top k; a stale value is only a weaker bound.
        if (p.gthr && wave == 0 && lane < kQT) {
            if (g_seen) {
                const float tg = unord32(g_seen);
                if (tg > thr[lane]) thr[lane] = tg;   // NaN (dead query) and +inf (unused column) filters stay
            }
            g_seen = __hip_atomic_load(p.gthr + lane * kIvfThrStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;

        constexpr int G = D >= 4 ? 4 : D;
        uint32_t tile_n = kIvfNoTile, imask_n = 0u;
        bool more_n = false;
        auto read_next = [&]() {
            while (itemw[32] != (uint32_t)(it + 1)) {
            }
            volatile uint32_t* w = itemw + 16 * ((it + 1) & 1);
            tile_n = __builtin_amdgcn_readfirstlane(w[2 * wave]);
            imask_n = __builtin_amdgcn_readfirstlane(w[2 * wave + 1]);
            more_n = __builtin_amdgcn_readfirstlane(w[0]) != kIvfNoTile;
        };
        if (active) {   // wave-uniform
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
                        acc = scan_step<0>(xb[i], qcur, acc);
                        qcur = qn;
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    }
#pragma unroll
                    for (int j = 0; j < G; ++j) xb[g + j] = *reinterpret_cast<const f32x4*>(src + 8 * (g + j));
                    __builtin_amdgcn_sched_group_barrier(0x020, G, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            read_next();
            // last D steps: the ring refills from this wave's tile of the next item (or once more from this one's start)
            const float* pn = tile_n != kIvfNoTile ? row_ptr(tile_n) : pc;
            {
                const f32x4* qs = qf + (size_t)(s0 + 1) * 64 + lane;
#pragma unroll
                for (int g = 0; g < D; g += G) {
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int i = g + j;
                        const f32x4 qn = qs[(i + 1 < D ? i : -1 - s0) * 64];
                        acc = scan_step<0>(xb[i], qcur, acc);
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
        } else {
            read_next();
            if (tile_n != kIvfNoTile) {
                pc = row_ptr(tile_n);
#pragma unroll
                for (int i = 0; i < D; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pc + 8 * i);
            }
        }

        // ---- ranking scores.  Lane (r, h): query r, tile rows (i & 3) + 8 (i >> 2) + 4 h
        const long long row0 = (long long)tile * kTileRows;
        float sc[16];
        if (p.l2) {   // kernel-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 xn = {0.f, 0.f, 0.f, 0.f};
                if (active) xn = *reinterpret_cast<const f32x4*>(p.xnorm + row0 + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[4 * g + j] = __builtin_fmaf(2.0f, acc[4 * g + j], -xn[j]) + 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = acc[i] + 0.0f;
        }

        // ---- filter + append: only the queries that probe this item's list
        const bool mine = active && ((imask >> r) & 1u);
        uint32_t pending = 0;
        if (mine) {
            const float t = thr[r];
            float m = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, sc[i]);
            if (m >= t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (sc[i] >= t) {
                        const uint32_t id = p.ids[row0 + (i & 3) + 8 * (i >> 2) + 4 * h];
                        const u64 key = make_key(sc[i], id);
                        if (id != kIvfPadId && key < key_ceil) {
                            const uint32_t slot = atomicAdd(&cnt[r], 1u);
                            if (slot < (uint32_t)C) {
                                keys[(size_t)r * C + slot] = key;
                            } else {
                                pending |= 1u << i;
                                flag[seq & 3] = 1;
                            }
                        }
                    }
                }
            }
        }

        // ---- overflow handling: synchronised compaction of every query, then retry (flat_kernels.hip.h)
        for (;;) {
            __syncthreads();
            const bool need = flag[seq & 3] != 0;
            if (tid == 0) flag[(seq + 2) & 3] = 0;
            ++seq;
            if (!need) break;
            {
                constexpr int NQW = kQT / NW;
                u64 kk[NQW][E];
                uint32_t nn[NQW];
#pragma unroll
                for (int j = 0; j < NQW; ++j) {
                    const int q = wave + NW * j;
                    nn[j] = min(cnt[q], (uint32_t)C);
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t idx = e * 64 + lane;
                        kk[j][e] = idx < nn[j] ? keys[(size_t)q * C + idx] : 0ull;
                    }
                }
                wave_sort_desc<E, NQW>(kk, lane);
#pragma unroll
                for (int j = 0; j < NQW; ++j) {
                    const int q = wave + NW * j;
                    if (nn[j] > 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int idx = e * 64 + lane;
                            if (idx < p.k) keys[(size_t)q * C + idx] = kk[j][e];
                            if (idx == p.k - 1 && nn[j] >= (uint32_t)p.k) {
                                const float kth = unord32((uint32_t)(kk[j][e] >> 32));
                                if (kth > thr[q]) {   // (a refreshed filter may already be above this workgroup's k-th best)
                                    thr[q] = kth;
                                    if (p.gthr) atomicMax(p.gthr + q * kIvfThrStride, (uint32_t)(kk[j][e] >> 32));
                                }
                            }
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
                            const uint32_t id = p.ids[row0 + (i & 3) + 8 * (i >> 2) + 4 * h];
                            const uint32_t slot = atomicAdd(&cnt[r], 1u);
                            if (slot < (uint32_t)C) {
                                keys[(size_t)r * C + slot] = make_key(sc[i], id);
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
        tile = tile_n;
        imask = imask_n;
        more = more_n;
        active = tile != kIvfNoTile;
    }

    // ---- epilogue: sort every buffer, emit k keys per query for this workgroup
    {
        constexpr int NQW = kQT / NW;
        u64 kk[NQW][E];
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = wave + NW * j;
            const uint32_t n = min(cnt[q], (uint32_t)C);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t idx = e * 64 + lane;
                kk[j][e] = idx < n ? keys[(size_t)q * C + idx] : 0ull;
            }
        }
        wave_sort_desc<E, NQW>(kk, lane);
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = wave + NW * j;
            u64* out = p.partial + ((size_t)q * gridDim.x + blockIdx.x) * p.k;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = e * 64 + lane;
                if (idx < p.k) out[idx] = kk[j][e];
            }
        }
    }
    // the last workgroup to leave puts the ticket counter back (launches of an index are stream-ordered)
    __syncthreads();
    if (tid == 0) {
        if (atomicAdd(p.done, 1u) == gridDim.x - 1) {
            *p.ticket = 0u;
            *p.done = 0u;
        }
    }
}

}  // namespace ragk
