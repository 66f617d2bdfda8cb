// bert_kernels.hip.h — gfx950 kernels of the BERT-family encoder used by the query embedder and the
// cross-encoder reranker.
//
// Replaces the arithmetic behind SentenceTransformer.encode (reference
// src/pipeline/components/embedding.py:127-133) and AutoModelForSequenceClassification.forward
// (reference src/pipeline/components/reranker.py:248-252).  Everything is fp32: the north star asks
// for the same doc ids as the CPU fp32 embedder, which rules out bf16/fp16 GEMMs; the GEMMs run on
// the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (same rate as the fp32 vector peak, but
// one VGPR per operand and the VALU left free for the epilogue).
//
// Sequences are PACKED: T = sum of lengths tokens, no padding rows; cu[s] .. cu[s+1] are the tokens
// of sequence s.  All weights are torch.nn.Linear layout: W[out][in], row-major.
//
//   embed_ln_kernel     word + position + type embedding, LayerNorm                (1 wave / token)
//   gemm_nt_kernel      C = A · Wᵀ + bias [+ residual] [activation], 128x128x32 tiles, fp32 MFMA
//   attention_mfma_kernel softmax(QKᵀ/√dh)V per (sequence, head, 32 rows) on fp32 MFMA (1 wave / block)
//   attention_kernel    the same on the VALU (kept as the reference implementation for tests)
//   splitk_bias_res_ln  y = LN(sum of split-K slabs + bias + residual)              (1 wave / token)
//   pool_kernel         mean / CLS pooling + optional L2 normalisation              (1 block / seq)
//   gather_rows_kernel  first-token rows for the classifier head
//   head_out_kernel     logits = x · Wcᵀ + bc (tiny N), plus sigmoid
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Act { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_TANH = 2, ACT_RELU = 3, ACT_TANH = 4 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
        case ACT_GELU_TANH: {
            const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}

// ---- embeddings + LayerNorm ------------------------------------------------------------------------
// One wave per token, each lane owns float4 chunks lane, lane+64, ... of the row (H <= 1024, H % 4 == 0).
constexpr int kMaxChunks = 4;  // 64 lanes * 4 chunks * 4 floats = 1024

struct EmbedParams {
    const int* ids;        // [T]
    const int* type_ids;   // [T] or null
    const int* cu;         // [nseq + 1]
    const float* word_emb; // [V][H]
    const float* pos_emb;  // [P][H]
    const float* type_emb; // [Tv][H] or null
    const float* ln_g;
    const float* ln_b;
    float* out;            // [T][H]
    int T, nseq, H, pos_offset, max_pos, vocab, type_vocab;
    float eps;
};

__device__ __forceinline__ int find_seq(const int* cu, int nseq, int t) {
    int lo = 0, hi = nseq;  // largest s with cu[s] <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// Normalise the row held in v[] (chunk j valid when lane + 64 j < H/4) and store it.
__device__ __forceinline__ void ln_store(f32x4 (&v)[kMaxChunks], int lane, int H, float eps, const float* g,
                                         const float* b, float* o) {
    const int nch = H >> 2;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j)
        if (lane + 64 * j < nch) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(sum) / (float)H;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j)
        if (lane + 64 * j < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = v[j][e] - mean;
                var = __builtin_fmaf(dlt, dlt, var);
            }
        }
    const float rstd = rsqrtf(wave_sum(var) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nch) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * ch);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * ch);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mean) * rstd * gg[e] + bb[e];
            *reinterpret_cast<f32x4*>(o + 4 * ch) = y;
        }
    }
}

__global__ __launch_bounds__(256) void embed_ln_kernel(const EmbedParams p) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= p.T) return;
    const int s = find_seq(p.cu, p.nseq, t);
    int pos = t - p.cu[s] + p.pos_offset;
    pos = pos < p.max_pos ? pos : p.max_pos - 1;
    int id = p.ids[t];
    id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
    int ty = p.type_ids ? p.type_ids[t] : 0;  // clamped like id and pos: a pair-encoded input on a one-type checkpoint
    ty = ty < 0 ? 0 : (ty >= p.type_vocab ? p.type_vocab - 1 : ty);  // must not read past the table
    const float* w = p.word_emb + (size_t)id * p.H;
    const float* pe = p.pos_emb + (size_t)pos * p.H;
    const float* te = p.type_emb ? p.type_emb + (size_t)ty * p.H : nullptr;
    const int nch = p.H >> 2;
    f32x4 v[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int ch = lane + 64 * j;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (ch < nch) {
            x = *reinterpret_cast<const f32x4*>(w + 4 * ch) + *reinterpret_cast<const f32x4*>(pe + 4 * ch);
            if (te) x += *reinterpret_cast<const f32x4*>(te + 4 * ch);
        }
        v[j] = x;
    }
    ln_store(v, lane, p.H, p.eps, p.ln_g, p.ln_b, p.out + (size_t)t * p.H);
}

// Epilogue shared by the GEMM kernels.  The MFMAs are issued as D = W_tile · A_tileᵀ, so the LANE is the
// token row and registers 4g..4g+3 are four CONSECUTIVE output features: bias, residual and the result
// move as float4 (4 wide stores per 32x32 tile instead of 16 scalar ones; the scalar epilogue was a
// fixed cost worth ~6 K-tiles per output tile at K = 384).
__device__ __forceinline__ void store_tile_rows(const f32x16& acc, int m, int n_base, int h, int M, int N,
                                                const float* bias, const float* R, int ldr, float* C, int ldc,
                                                int act, bool plain) {
    if (m >= M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n_base + 8 * g + 4 * h;
        if (n >= N) continue;
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (n + 3 < N) {
            if (!plain) {
                if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
                if (R) v += *reinterpret_cast<const f32x4*>(R + (size_t)m * ldr + n);
            }
            *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e < N) {
                    float x = v[e];
                    if (!plain) {
                        x = apply_act(x + (bias ? bias[n + e] : 0.f), act);
                        if (R) x += R[(size_t)m * ldr + n + e];
                    }
                    C[(size_t)m * ldc + n + e] = x;
                }
            }
        }
    }
}

// Epilogue of the 128 x 128 big-M kernels: through LDS, so that global memory sees whole rows.
// store_tile_rows writes what a lane holds — 16 bytes of one token row per instruction, 64 rows per
// wave-instruction, 4.6 KB apart: on M = 178k, N = 1152 those scattered stores cost 0.40 ms of a 1.29 ms
// GEMM (measured by removing them).  Here the four waves park one 64-row half of the tile in LDS
// (row stride 132 floats: conflict-free 16-byte writes), and every wave-instruction then moves two
// complete 512-byte rows: bias / activation / residual are applied on that side, so the residual is
// read coalesced as well.  `Cs` is LDS the K loop no longer needs (>= 64 * 132 floats); the caller
// has passed a barrier since its last read.
constexpr int WGLD = 132;
__device__ __forceinline__ void store_wg_tile_128(const f32x16 (&acc)[2][2], float* Cs, int tid, int wm, int wn, int r,
                                                  int h, int m0, int n0, int M, int N, const float* bias,
                                                  const float* R, int ldr, float* C, int ldc, int act) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a) __syncthreads();  // the first half has left LDS
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *reinterpret_cast<f32x4*>(&Cs[(wm * 32 + r) * WGLD + wn * 64 + b * 32 + 8 * g + 4 * h]) = v;
            }
        __syncthreads();
        const int c4 = tid & 31;             // float4 column of the 128-wide tile row
        const int n = n0 + 4 * c4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && n + 3 < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = (tid >> 5) + 8 * i;                        // 0..63: LDS row
            const int m = m0 + (lr >> 5) * 64 + a * 32 + (lr & 31);   // its token row
            if (m >= M || n >= N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * WGLD + 4 * c4]);
            if (n + 3 < N) {
                v += bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
                if (R) v += *reinterpret_cast<const f32x4*>(R + (size_t)m * ldr + n);
                *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < N) {
                        float x = apply_act(v[e] + (bias ? bias[n + e] : 0.f), act);
                        if (R) x += R[(size_t)m * ldr + n + e];
                        C[(size_t)m * ldc + n + e] = x;
                    }
                }
            }
        }
    }
}

// Big-M launches use a 1-D grid and this map from workgroup id to output tile.  Workgroups are dealt
// round-robin to the 8 XCDs, each with its own 4 MB L2: with the plain (x = n tile, y = m tile) grid the
// n tiles that share one 128-row slab of A land on eight different XCDs and A is fetched from memory
// eight times (measured on M = 178k, N = 1152, K = 384: the kernel took 1.0 ms with its MFMAs removed).
// Here XCD x owns m tiles x, x + 8, ... and sweeps all n tiles of one m tile before the next, so a slab
// of A is read from HBM once and W (a few MB) stays in that XCD's L2.
constexpr int kXcds = 8;
__host__ __device__ inline int xcd_grid(int M, int N, int BM, int BN) {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    return (tm + kXcds - 1) / kXcds * kXcds * tn;
}
__device__ __forceinline__ bool xcd_tile(int M, int N, int BM, int BN, int& m0, int& n0) {
    const int tn = (N + BN - 1) / BN;
    const int id = blockIdx.x, xcd = id % kXcds, j = id / kXcds;
    const int mt = (j / tn) * kXcds + xcd;
    m0 = mt * BM;
    n0 = (j % tn) * BN;
    return m0 < M;
}

// ---- GEMM: C[M][N] = A[M][K] · W[N][K]ᵀ (+ bias[N]) (+ R[M][N]) (act) -------------------------------------
// (64·TM) x (64·TN) x 32 tiles, 4 waves (2 x 2), each wave TM x TN MFMA tiles of 32 x 32:
//   TM = TN = 2: 128 x 128 tiles for big M (cross-encoder: M = all tokens of all pairs);
//   TM = TN = 1:  64 x 64 tiles for small M (32 queries ≈ 450 tokens) so the grid still fills 256 CUs,
//                 with split-K over blockIdx.z writing fp32 partial slabs (no epilogue) that
//                 splitk_bias_res_ln_kernel reduces in a fixed order.
// LDS rows are padded to 36 floats: a ds_read_b128 fragment read (16 lanes, 16 different rows, same
// column) then touches 16 distinct 4-bank groups — conflict free (36 r mod 64 is a bijection on r mod 16).
constexpr int GBK = 32, GLD = 36;

struct GemmParams {
    const float* A;    // [M][lda]
    const float* W;    // [N][ldw]
    const float* bias; // [N] or null
    const float* R;    // [M][ldr] residual or null
    float* C;          // [M][ldc]; with split-K: [splits][M][ldc] partial slabs
    int M, N, K;
    int lda, ldw, ldr, ldc;
    int act;
    int k_per_split;   // K range of one blockIdx.z (multiple of 32); == K when not split
};

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int NBUF = (TM * TN == 1) ? 2 : 1;  // small tiles double-buffer LDS: one barrier per K-tile
    __shared__ __attribute__((aligned(16))) float smem_[NBUF * (BM + BN) * GLD];  // one block: the epilogue reuses it
    float(*As_)[BM * GLD] = reinterpret_cast<float(*)[BM * GLD]>(smem_);
    float(*Ws_)[BN * GLD] = reinterpret_cast<float(*)[BN * GLD]>(smem_ + NBUF * BM * GLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if constexpr (TM * TN > 1) {  // big-M instantiation: 1-D XCD-aware grid
        if (!xcd_tile(p.M, p.N, BM, BN, m0, n0)) return;
    }
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const bool split = gridDim.z > 1;

    // staging map: rows x 8 float4 per tile; thread -> (row = tid/8 + 32 j, float4 col = tid%8)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const float* ag[2 * TM];
    const float* wg[2 * TN];
#pragma unroll
    for (int j = 0; j < 2 * TM; ++j) {
        int am = m0 + srow + 32 * j;
        am = am < p.M ? am : p.M - 1;
        ag[j] = p.A + (size_t)am * p.lda + kbeg + scol;
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
        int wr = n0 + srow + 32 * j;
        wr = wr < p.N ? wr : p.N - 1;
        wg[j] = p.W + (size_t)wr * p.ldw + kbeg + scol;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // Register staging.  Small tiles run two K-tiles ahead: while tile kt is multiplied out of LDS, tiles
    // kt+1 and kt+2 are in flight (16 MFMAs per wave per tile do not cover an HBM/L2 round trip at the
    // low occupancy small-M grids run at).
    const int nk = (kend - kbeg) / GBK;
    f32x4 ra0[2 * TM], rw0[2 * TN], ra1[2 * TM], rw1[2 * TN];
    auto load_tile = [&](f32x4 (&ra)[2 * TM], f32x4 (&rw)[2 * TN], int kt) {
        if (kt < nk) {
#pragma unroll
            for (int j = 0; j < 2 * TM; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ag[j] + (size_t)kt * GBK);
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) rw[j] = *reinterpret_cast<const f32x4*>(wg[j] + (size_t)kt * GBK);
        }
    };
    auto write_tile = [&](const f32x4 (&ra)[2 * TM], const f32x4 (&rw)[2 * TN], int buf) {
        float* As = As_[buf];
        float* Ws = Ws_[buf];
#pragma unroll
        for (int j = 0; j < 2 * TM; ++j) *reinterpret_cast<f32x4*>(&As[(srow + 32 * j) * GLD + scol]) = ra[j];
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) *reinterpret_cast<f32x4*>(&Ws[(srow + 32 * j) * GLD + scol]) = rw[j];
    };
    auto multiply_tile = [&](int buf) {
        const float* As = As_[buf];
        const float* Ws = Ws_[buf];
#pragma unroll
        for (int kg = 0; kg < GBK / 8; ++kg) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[a] = *reinterpret_cast<const f32x4*>(&As[(wm * 32 * TM + a * 32 + r) * GLD + kg * 8 + 4 * h]);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b] = *reinterpret_cast<const f32x4*>(&Ws[(wn * 32 * TN + b * 32 + r) * GLD + kg * 8 + 4 * h]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[b][t], af[a][t], acc[a][b], 0, 0, 0);
        }
    };
    if constexpr (TM * TN == 1) {
        // LDS double-buffered, registers two tiles ahead: per K-tile {write the tile fetched two tiles ago
        // into the free buffer, fetch the tile after next into those registers, multiply the current
        // buffer, ONE barrier}.  The write comes first so that its registers were loaded two multiplies
        // (not one) earlier — one multiply of 16 MFMAs per wave is shorter than an L2 round trip.
        load_tile(ra0, rw0, 0);
        load_tile(ra1, rw1, 1);
        write_tile(ra0, rw0, 0);
        load_tile(ra0, rw0, 2);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 1 < nk) {  // block-uniform
                write_tile(ra1, rw1, 1);
                load_tile(ra1, rw1, kt + 3);
            }
            multiply_tile(0);
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 2 < nk) {
                    write_tile(ra0, rw0, 0);
                    load_tile(ra0, rw0, kt + 4);
                }
                multiply_tile(1);
                __syncthreads();
            }
        }
    } else {
        // big tiles: single LDS buffer, one register set, two barriers per K-tile.  64 MFMAs per wave per
        // tile and 3 workgroups per CU already hide the staging; measured alternatives were slower —
        // a second register set (203 VGPRs) 0.65x, a second LDS buffer (74 KB, 2 workgroups/CU) 0.9x.
        load_tile(ra0, rw0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();  // previous tile fully consumed
            write_tile(ra0, rw0, 0);
            __syncthreads();
            load_tile(ra0, rw0, kt + 1);
            multiply_tile(0);
        }
    }

    // epilogue: lane r is token row .. + r, registers hold 16 of the 32 output features of each tile
    if constexpr (TM == 2 && TN == 2) {  // big M (never split): whole rows through LDS
        __syncthreads();
        store_wg_tile_128(acc, smem_, tid, wm, wn, r, h, m0, n0, p.M, p.N, p.bias, p.R, p.ldr, p.C, p.ldc, p.act);
    } else {
        float* Cz = p.C + (split ? (size_t)blockIdx.z * p.M * p.ldc : 0);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                store_tile_rows(acc[a][b], m0 + wm * 32 * TM + a * 32 + r, n0 + wn * 32 * TN + b * 32, h, p.M, p.N,
                                p.bias, p.R, p.ldr, Cz, p.ldc, p.act, split);
    }
}

// Half-precision-input GEMM (opt-in, big M only): A is fp32 in memory and rounded to fp16 while it is
// staged, W is an fp16 copy of the weights, products accumulate in fp32 on v_mfma_f32_32x32x16_f16
// (16x the fp32 matrix rate), bias / activation / residual / output stay fp32.  This is the precision
// the reference itself runs the reranker at on a GPU (reranker.py:91-93: float16 on cuda); the default
// build path stays fp32 because the parity bar is the CPU fp32 path.
// 128 x 128 x 64 tiles, 4 waves (2 x 2) of 64 x 64; LDS rows are 64 + 8 halfs = 144 B (the same
// 36-dword stride as the fp32 kernel: conflict-free 16-byte fragment reads).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int HBK = 64, HLD = 72;

struct GemmF16Params {
    const float* A;       // [M][lda] fp32
    const _Float16* W;    // [N][ldw] fp16
    const float* bias;
    const float* R;
    float* C;
    int M, N, K;          // K % 64 == 0
    int lda, ldw, ldr, ldc;
    int act;
};

__global__ __launch_bounds__(256) void gemm_nt_f16_kernel(const GemmF16Params p) {
    __shared__ __attribute__((aligned(16))) _Float16 smem_[2 * 128 * HLD];  // one block: the epilogue reuses it
    _Float16* const As = smem_;
    _Float16* const Ws = smem_ + 128 * HLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, 128, 128, m0, n0)) return;

    // A staging: 128 rows x 16 float4 -> thread (row = tid/16 + 16 j, float4 col = tid%16), 8 passes
    // W staging: 128 rows x  8 f16x8  -> thread (row = tid/8  + 32 j, f16x8  col = tid%8),  4 passes
    const int arow = tid >> 4, acol = (tid & 15) * 4;
    const int wrow = tid >> 3, wcol = (tid & 7) * 8;
    const float* ag[8];
    const _Float16* wg[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int am = m0 + arow + 16 * j;
        am = am < p.M ? am : p.M - 1;
        ag[j] = p.A + (size_t)am * p.lda + acol;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int wr = n0 + wrow + 32 * j;
        wr = wr < p.N ? wr : p.N - 1;
        wg[j] = p.W + (size_t)wr * p.ldw + wcol;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    f32x4 ra[8];
    f16x8 rw[4];
    const int nk = p.K / HBK;
#pragma unroll
    for (int j = 0; j < 8; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ag[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = *reinterpret_cast<const f16x8*>(wg[j]);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f16x4 hv = {(_Float16)ra[j][0], (_Float16)ra[j][1], (_Float16)ra[j][2], (_Float16)ra[j][3]};
            *reinterpret_cast<f16x4*>(&As[(arow + 16 * j) * HLD + acol]) = hv;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f16x8*>(&Ws[(wrow + 32 * j) * HLD + wcol]) = rw[j];
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ag[j] + (size_t)(kt + 1) * HBK);
#pragma unroll
            for (int j = 0; j < 4; ++j) rw[j] = *reinterpret_cast<const f16x8*>(wg[j] + (size_t)(kt + 1) * HBK);
        }
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {  // lane (r,h) feeds A[row r][16 ks + 8 h .. + 7]
            f16x8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                af[a] = *reinterpret_cast<const f16x8*>(&As[(wm * 64 + a * 32 + r) * HLD + 16 * ks + 8 * h]);
#pragma unroll
            for (int b = 0; b < 2; ++b)
                bf[b] = *reinterpret_cast<const f16x8*>(&Ws[(wn * 64 + b * 32 + r) * HLD + 16 * ks + 8 * h]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[b], af[a], acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
    store_wg_tile_128(acc, reinterpret_cast<float*>(smem_), tid, wm, wn, r, h, m0, n0, p.M, p.N, p.bias, p.R, p.ldr, p.C,
                      p.ldc, p.act);
}

// fp32-accurate GEMM on the bf16 matrix cores (default for big M).  gfx950 multiplies fp32 at 1/16 of
// its bf16 rate, so an fp32 GEMM is MFMA-bound long before HBM matters.  Every fp32 value is the exact
// sum of three bf16 numbers (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8
// significand bits); a product a*w expands into nine bf16 products of which the six of weight >= 2^-16
// are kept — hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid — each exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  The dropped terms are below 2^-24 of |a w|, i.e. below the rounding of
// the fp32 accumulation itself (measured against float64 on 256x384x256: 2.6e-9 relative, a hundred
// times smaller than an fp32 GEMM's own summation error).  Six bf16 MFMAs cover K = 16 in 192 cycles
// where eight fp32 MFMAs need 512.
// A is split while it is staged into LDS (VALU work issued between the MFMAs).  W is split once, at model
// creation, into an image already in MFMA-fragment order: for the 32-row tile nt, K-step ks (16
// columns) and plane pl, lane (r, h) finds its eight values W[32 nt + r][16 ks + 8 h ..] at
// (((nt * K/16 + ks) * 3 + pl) * 64 + 32 h + r) * 8 — so every wave loads its W operands straight from
// L2 into registers with one coalesced 1-KiB read per fragment and W never touches LDS (staging both
// operands made the kernel LDS-bound: 36 KB of LDS traffic per wave per K-tile against 48 MFMAs).
// 128 x 128 x 32 tiles, 4 waves (2 x 2) of 64 x 64; the A image is double-buffered (one barrier per
// K-tile); LDS rows are 32 + 8 bf16 = 80 B (20-dword stride: the 16 lanes of a ds_read_b128 phase hit
// 16 distinct 4-bank groups).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int XBK = 32, XLD = 40;
// experiment-only switches (scripts/exp/gemm_x6_bench.hip; never defined in the product build)
// RAGB_X6_TRACE=<workgroup>: that workgroup writes cycle stamps of its K loop (5 per iteration per wave)
#if defined(RAGB_X6_TRACE)
__device__ unsigned long long g_x6_trace[4 * 128 * 5];
#define RAGB_X6_STAMP(i) do { if (blockIdx.x == RAGB_X6_TRACE && lane == 0 && kt < 128) g_x6_trace[(wave * 128 + kt) * 5 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RAGB_X6_STAMP(i) do {} while (0)
#endif
#if defined(RAGB_X6_NO_MFMA)
#define RAGB_X6_MFMA(w, a, c) ([&] { asm volatile("" ::"v"(w), "v"(a)); return c; }())
#else
#define RAGB_X6_MFMA(w, a, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0)
#endif

__global__ void pack_x6_kernel(const float* W, int N, int K, int ldw, __bf16* out) {  // N % 32 == 0, K % 16 == 0
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    const float x = W[(size_t)n * ldw + k];
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    const int nt = n >> 5, r = n & 31, ks = k >> 4, h = (k >> 3) & 1, j = k & 7;
    const size_t frag = ((size_t)nt * (K / 16) + ks) * 3;  // fragment index of plane 0
    const size_t lane_off = (size_t)(32 * h + r) * 8 + j;
    out[(frag + 0) * 512 + lane_off] = hi;
    out[(frag + 1) * 512 + lane_off] = mid;
    out[(frag + 2) * 512 + lane_off] = lo;
}

struct GemmX6Params {
    const float* A;     // [M][lda] fp32
    const __bf16* Wx;   // fragment-order split image (pack_x6_kernel)
    const float* bias;
    const float* R;
    float* C;
    int M, N, K;        // N % 32 == 0, K % 32 == 0
    int lda, ldr, ldc;
    int act;
};

__global__ __launch_bounds__(256, 2) void gemm_nt_x6_kernel(const GemmX6Params p) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][128 * XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, 128, 128, m0, n0)) return;
    const int nk = p.K / XBK, nks = p.K / 16;

    // A staging: 128 rows x 8 float4 -> thread (row = tid/8 + 32 j, float4 col = tid%8), 4 passes
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const float* ag[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int am = m0 + arow + 32 * j;
        am = am < p.M ? am : p.M - 1;
        ag[j] = p.A + (size_t)am * p.lda + acol;
    }
    // this wave's two 32-row W tiles (clamped inside the matrix; stores are guarded)
    const bf16x8* wfrag[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int nt = (n0 >> 5) + wn * 2 + b;
        nt = nt < (p.N >> 5) ? nt : (p.N >> 5) - 1;
        wfrag[b] = reinterpret_cast<const bf16x8*>(p.Wx) + (size_t)nt * nks * 3 * 64 + lane;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    auto stage = [&](const f32x4 (&ra)[4], int buf, int j0, int j1) {  // split rows j0..j1-1 of this thread into LDS
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            bf16x4 hi, mid, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = ra[j][e];
                hi[e] = (__bf16)x;
                const float r1 = x - (float)hi[e];
                mid[e] = (__bf16)r1;
                lo[e] = (__bf16)(r1 - (float)mid[e]);
            }
            const int o = (arow + 32 * j) * XLD + acol;
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = hi;
            *reinterpret_cast<bf16x4*>(&As[buf][1][o]) = mid;
            *reinterpret_cast<bf16x4*>(&As[buf][2][o]) = lo;
        }
    };

    // Software pipeline of one K-tile iteration (cycle stamps of the plain loop showed ~4000 cycles per
    // iteration for 1536 cycles of MFMA even with the CU to itself: every phase waited for what it had
    // issued a moment earlier).  Here everything an iteration consumes was requested an iteration before:
    //   A rows      global -> registers one whole iteration ahead (two register sets, raA / raB)
    //   A fragments LDS -> registers one K-step ahead (af0 / af1)
    //   W fragments L2 -> registers one iteration ahead (refilled as soon as their MFMAs have issued)
    // and an iteration is two branch-free scheduling regions (the last tiles stage / refill clamped,
    // unused data) in which the hints spread loads, split arithmetic and LDS stores between the MFMAs:
    //   step 0: MFMAs (kt, K-step 0) | A loads of tile kt + 2 | fragments of (kt, 1) | split + store tile kt + 1
    //   barrier
    //   step 1: MFMAs (kt, K-step 1) | fragments of (kt + 1, 0) | W refills
    f32x4 raA[4], raB[4];
    bf16x8 wr[2][3][2];  // [K-step of the tile][plane][n tile]
    auto load_a = [&](f32x4 (&ra)[4], int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ag[j] + (size_t)kt * XBK);
    };
    auto read_af = [&](bf16x8 (&af)[3][2], int buf, int ks) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                af[pl][a] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * 64 + a * 32 + r) * XLD + 16 * ks + 8 * h]);
    };
    // (W plane, A plane) of the six kept terms, smallest first; the four output tiles take turns so that
    // dependent MFMAs on one accumulator are three instructions apart
    constexpr int kTerm[6][2] = {{0, 2}, {2, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    auto mfma24 = [&](const bf16x8 (&af)[3][2], int ks) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = RAGB_X6_MFMA(wr[ks][kTerm[t][0]][b], af[kTerm[t][1]][a], acc[a][b]);
    };
    auto refill_w = [&](int ks, int kt_next) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int b = 0; b < 2; ++b) wr[ks][pl][b] = wfrag[b][(size_t)((kt_next * 2 + ks) * 3 + pl) * 64];
    };

    load_a(raA, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) refill_w(ks, 0);
    stage(raA, 0, 0, 4);
    load_a(raB, min(1, nk - 1));   // tile 1: staged during iteration 0
    __syncthreads();
    bf16x8 af0[3][2], af1[3][2];
    read_af(af0, 0, 0);

    auto iteration = [&](int kt, int cur, const f32x4 (&ra_stage)[4], f32x4 (&ra_load)[4]) {
        const int kt1 = min(kt + 1, nk - 1), kt2 = min(kt + 2, nk - 1);
        // ---- step 0
        RAGB_X6_STAMP(0);
        load_a(ra_load, kt2);
        read_af(af1, cur, 1);
#ifndef RAGB_X6_NO_STAGE
        stage(ra_stage, cur ^ 1, 0, 4);
#endif
        mfma24(af0, 0);
#ifndef RAGB_X6_NO_WLOAD
        refill_w(0, kt1);
#endif
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // A loads of tile kt + 2
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // A fragments of this tile's second step
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);  // a share of the split arithmetic
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one LDS store
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);    // W refills of step 0
        RAGB_X6_STAMP(1);
        __syncthreads();
        RAGB_X6_STAMP(2);
        // ---- step 1
        read_af(af0, cur ^ 1, 0);
        mfma24(af1, 1);
#ifndef RAGB_X6_NO_WLOAD
        refill_w(1, kt1);
#endif
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // A fragments of the next tile's first step
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (g >= 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // W refills under the second half
        }
        RAGB_X6_STAMP(3);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        iteration(kt, 0, raB, raA);          // stages tile kt + 1 (in raB), loads tile kt + 2 into raA
        if (kt + 1 < nk) iteration(kt + 1, 1, raA, raB);
    }
    __syncthreads();  // every wave is done with the A image
#ifdef RAGB_X6_NO_STORE
    {
        float live = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) live += acc[a][b][i];
        if (live != 12345.678f) return;
    }
#endif
    // the loop's last barrier is behind every wave: the A image is free to carry the output tile
    store_wg_tile_128(acc, reinterpret_cast<float*>(&As[0][0][0]), tid, wm, wn, r, h, m0, n0, p.M, p.N, p.bias, p.R,
                      p.ldr, p.C, p.ldc, p.act);
}

// Small-M GEMM: 64 x (32 WN) x 64 tiles, 4 WN waves (WN = 2: 64 x 64 tiles, eight waves; WN = 3: 64 x 96, twelve).
// Waves 0 .. 2WN-1 (2 x WN over the tile) multiply columns 0-31 of every staged K-tile, the other 2WN waves
// columns 32-63, so each SIMD holds waves whose LDS waits and barrier waits hide behind each other's MFMAs
// (with one wave per SIMD half of each K-step was exposed latency); the two partial tiles are added through
// LDS at the end in a fixed order.  Same split-K / epilogue contract as gemm_nt_kernel.  Requires
// k_per_split % 64 == 0.
// WN = 3 exists for one shape class: at 450 tokens a GEMM with N = 3072 (bge-base FFN input projection) is
// 7 x 48 = 336 tiles of 64 x 64 on 256 CUs — the 80 CUs that get two take twice as long and everyone waits
// (33.9 us for 13 us of matrix work per tile); 7 x 32 = 224 tiles of 64 x 96 all run at once, each 1.5x the work.
constexpr int SBK = 64, SLD = 68;  // 68 r mod 64 = 4 r: 16 rows of a ds_read_b128 group hit 16 distinct slots

template <int WN>
__global__ __launch_bounds__(256 * WN) void gemm_nt_small_kernel(const GemmParams p) {
    constexpr int BN = 32 * WN, ROWS = 64 + BN, NT = 256 * WN;   // staged rows per K-tile: A rows 0..63, W rows 64..
    constexpr int NI = (ROWS * 16 + NT - 1) / NT;                // float4 per thread per K-tile (4; WN = 3: the last pass is partial)
    __shared__ __attribute__((aligned(16))) float T_[2][ROWS * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave / (2 * WN), wm = (wave % (2 * WN)) / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const bool split = gridDim.z > 1;

    // staging: item i of the K-tile = (row i / 16, float4 column i % 16); thread tid takes items tid + NT j
    const float* src[NI];
    int dst[NI];
    bool have[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int item = tid + NT * j, row = item >> 4, col = (item & 15) * 4;
        have[j] = (ROWS * 16) % NT == 0 || item < ROWS * 16;  // WN = 2: every pass is full, known at compile time
        const int rr = have[j] ? row : 0;
        if (rr < 64) {
            int am = m0 + rr;
            am = am < p.M ? am : p.M - 1;
            src[j] = p.A + (size_t)am * p.lda + kbeg + col;
        } else {
            int wr = n0 + rr - 64;
            wr = wr < p.N ? wr : p.N - 1;
            src[j] = p.W + (size_t)wr * p.ldw + kbeg + col;
        }
        dst[j] = rr * SLD + col;
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int nk = (kend - kbeg) / SBK;
    f32x4 r0[NI], r1[NI];
    auto load_tile = [&](f32x4 (&rg)[NI], int kt) {
        if (kt < nk) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (have[j]) rg[j] = *reinterpret_cast<const f32x4*>(src[j] + (size_t)kt * SBK);
        }
    };
    auto write_tile = [&](const f32x4 (&rg)[NI], int buf) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (have[j]) *reinterpret_cast<f32x4*>(&T_[buf][dst[j]]) = rg[j];
    };
    auto multiply_tile = [&](int buf) {
        const float* As = &T_[buf][(wm * 32 + r) * SLD + kh * 32 + 4 * h];
        const float* Ws = &T_[buf][(64 + wn * 32 + r) * SLD + kh * 32 + 4 * h];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + kg * 8);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(Ws + kg * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t], af[t], acc, 0, 0, 0);
        }
    };
    load_tile(r0, 0);
    load_tile(r1, 1);
    write_tile(r0, 0);
    load_tile(r0, 2);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) {  // block-uniform
            write_tile(r1, 1);
            load_tile(r1, kt + 3);
        }
        multiply_tile(0);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) {
                write_tile(r0, 0);
                load_tile(r0, kt + 4);
            }
            multiply_tile(1);
            __syncthreads();
        }
    }

    // add the two K-halves: the upper-half waves park their tile in LDS (the staging buffers are free after the
    // loop's last barrier), the lower-half waves add it to theirs (low half + high half, always in that order)
    float* xch = &T_[0][0];  // 2 WN waves x 16 regs x 64 lanes x 4 B = 8 WN KB <= sizeof(T_[0])
    const int slot = wave % (2 * WN);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xch[(slot * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += xch[(slot * 16 + i) * 64 + lane];

    float* Cz = p.C + (split ? (size_t)blockIdx.z * p.M * p.ldc : 0);
    store_tile_rows(acc, m0 + wm * 32 + r, n0 + wn * 32, h, p.M, p.N, p.bias, p.R, p.ldr, Cz, p.ldc, p.act, split);
}

// y[t] = LayerNorm( sum_z part[z][t] + bias + R[t] ): reduces split-K slabs in slab order (a fixed
// summation order), adds bias and residual, normalises.  One wave per token.  With splits == 1 and
// bias == R == null it is the plain LayerNorm of a finished GEMM output.  All slab loads of a chunk
// are issued before the first add (they are independent; a runtime-length loop would serialise them).
constexpr int kMaxSplitK = 8;

__global__ __launch_bounds__(256) void splitk_bias_res_ln_kernel(const float* part, int splits, const float* bias,
                                                                 const float* R, const float* g, const float* b,
                                                                 float* y, int T, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int nch = H >> 2;
    const size_t slab = (size_t)T * H;
    f32x4 v[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int ch = lane + 64 * j;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (ch < nch) {
            const float* src = part + (size_t)t * H + 4 * ch;
            f32x4 pz[kMaxSplitK];
#pragma unroll
            for (int z = 0; z < kMaxSplitK; ++z)
                pz[z] = z < splits ? *reinterpret_cast<const f32x4*>(src + z * slab) : x;
            f32x4 bb = x, rr = x;
            if (bias) bb = *reinterpret_cast<const f32x4*>(bias + 4 * ch);
            if (R) rr = *reinterpret_cast<const f32x4*>(R + (size_t)t * H + 4 * ch);
            x = pz[0];
#pragma unroll
            for (int z = 1; z < kMaxSplitK; ++z) x += pz[z];
            x += bb;
            x += rr;
        }
        v[j] = x;
    }
    ln_store(v, lane, H, eps, g, b, y + (size_t)t * H);
}

// ---- attention ---------------------------------------------------------------------------------------
// qkv: [T][3H] (q | k | v, each H = heads * DH).  One wave per (sequence, head, block of 64 query rows);
// lane = query row.  Keys/values stream through LDS in tiles of 64; online softmax in registers.
template <int DH>
__global__ __launch_bounds__(64) void attention_kernel(const float* qkv, const int* cu, float* ctx, int H, int heads,
                                                       float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * DH];
    __shared__ __attribute__((aligned(16))) float Vs[64 * DH];
    const int lane = threadIdx.x;
    const int s = blockIdx.z, head = blockIdx.y, qb = blockIdx.x;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    if (qb * 64 >= L) return;
    const int qi = qb * 64 + lane;
    const bool valid = qi < L;
    const size_t ld = (size_t)3 * H;
    const float* qrow = qkv + (size_t)(t0 + (valid ? qi : L - 1)) * ld + head * DH;
    float q[DH], o[DH];
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + c);
        q[c] = v[0] * scale; q[c + 1] = v[1] * scale; q[c + 2] = v[2] * scale; q[c + 3] = v[3] * scale;
        o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
    }
    float mx = -__builtin_inff(), den = 0.f;
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int nj = min(64, L - j0);
        __syncthreads();
        // stage K/V tile: 64 keys x DH floats each, lane-strided float4 copies
        for (int idx = lane; idx < nj * (DH / 4); idx += 64) {
            const int j = idx / (DH / 4), c = (idx % (DH / 4)) * 4;
            const float* base = qkv + (size_t)(t0 + j0 + j) * ld + head * DH + c;
            *reinterpret_cast<f32x4*>(&Ks[j * DH + c]) = *reinterpret_cast<const f32x4*>(base + H);
            *reinterpret_cast<f32x4*>(&Vs[j * DH + c]) = *reinterpret_cast<const f32x4*>(base + 2 * H);
        }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            // four independent partial dot products (a single 32..64-long fmaf chain is pure latency)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(&Ks[j * DH + c]);  // broadcast read
                s0 = __builtin_fmaf(q[c], kv[0], s0);
                s1 = __builtin_fmaf(q[c + 1], kv[1], s1);
                s2 = __builtin_fmaf(q[c + 2], kv[2], s2);
                s3 = __builtin_fmaf(q[c + 3], kv[3], s3);
            }
            const float sc = (s0 + s1) + (s2 + s3);
            if (sc > mx) {  // new running maximum: rescale what has been accumulated (rare after a few keys)
                const float alpha = expf(mx - sc);  // exp(-inf) = 0 on the first key
                den *= alpha;
#pragma unroll
                for (int c = 0; c < DH; ++c) o[c] *= alpha;
                mx = sc;
            }
            const float pj = expf(sc - mx);
            den += pj;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[j * DH + c]);
                o[c] = __builtin_fmaf(pj, vv[0], o[c]);
                o[c + 1] = __builtin_fmaf(pj, vv[1], o[c + 1]);
                o[c + 2] = __builtin_fmaf(pj, vv[2], o[c + 2]);
                o[c + 3] = __builtin_fmaf(pj, vv[3], o[c + 3]);
            }
        }
    }
    if (valid) {
        const float inv = 1.0f / den;
        float* orow = ctx + (size_t)(t0 + qi) * H + head * DH;
#pragma unroll
        for (int c = 0; c < DH; c += 4) {
            f32x4 v = {o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + c) = v;
        }
    }
}

// ---- attention on the matrix pipe -------------------------------------------------------------------
// One wave per (sequence, head, 32 query rows); fp32 MFMA for both products, online softmax between
// them, everything arranged so that the LANE is the query:
//   Sᵀ[key][query] = K·Qᵀ      A = K fragment (LDS), B = Q fragment (registers, pre-scaled by 1/√dh)
//                               -> lane (r,h) holds Sᵀ for query r and the 16 keys 8(i>>2)+(i&3)+4h
//   softmax over keys           = over the lane's 16 registers, its partner lane r+32, and key tiles
//   Oᵀ[dh][query] += Vᵀ·Pᵀ      B = the probabilities exactly as they sit in the accumulator (register i of
//                               lane (r,h) IS Pᵀ[key(i,h)][r], the k-slot layout MFMA wants), A = V read
//                               from LDS by column -> no shuffle, no transpose of P.
// The VALU version this replaces issued ~200 vector instructions per key per wave and ran at 40 % of the
// VALU issue rate (12 % of cross-encoder time); here a 32-key tile is 16..32 + 16..32 MFMAs.
// first_only: only each sequence's FIRST token is a query and its output row goes to ctx[s] (one row per
// sequence) — the last layer of a model whose output reads first tokens only (CLS pooling, classifier head).
template <int DH>
__global__ __launch_bounds__(64) void attention_mfma_kernel(const float* qkv, const int* cu, float* ctx, int H, int heads,
                                                            float scale, int first_only = 0) {
    constexpr int KLD = DH + 4;  // padded K rows: conflict-free ds_read_b128 fragments
    __shared__ __attribute__((aligned(16))) float Ks[32 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[32 * DH];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int s = blockIdx.z, head = blockIdx.y, qb = blockIdx.x;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    if (qb * 32 >= L) return;
    const int qidx = qb * 32 + r;
    const bool qvalid = first_only ? qidx == 0 : qidx < L;
    const size_t ld = (size_t)3 * H;

    // Q fragments: lane (r,h) holds Q[qidx][8 s + 4 h .. + 3] for every 8-column step s
    f32x4 qf[DH / 8];
    {
        const float* qrow = qkv + (size_t)(t0 + (qvalid ? qidx : L - 1)) * ld + head * DH + 4 * h;
#pragma unroll
        for (int st = 0; st < DH / 8; ++st) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qrow + 8 * st);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = qvalid ? v[e] * scale : 0.f;
            qf[st] = v;
        }
    }
    f32x16 oT[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oT[dt][i] = 0.f;
    float mx = -__builtin_inff(), den = 0.f;

    const int n_kt = (L + 31) / 32;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * 32, nk = min(32, L - k0);
        __syncthreads();  // previous tile's LDS reads are done (one wave: this is just a waitcnt)
        for (int idx = lane; idx < 32 * (DH / 4); idx += 64) {
            const int j = idx / (DH / 4), c = (idx % (DH / 4)) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (j < nk) {
                const float* base = qkv + (size_t)(t0 + k0 + j) * ld + head * DH + c;
                kv = *reinterpret_cast<const f32x4*>(base + H);
                vv = *reinterpret_cast<const f32x4*>(base + 2 * H);
            }
            *reinterpret_cast<f32x4*>(&Ks[j * KLD + c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[j * DH + c]) = vv;
        }
        __syncthreads();

        // Sᵀ tile
        f32x16 sT;
#pragma unroll
        for (int i = 0; i < 16; ++i) sT[i] = 0.f;
#pragma unroll
        for (int st = 0; st < DH / 8; ++st) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[r * KLD + 8 * st + 4 * h]);
#pragma unroll
            for (int t = 0; t < 4; ++t) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[st][t], sT, 0, 0, 0);
        }
        // mask keys beyond the sequence, running max over this lane's keys and its partner's
        float tmax = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
            if (key >= nk) sT[i] = -__builtin_inff();
            tmax = fmaxf(tmax, sT[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mx, tmax);  // finite: every tile has at least one valid key
        const float alpha = expf(mx - mnew);  // 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sT[i] = expf(sT[i] - mnew);  // masked keys: exp(-inf) = 0
            psum += sT[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * alpha + psum;
        mx = mnew;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oT[dt][i] *= alpha;
        // Oᵀ += Vᵀ Pᵀ : MFMA i consumes the key pair (key(i,0), key(i,1)) = registers i of the two halves
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const float vf = Vs[key * DH + dt * 32 + r];
                oT[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sT[i], oT[dt], 0, 0, 0);
            }
        }
    }
    if (qvalid) {
        const float inv = 1.0f / den;
        float* orow = ctx + (size_t)(first_only ? s : t0 + qidx) * H + head * DH;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are dh rows 8g + 4h + 0..3: one 16-byte store
                f32x4 v = {oT[dt][4 * g] * inv, oT[dt][4 * g + 1] * inv, oT[dt][4 * g + 2] * inv, oT[dt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * g + 4 * h) = v;
            }
    }
}

// ---- pooling ----------------------------------------------------------------------------------------
// mode 0: mean over the sequence's tokens, mode 1: first token, mode 2: x holds one row per sequence.
// normalize: divide by the L2 norm
// (torch.nn.functional.normalize: x / max(||x||, 1e-12)).
__global__ __launch_bounds__(256) void pool_kernel(const float* x, const int* cu, float* out, int H, int mode,
                                                   int normalize) {
    __shared__ float red[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    // mode 2: x is already one row per sequence ([nseq][H], the last layer ran on first tokens only)
    const int t0 = mode == 2 ? s : cu[s], L = mode == 2 ? 1 : cu[s + 1] - t0;
    float v[4];  // H <= 1024: 4 columns per thread
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        float a = 0.f;
        if (c < H) {
            if (mode == 0) {
                for (int t = 0; t < L; ++t) a += x[(size_t)(t0 + t) * H + c];
                a /= (float)L;
            } else {
                a = x[(size_t)t0 * H + c];
            }
        }
        v[j] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(tot), 1e-12f) : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        if (c < H) out[(size_t)s * H + c] = v[j] * inv;
    }
}

__global__ void gather_rows_kernel(const float* x, const int* cu, float* out, int nseq, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * H) return;
    const int s = i / H, c = i - s * H;
    out[i] = x[(size_t)cu[s] * H + c];
}

// logits[s][j] = x[s] · Wc[j] + bc[j]; one wave per (s, j).  probs = sigmoid(logits) when requested.
__global__ __launch_bounds__(64) void head_out_kernel(const float* x, const float* Wc, const float* bc, float* logits,
                                                      float* probs, int H, int n_labels) {
    const int s = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    float a = 0.f;
    for (int c = lane; c < H; c += 64) a = __builtin_fmaf(x[(size_t)s * H + c], Wc[(size_t)j * H + c], a);
    a = wave_sum(a);
    if (lane == 0) {
        const float z = a + (bc ? bc[j] : 0.f);
        logits[(size_t)s * n_labels + j] = z;
        if (probs) probs[(size_t)s * n_labels + j] = 1.0f / (1.0f + expf(-z));
    }
}

}  // namespace ragb

#include "gemm_wl.hip.h"  // big-batch GEMM with both operands through LDS-DMA (uses the types and helpers above)
#include "gemm_w6.hip.h"  // the default mode's big-batch GEMM on 128 x 192 tiles
#include "bert_lnf.hip.h"  // small kernels of the query encoder's folded LayerNorms
#include "gemm_wt.hip.h"  // the same over activations in MFMA-fragment order (RAG_GEMM_F16 big-batch path)
#include "bert_tiled.hip.h"  // embedding, LayerNorm, attention, pooling over that layout
#include "ffn_fused_t16.hip.h"  // the feed-forward block as one kernel on that layout (hidden 384)
