// gemm_wl.hip.h — big-batch GEMM of the transformer layers, both operands through LDS-DMA.
//
//   C[M][N] = A[M][K] · W[N][K]ᵀ (+ bias[N]) (activation) (+ R[M][N]),   A, C, R fp32 in memory
//
// Replaces gemm_nt_x6_kernel / gemm_nt_f16_kernel (bert_kernels.hip.h) for M > 1024 — the cross-encoder's
// GEMMs (reference src/pipeline/components/reranker.py:248-252: model(**inputs) over all (query, doc) pairs of a batch).
// Same arithmetic, same summation order per output element, so the results are bit-identical to those kernels;
// what changes is how the operands reach the matrix cores.
//
// What bounded the old kernels (41 % MFMA utilisation, ablations and cycle stamps): operand
// delivery.  A went global -> VGPR -> split (VALU) -> LDS (ds_write, the slow LDS path) and was read back by the two
// waves that share its rows; W fragments went L2 -> VGPR in every one of the two waves that share them; eight waves
// with 16-20 KB of loads in flight each are at the edge of what a CU needs (~160 KB) to cover L2/HBM latency.  Here:
//
//   * BOTH operands arrive by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no VALU on the way
//     in, and — with no ordinary register load in the K loop — hipcc leaves the DMA ring alone (it would drain it with
//     a vmcnt(0) at every use of a plain load).  A ring of NS stages keeps NS - 1 stages in flight or landed.
//   * A wave owns 32 token ROWS x all 128 columns of the 128 x 128 tile.  Its A rows are loaded by itself, read by
//     itself and split (fp32 -> three bf16 planes) in ITS registers: no other wave repeats that work, and A needs no
//     cross-wave synchronisation at all.  A's LDS image is the raw fp32 K-slice, 64 bytes per row, 16-byte chunks
//     XOR-swizzled by (row >> 2) & 3 on the DMA's source side so that the fragment reads (32 bytes per lane, lane =
//     row) are bank-conflict free.
//   * W is stored once, at model creation, in MFMA-fragment order (pack_x6_kernel / pack_f16_frag_kernel): a fragment
//     is 1 KiB, lane-linear — exactly the shape an LDS-DMA instruction writes — and is read back with one
//     conflict-free ds_read_b128 per lane.  The four waves share the tile's W fragments through LDS instead of each
//     pulling them from L2.
//   * one raw s_barrier per stage; the wait in front of it is a COUNTED vmcnt that leaves the younger stages in flight.
//
// MODE 0 (RAG_GEMM_F32): every fp32 operand is the exact sum of three bf16 numbers; the six products of weight
//   >= 2^-16 go through v_mfma_f32_32x32x16_bf16 (see gemm_nt_x6_kernel for the error analysis).
// MODE 2 (experiment, scripts/exp/gemm_wl_bench.hip): every fp32 operand as TWO fp16 numbers, hi = fp16(x) and
//   lo = fp16((x - hi) 2^11) — 22 significand bits — and the three products hi hi, lo hi, hi lo on
//   v_mfma_f32_32x32x16_f16 (the cross terms in an accumulator of their own, scaled by 2^-11 at the end): half the
//   MFMAs and two thirds of the operand bytes of MODE 0, relative error ~2^-22 per operand instead of ~2^-24, and
//   fp16's range (|x| < 65504) instead of fp32's.
// MODE 1 (RAG_GEMM_F16): A rounded to fp16 on the way to the matrix core, W an fp16 image, one
//   v_mfma_f32_32x32x16_f16 per fragment pair — the precision the reference runs its reranker at on a GPU
//   (reranker.py:91-93).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

struct GemmWlParams {
    const float* A;      // [M][lda] fp32
    const void* Wimg;    // fragment-order image: MODE 0 three bf16 planes (pack_x6_kernel), MODE 1 one fp16 plane
    const float* bias;   // [N] or null
    const float* R;      // [M][ldr] residual or null
    float* C;            // [M][ldc]
    int M, N, K;         // N % 32 == 0, K % (16 KS) == 0
    int lda, ldr, ldc;
    int act;
    uint32_t* range_flag;  // MODE 2: set to 1 when an element of A is outside fp16's range (|x| >= 65504); may be null
    // ---- LayerNorm folded into its consumers, big-batch form (wl_epilogue_lnf; all null / 0 = the plain epilogue).  Block
    // statistics here are per (row, 128-column tile).
    const float2* rs_in;   // [M] (mean, rstd) of the rows whose LayerNorm is pending (A's rows for form A, Ry's for B), finished
                           // by lnf_finalize_kernel from the producer's tile statistics; the kernel DMAs its 128 rows' worth
                           // into LDS with its first stage, so no load sits in front of the epilogue
    const float* fold_s;   // form A: s[n] = sum_k W'[n][k]; C = act(rstd (acc - mean s[n]) + bias[n]), bias = b + W beta
    const float* Ry;       // form B: residual BEFORE its LayerNorm [M][ldr]: R = (Ry - mean) rstd ln_g[n] + ln_b[n]
    const float* ln_g;
    const float* ln_b;
    float2* ts_out;        // form B: [M][N / 128] statistics of the rows written (C = acc + bias + R, no LayerNorm applied)
};

// W fp32 [N][K] -> fp16 image in MFMA-fragment order: fragment (nt, ks) holds, for lane (r, h), the eight values
// W[32 nt + r][16 ks + 8 h ..] at ((nt * K/16 + ks) * 64 + 32 h + r) * 8.
__global__ void pack_f16_frag_kernel(const float* W, int N, int K, int ldw, _Float16* out) {  // N % 32 == 0, K % 16 == 0
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    const int nt = n >> 5, r = n & 31, ks = k >> 4, h = (k >> 3) & 1, j = k & 7;
    out[(((size_t)nt * (K / 16) + ks) * 64 + 32 * h + r) * 8 + j] = (_Float16)W[(size_t)n * ldw + k];
}

// W fp32 [N][K] -> TWO fp16 planes in MFMA-fragment order (MODE 2): hi = fp16(w), lo = fp16((w - hi) * 2^11); fragment
// (nt, ks, plane) at ((nt * K/16 + ks) * 2 + plane) * 512 halves, lane (r, h)'s eight values at (32 h + r) * 8.
constexpr float kX3Scale = 2048.0f;        // 2^11: the low plane is stored scaled so that it stays in fp16's normal range
constexpr float kX3Unscale = 1.0f / 2048.0f;
constexpr float kF16Max = 65504.0f;
// colscale (may be null): the image holds W[n][k] * colscale[k] — a LayerNorm's gamma folded into the GEMM that consumes
// its output (see "LayerNorm folded into its consumers" below).
__global__ void pack_f16x2_frag_kernel(const float* W, int N, int K, int ldw, _Float16* out, uint32_t* range_flag,
                                       const float* colscale = nullptr) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    const int nt = n >> 5, r = n & 31, ks = k >> 4, h = (k >> 3) & 1, j = k & 7;
    const float x = colscale ? W[(size_t)n * ldw + k] * colscale[k] : W[(size_t)n * ldw + k];
    if (range_flag && fabsf(x) >= kF16Max) *range_flag = 1u;   // (inf / huge: the caller keeps the split-bf16 image)
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)((x - (float)hi) * kX3Scale);
    const size_t frag = ((size_t)nt * (K / 16) + ks) * 2;
    const size_t lane_off = (size_t)(32 * h + r) * 8 + j;
    out[(frag + 0) * 512 + lane_off] = hi;
    out[(frag + 1) * 512 + lane_off] = lo;
}

#if defined(RAGB_WL_NO_MFMA)   // experiment builds only (scripts/exp/gemm_wl_bench.hip)
#define RAGB_WL_MFMA_BF16(w, a, c) ([&] { asm volatile("" ::"v"(w), "v"(a)); return c; }())
#define RAGB_WL_MFMA_F16(w, a, c) ([&] { asm volatile("" ::"v"(w), "v"(a)); return c; }())
#else
#define RAGB_WL_MFMA_BF16(w, a, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0)
#define RAGB_WL_MFMA_F16(w, a, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0)
#endif

// One LDS-DMA instruction: 64 lanes x 16 bytes, each lane's own global address -> LDS at `lds` + 16 * lane
// (`lds` must be wave-uniform).  (A helper, not a lambda: a lambda returning an address-space-qualified pointer
// silently drops the kernel's host stub under hipcc 7.2.)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int MODE, int KS, int NS>
struct WlGeom {
    static constexpr int PL = MODE == 0 ? 3 : (MODE == 2 ? 2 : 1);   // W planes
    static constexpr int A_STAGE = KS * 128 * 64;             // bytes: KS K-steps x 128 rows x 16 fp32
    static constexpr int W_STAGE = 4 * KS * PL * 1024;        // bytes: 4 column tiles x KS x PL fragments of 1 KiB
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int RING = NS * STAGE;
    static constexpr int EPI = 4 * 16 * 132 * 4;              // epilogue: per wave 16 rows x 132 floats
    static constexpr int ROWST = RING > EPI ? RING : EPI;     // behind both: 4 waves x 32 rows x (mean, rstd), DMA-filled
    static constexpr int LDS = ROWST + 4 * 32 * 8;
    static constexpr int G = 2 * KS + KS * PL;                // LDS-DMA instructions per wave per stage
};

// ---- block statistics of rows whose LayerNorm is applied by their consumers ("LayerNorm folded into its consumers", below)
// `blkw` = columns per block: 32 on the query-encoder path (a wave's 32 x 32 accumulator tile), 128 on the big-batch path
// (a row of a 128 x 128 workgroup tile).
__device__ __forceinline__ void lnf_row_stats(const float2* ts, int nblk, float eps, float& mean, float& rstd, float blkw = 32.f) {
    float sm = 0.f, sM2 = 0.f;
    for (int b = 0; b < nblk; ++b) {
        const float2 q = ts[b];
        sm += q.x;
        sM2 += q.y;
    }
    mean = sm / (float)nblk;
    float dev = 0.f;
    for (int b = 0; b < nblk; ++b) {
        const float dl = ts[b].x - mean;
        dev = __builtin_fmaf(dl, dl, dev);
    }
    rstd = rsqrtf((sM2 + blkw * dev) / (blkw * (float)nblk) + eps);
}

// (mean, M2) of the 32 values of one row that lanes (r, 0) and (r, 1) hold between them, 16 each
__device__ __forceinline__ float2 lnf_block_stats(const f32x16& v) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    const float m16 = s * (1.f / 16.f);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float dl = v[i] - m16;
        m2 = __builtin_fmaf(dl, dl, m2);
    }
    const float om = __shfl_xor(m16, 32), om2 = __shfl_xor(m2, 32);
    const float dm = m16 - om;
    return make_float2(0.5f * (m16 + om), m2 + om2 + 8.f * dm * dm);   // n_a n_b / (n_a + n_b) = 8
}

__host__ __device__ inline int ws_grid(int M, int N, int NB, int splits) {
    const int mt = (M + 63) / 64, nct = (N + 64 * NB - 1) / (64 * NB);
    return (nct * splits * mt + 7) / 8 * 8;
}


// Epilogue of the LDS-DMA GEMMs.  Lane (r, h) holds, for its token row, features 32 b + 8 g + 4 h + 0..3 in
// acc[b][4g..4g+3].  Each wave transposes its own 32 x 128 tile through its own 16-row LDS strip (no cross-wave
// traffic, so no barrier: a wave's LDS operations execute in order), two halves of 16 rows; every store instruction
// then writes two complete 512-byte rows, and bias / activation / residual are applied on that side (coalesced too).
// The caller has passed a barrier since the last read of the ring, which this reuses.
__device__ __forceinline__ void wl_epilogue(const GemmWlParams& p, const f32x16 (&acc)[4], char* smem, int m0, int n0,
                                            int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    float* Cs = reinterpret_cast<float*>(smem) + wave * 16 * 132;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((r >> 4) == hh) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(&Cs[(r & 15) * 132 + b * 32 + 8 * g + 4 * h]) = v;
                }
        }
        const int c4 = lane & 31;
        const int n = n0 + 4 * c4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = (lane >> 5) + 2 * i;                       // row of the strip
            const int m = m0 + wave * 32 + hh * 16 + lr;
            f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * 132 + 4 * c4]);
            if (m < p.M && n < p.N) {                                  // N % 32 == 0: a float4 is inside or outside
                v += bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
#if defined(RAGB_WL_NO_STORE)   // experiment build: keep the values live, skip the store
                if (v[0] == 12345.678f) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
#else
                *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
#endif
            }
        }
    }
}

// The same epilogue with a LayerNorm folded in (big-batch form of gemm_nt_ws_kernel's forms A and B, see "LayerNorm folded
// into its consumers" below): after the transposition a half wave holds one complete 128-column row of the tile, four
// values per lane, so
//   form A (p.fold_s): C = act(rstd (acc - mean s[n]) + bias[n]) — the input rows' LayerNorm, its gamma folded into the image;
//   form B (p.ts_out): C = acc + bias + R with R plain (p.R) or the LayerNorm of p.Ry recomputed per element, and the
//           (mean, M2) of each written row's 128 columns left in ts_out — two 32-lane reductions per row, eight rows'
//           chains side by side.
// Per-row (mean, rstd) of the pending LayerNorm come from lnf_finalize_kernel's array, DMA'd into LDS behind the ring by the
// kernel's first instructions (a plain load in front of the epilogue cost 3 us per tile: two dependent round trips).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {   // v of the lane the DPP control selects (all rows, all banks)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes of this lane's half; every lane gets the sum
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
    v += dpp_f32<0x141>(v);   // row_half_mirror: ... of its 8
    v += dpp_f32<0x140>(v);   // row_mirror: ... of its row of 16
    return v + __shfl_xor(v, 16);   // the other row of the half (one ds_bpermute instead of five)
}

__device__ __forceinline__ void wl_epilogue_lnf(const GemmWlParams& p, const f32x16 (&acc)[4], char* smem, const float2* rowst_all,
                                                int m0, int n0, int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    float* Cs = reinterpret_cast<float*>(smem) + wave * 16 * 132;
    const float2* rowst = rowst_all + wave * 32;   // [32] (mean, rstd) of this wave's rows (DMA-filled at the kernel's start)
    const int c4 = lane & 31;
    const int n = n0 + 4 * c4;
    const bool n_ok = n < p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = bv, gv = bv, ev = bv;
    if (p.bias && n_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.fold_s && n_ok) sv = *reinterpret_cast<const f32x4*>(p.fold_s + n);
    if (p.Ry && n_ok) {
        gv = *reinterpret_cast<const f32x4*>(p.ln_g + n);
        ev = *reinterpret_cast<const f32x4*>(p.ln_b + n);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((r >> 4) == hh) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(&Cs[(r & 15) * 132 + b * 32 + 8 * g + 4 * h]) = v;
                }
        }
        f32x4 vals[8];
        float rsum[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = (lane >> 5) + 2 * i;                       // row of the strip
            const int m = m0 + wave * 32 + hh * 16 + lr;
            f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * 132 + 4 * c4]);
            const bool ok = m < p.M && n_ok;
            float2 st = make_float2(0.f, 1.f);
            if (p.rs_in) st = rowst[hh * 16 + lr];
            if (p.fold_s) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = apply_act(__builtin_fmaf(st.y, v[e], __builtin_fmaf(-st.x * st.y, sv[e], bv[e])), p.act);
            } else {
                v += bv;
                if (ok) {
                    if (p.Ry) {
                        const f32x4 y4 = *reinterpret_cast<const f32x4*>(p.Ry + (size_t)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += __builtin_fmaf((y4[e] - st.x) * st.y, gv[e], ev[e]);
                    } else if (p.R) {
                        v += *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
                    }
                }
            }
            if (ok) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
            if (!n_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};   // (N % 128 != 0 only: such a tile's statistics would be short — the host does not ask)
            vals[i] = v;
            rsum[i] = (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (p.ts_out) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rsum[i] = half_wave_sum(rsum[i]) * (1.f / 128.f);   // the row's mean over the tile
            float m2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dl = vals[i][e] - rsum[i];
                    a = __builtin_fmaf(dl, dl, a);
                }
                m2[i] = a;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) m2[i] = half_wave_sum(m2[i]);
            if (c4 == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = m0 + wave * 32 + hh * 16 + (lane >> 5) + 2 * i;
                    if (m < p.M) p.ts_out[(size_t)m * (p.N >> 7) + (n0 >> 7)] = make_float2(rsum[i], m2[i]);
                }
            }
        }
    }
}

// fp32 -> the three bf16 planes of the exact split (hi, mid, lo), eight values of a lane's fragment
__device__ __forceinline__ void wl_split3(const f32x4 x0, const f32x4 x1, bf16x8 (&af)[3]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
#if defined(RAGB_WL_NO_SPLIT)   // experiment build: no split arithmetic (wrong values, same data flow)
        af[0][e] = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(x) >> 16));
        af[1][e] = af[0][e];
        af[2][e] = af[0][e];
#else
        const __bf16 hi = (__bf16)x;
        const float r1 = x - (float)hi;
        const __bf16 mid = (__bf16)r1;
        af[0][e] = hi;
        af[1][e] = mid;
        af[2][e] = (__bf16)(r1 - (float)mid);
#endif
    }
}

// PIPE (KS == 1 only): the A fragment of stage st + 1 is read and converted DURING stage st's MFMAs — its rows are this
// wave's own, so a counted vmcnt of its own is all the synchronisation that read needs — instead of in front of its
// own stage's MFMAs, where nothing of this wave's covers the ~50 conversion instructions.
template <int MODE, int KS, int NS, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_wl_kernel(const GemmWlParams p) {
    using Geo = WlGeom<MODE, KS, NS>;
    constexpr int PL = Geo::PL;
    static_assert(NS >= 3, "the ring needs a stage in flight beside the one being read and the one being refilled");
    static_assert(!PIPE || KS == 1, "the pipelined form keeps one K-step's A fragment in registers");
    static_assert(!PIPE || MODE != 2, "MODE 2 has no pipelined form");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ALL of the kernel's LDS (a second object beside a
                                                                  // DMA-filled array makes hipcc drain the ring)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, 128, 128, m0, n0)) return;
    const int nks = p.K / 16;
    const int n_stages = nks / KS;

    // ---- LDS-DMA sources.  A: instruction q of a stage covers K-step q / 2 ... for THIS wave's 32 rows: two
    // instructions of 16 rows x 64 bytes per K-step; lane l -> row 16 q' + l / 4, LDS slot l % 4, source chunk
    // slot ^ ((row >> 2) & 3).  W: this wave brings column tile `wave` of the workgroup's four.
    const char* a_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);          // row of the tile
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;                               // rows past M: valid memory, results dropped
        a_src[q] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16;
    }
    int nt_w = (n0 >> 5) + wave;
    nt_w = nt_w < (p.N >> 5) ? nt_w : (p.N >> 5) - 1;
    const char* w_src = static_cast<const char*>(p.Wimg) + (size_t)nt_w * nks * PL * 1024 + lane * 16;

    auto issue_stage = [&](int st) {   // stage number st (clamped by the caller) into slot st % NS
        char* slot = smem + (st % NS) * Geo::STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t koff = (size_t)(st * KS + ks) * 64;       // bytes along K in a row of A
#pragma unroll
            for (int q = 0; q < 2; ++q)
                glds16(a_src[q] + koff, slot + ks * 8192 + (wave * 32 + q * 16) * 64);
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                glds16(w_src + ((size_t)(st * KS + ks) * PL + pl) * 1024,
                       slot + Geo::A_STAGE + ((wave * KS + ks) * PL + pl) * 1024);
        }
    };

    f32x16 acc[4];
    f32x16 accx[MODE == 2 ? 4 : 1];   // MODE 2: the cross terms (scaled by 2^11)
    float amax = 0.f;                 // MODE 2: largest |a| this lane has converted
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int b = 0; b < (MODE == 2 ? 4 : 1); ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) accx[b][i] = 0.f;

    // this lane's two 16-byte chunks of its row's K-slice: chunks 2h and 2h + 1, swizzled
    const int my_row = wave * 32 + r;
    const int sw = (my_row >> 2) & 3;
    const int a_off0 = my_row * 64 + (((2 * h) ^ sw) << 4), a_off1 = my_row * 64 + (((2 * h + 1) ^ sw) << 4);

    if (p.rs_in) {   // (mean, rstd) of this wave's 32 rows: 64 dwords, one 4-byte DMA per lane, the oldest of the wave's DMAs
        int mrow = m0 + wave * 32 + (lane >> 1);
        mrow = mrow < p.M ? mrow : p.M - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(p.rs_in + mrow) + (lane & 1)),
                                         (__attribute__((address_space(3))) void*)(smem + Geo::ROWST + wave * 256), 4, 0, 0);
    }
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_stage(st < n_stages ? st : n_stages - 1);

    // (W plane, A plane) of the six kept terms, smallest first — the order gemm_nt_x6_kernel adds them in
    constexpr int kTerm[6][2] = {{0, 2}, {2, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};

    auto wait_stages_in_flight = [&]() {   // own DMAs of all but the NS - 2 youngest stages have landed
        if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::G) : "memory");
        else if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Geo::G) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * Geo::G) : "memory");
    };
    if constexpr (!PIPE) {
        for (int st = 0; st < n_stages; ++st) {
            // own parts of stage st have landed once at most the NS - 2 younger stages' DMAs are outstanding ...
            wait_stages_in_flight();
            // ... and after the barrier everybody's have; every wave has also finished reading stage st - 1 (its MFMAs
            // consumed those registers), so that slot may be refilled
            __builtin_amdgcn_s_barrier();
            {
                const int nx = st + NS - 1;
                issue_stage(nx < n_stages ? nx : n_stages - 1);   // past the end: a harmless re-load keeps the counts fixed
            }
            const char* slot = smem + (st % NS) * Geo::STAGE;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + ks * 8192 + a_off0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + ks * 8192 + a_off1);
                const char* wbase = slot + Geo::A_STAGE + ks * PL * 1024 + lane * 16;
                if constexpr (MODE == 0) {
                    bf16x8 af[3];
                    wl_split3(x0, x1, af);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        bf16x8 wf[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[pl] = *reinterpret_cast<const bf16x8*>(wbase + (b * KS * PL + pl) * 1024);
#pragma unroll
                        for (int t = 0; t < 6; ++t) acc[b] = RAGB_WL_MFMA_BF16(wf[kTerm[t][0]], af[kTerm[t][1]], acc[b]);
                    }
                } else if constexpr (MODE == 2) {
                    f16x8 ah, al;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = e < 4 ? x0[e] : x1[e - 4];
                        const _Float16 hi = (_Float16)x;
                        ah[e] = hi;
                        al[e] = (_Float16)((x - (float)hi) * kX3Scale);
                        amax = fmaxf(amax, fabsf(x));   // NaN never wins: a NaN input gives a NaN output, as in fp32
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(wbase + (b * KS * PL + 0) * 1024);
                        const f16x8 wl = *reinterpret_cast<const f16x8*>(wbase + (b * KS * PL + 1) * 1024);
                        acc[b] = RAGB_WL_MFMA_F16(wh, ah, acc[b]);
                        accx[b] = RAGB_WL_MFMA_F16(wl, ah, accx[b]);
                        accx[b] = RAGB_WL_MFMA_F16(wh, al, accx[b]);
                    }
                } else {
                    f16x8 af;
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[e] = (_Float16)(e < 4 ? x0[e] : x1[e - 4]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f16x8 wf = *reinterpret_cast<const f16x8*>(wbase + (b * KS * PL) * 1024);
                        acc[b] = RAGB_WL_MFMA_F16(wf, af, acc[b]);
                    }
                }
            }
        }
        if constexpr (MODE == 2) {   // the cross terms carry the low planes' 2^11
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = __builtin_fmaf(accx[b][i], kX3Unscale, acc[b][i]);
            // a value fp16 cannot hold became inf in the high plane: tell the host, which repeats the pass on the
            // split-bf16 path (fp32's exponent range)
            if (p.range_flag && amax >= kF16Max) *p.range_flag = 1u;
        }
    } else {
        // own A rows of stage 0: every DMA issued so far except the ones behind them
        bf16x8 af[3];
        f16x8 ah;
        auto take_a = [&](int st) {   // A fragment of stage st from its slot -> af / ah
            const char* slot = smem + (st % NS) * Geo::STAGE;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + a_off0);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + a_off1);
            if constexpr (MODE == 0) {
                wl_split3(x0, x1, af);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ah[e] = (_Float16)(e < 4 ? x0[e] : x1[e - 4]);
            }
        };
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * Geo::G - 2) : "memory");
        take_a(0);
        for (int st = 0; st < n_stages; ++st) {
            wait_stages_in_flight();
            __builtin_amdgcn_s_barrier();
            {
                const int nx = st + NS - 1;
                issue_stage(nx < n_stages ? nx : n_stages - 1);
            }
            const char* wbase = smem + (st % NS) * Geo::STAGE + Geo::A_STAGE + lane * 16;
            if constexpr (MODE == 0) {
                bf16x8 wf[4][3];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wf[b][pl] = *reinterpret_cast<const bf16x8*>(wbase + (b * PL + pl) * 1024);
                bf16x8 ac[3] = {af[0], af[1], af[2]};
                // stage st + 1's A rows (this wave's own DMAs): behind them are its W fragments and the younger stages
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL + (NS - 2) * Geo::G) : "memory");
                take_a(st + 1);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = RAGB_WL_MFMA_BF16(wf[b][kTerm[t][0]], ac[kTerm[t][1]], acc[b]);
            } else {
                f16x8 wf[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) wf[b] = *reinterpret_cast<const f16x8*>(wbase + b * PL * 1024);
                const f16x8 ac = ah;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL + (NS - 2) * Geo::G) : "memory");
                take_a(st + 1);
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = RAGB_WL_MFMA_F16(wf[b], ac, acc[b]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last stages
    __builtin_amdgcn_s_barrier();                       // every wave is done with the ring: it carries the output now

    if (p.fold_s || p.ts_out) wl_epilogue_lnf(p, acc, smem, reinterpret_cast<const float2*>(smem + Geo::ROWST), m0, n0, wave, lane);
    else wl_epilogue(p, acc, smem, m0, n0, wave, lane);
}

// ---- role-split rings --------------------------------------------------------------------------------------------
// What the first form above showed (scripts/exp/gemm_wl_bench.hip, ablation builds): with its MFMAs removed it still
// took 60 % of its time, and the two did not overlap — every stage waited ~2 us for its DMAs.  A's rows are a first
// touch of HBM (each 128-row slab is read by the N / 128 workgroups of its row tile at about the same moment: one of
// them misses, the others queue behind the same miss), W's fragments are L2 hits; but a wave's vmcnt retires IN ORDER,
// so a wave that issues both waits for the slow one whichever it needs, and a deeper A ring buys nothing.  Here the two
// streams are issued by DIFFERENT waves: waves 0-1 bring A (64 rows each) into a ring of NA stages, waves 2-3 bring W
// (two column tiles each) into a ring of NW stages, each with its own counted wait — A can be NA - 1 stages ahead
// (HBM latency) while W stays NW - 1 ahead (L2 latency) — and one barrier per stage publishes both.  All four waves
// compute as before: 32 rows x 128 columns each, A fragment from its own rows, split in its own registers.
template <int MODE, int NA, int NW>
struct Wl3Geom {
    static constexpr int PL = MODE == 0 ? 3 : 1;
    static constexpr int A_STAGE = 128 * 64;                  // one K-step: 128 rows x 16 fp32
    static constexpr int W_STAGE = 4 * PL * 1024;             // one K-step: 4 column tiles x PL fragments
    static constexpr int A_RING = NA * A_STAGE;
    static constexpr int RING = A_RING + NW * W_STAGE;
    static constexpr int EPI = 4 * 16 * 132 * 4;
    static constexpr int LDS = RING > EPI ? RING : EPI;
    static constexpr int GA = 4;                              // LDS-DMA instructions per A-loader wave per stage
    static constexpr int GW = 2 * PL;                         // ... per W-loader wave per stage
};

template <int MODE, int NA, int NW>
__global__ __launch_bounds__(256, 2) void gemm_nt_wl3_kernel(const GemmWlParams p) {
    using Geo = Wl3Geom<MODE, NA, NW>;
    constexpr int PL = Geo::PL;
    static_assert(NA >= 3 && NW >= 3, "each ring needs a stage in flight beside the one read and the one refilled");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, 128, 128, m0, n0)) return;
    const int n_stages = p.K / 16;
    const bool loader_a = wave < 2;   // wave-uniform

    // A-loader wave w: rows 64 w .. 64 w + 63, four instructions of 16 rows; lane l -> row + l / 4, LDS slot l % 4,
    // source chunk slot ^ ((row >> 2) & 3)
    const char* a_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave & 1) * 64 + q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_src[q] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16;
    }
    // W-loader wave w: column tiles 2 (w - 2) and 2 (w - 2) + 1 of the workgroup's four
    const char* w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int nt = (n0 >> 5) + (wave & 1) * 2 + j;
        nt = nt < (p.N >> 5) ? nt : (p.N >> 5) - 1;
        w_src[j] = static_cast<const char*>(p.Wimg) + (size_t)nt * n_stages * PL * 1024 + lane * 16;
    }
    auto issue_a = [&](int st, int slot) {
        char* dst = smem + slot * Geo::A_STAGE + (wave & 1) * 64 * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(a_src[q] + (size_t)st * 64, dst + q * 1024);
    };
    auto issue_w = [&](int st, int slot) {
        char* dst = smem + Geo::A_RING + slot * Geo::W_STAGE + (wave & 1) * 2 * PL * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) glds16(w_src[j] + ((size_t)st * PL + pl) * 1024, dst + (j * PL + pl) * 1024);
    };

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    const int my_row = wave * 32 + r;
    const int sw = (my_row >> 2) & 3;
    const int a_off0 = my_row * 64 + (((2 * h) ^ sw) << 4), a_off1 = my_row * 64 + (((2 * h + 1) ^ sw) << 4);

    if (loader_a) {
#pragma unroll
        for (int st = 0; st < NA - 1; ++st) issue_a(st < n_stages ? st : n_stages - 1, st);
    } else {
#pragma unroll
        for (int st = 0; st < NW - 1; ++st) issue_w(st < n_stages ? st : n_stages - 1, st);
    }
    constexpr int kTerm[6][2] = {{0, 2}, {2, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    int sa = 0, sw_ = 0;   // slots of stage st in the two rings
    for (int st = 0; st < n_stages; ++st) {
        // this wave's own DMAs of stage st have landed when only the younger stages' are outstanding ...
        if (loader_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::GA * (NA - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::GW * (NW - 2)) : "memory");
        // ... after the barrier everybody's have, and everybody is done reading stage st - 1: its slots are free
        __builtin_amdgcn_s_barrier();
        if (loader_a) {
            const int nx = st + NA - 1;
            issue_a(nx < n_stages ? nx : n_stages - 1, sa == 0 ? NA - 1 : sa - 1);
        } else {
            const int nx = st + NW - 1;
            issue_w(nx < n_stages ? nx : n_stages - 1, sw_ == 0 ? NW - 1 : sw_ - 1);
        }
        const char* aslot = smem + sa * Geo::A_STAGE;
        const char* wbase = smem + Geo::A_RING + sw_ * Geo::W_STAGE + lane * 16;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(aslot + a_off0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(aslot + a_off1);
        if constexpr (MODE == 0) {
            bf16x8 af[3];
            wl_split3(x0, x1, af);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                bf16x8 wf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wbase + (b * PL + pl) * 1024);
#pragma unroll
                for (int t = 0; t < 6; ++t) acc[b] = RAGB_WL_MFMA_BF16(wf[kTerm[t][0]], af[kTerm[t][1]], acc[b]);
            }
        } else {
            f16x8 af;
#pragma unroll
            for (int e = 0; e < 8; ++e) af[e] = (_Float16)(e < 4 ? x0[e] : x1[e - 4]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f16x8 wf = *reinterpret_cast<const f16x8*>(wbase + b * PL * 1024);
                acc[b] = RAGB_WL_MFMA_F16(wf, af, acc[b]);
            }
        }
        sa = sa + 1 == NA ? 0 : sa + 1;
        sw_ = sw_ + 1 == NW ? 0 : sw_ + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.fold_s || p.ts_out) wl_epilogue_lnf(p, acc, smem, reinterpret_cast<const float2*>(smem + Geo::ROWST), m0, n0, wave, lane);
    else wl_epilogue(p, acc, smem, m0, n0, wave, lane);
}

// ---- software-pipelined form --------------------------------------------------------------------------------------
// The ablations of the forms above (MFMAs removed: 60 % of the time remains; ring depth and who issues what: no
// effect) say the time is ISSUE time: a wave runs its stage strictly in order — five DMA issues (~100 cycles each),
// fourteen LDS reads, ~50 conversion instructions, THEN twenty-four MFMAs — and nothing of its own overlaps the MFMAs.
// An MFMA holds the SIMD's issue port for 8 of its 32 cycles, so everything else fits in the MFMAs' shadow if it sits
// BETWEEN them in program order.  For that the other work must not depend on this stage's wait:
//   * A runs one stage further ahead than W (ring of 4 against 3): a wave issues A(st + 3) before W(st + 2), so the one
//     counted wait at the top of stage st that retires W(st) has also retired this wave's own A(st + 1) rows;
//   * the A fragment of stage st + 1 is read and split during stage st's MFMAs (its rows are this wave's own DMAs:
//     no barrier needed), the DMAs of the coming stages are issued between MFMAs too;
//   * sched_group_barrier pins the interleave (left alone hipcc clusters the MFMAs at the end).
template <int MODE>
struct Wl4Geom {
    static constexpr int PL = MODE == 0 ? 3 : 1;
    static constexpr int NA = 4, NW = 3;
    static constexpr int A_STAGE = 128 * 64;
    static constexpr int W_STAGE = 4 * PL * 1024;
    static constexpr int A_RING = NA * A_STAGE;
    static constexpr int RING = A_RING + NW * W_STAGE;
    static constexpr int EPI = 4 * 16 * 132 * 4;
    static constexpr int LDS = RING > EPI ? RING : EPI;
    static constexpr int G = 2 + PL;                          // LDS-DMA instructions per wave per stage
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_wl4_kernel(const GemmWlParams p) {
    using Geo = Wl4Geom<MODE>;
    constexpr int PL = Geo::PL, NA = Geo::NA, NW = Geo::NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, 128, 128, m0, n0)) return;
    const int n_stages = p.K / 16;

    const char* a_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_src[q] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16;
    }
    int nt_w = (n0 >> 5) + wave;
    nt_w = nt_w < (p.N >> 5) ? nt_w : (p.N >> 5) - 1;
    const char* w_src = static_cast<const char*>(p.Wimg) + (size_t)nt_w * n_stages * PL * 1024 + lane * 16;
    auto issue_a = [&](int st) {   // this wave's 32 rows of stage st (clamped) into A slot st % NA
        const int sc = st < n_stages ? st : n_stages - 1;
        char* dst = smem + (st % NA) * Geo::A_STAGE + wave * 32 * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(a_src[q] + (size_t)sc * 64, dst + q * 1024);
    };
    auto issue_w = [&](int st) {   // column tile `wave` of stage st (clamped) into W slot st % NW
        const int sc = st < n_stages ? st : n_stages - 1;
        char* dst = smem + Geo::A_RING + (st % NW) * Geo::W_STAGE + wave * PL * 1024;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) glds16(w_src + ((size_t)sc * PL + pl) * 1024, dst + pl * 1024);
    };

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    const int my_row = wave * 32 + r;
    const int sw = (my_row >> 2) & 3;
    const int a_off0 = my_row * 64 + (((2 * h) ^ sw) << 4), a_off1 = my_row * 64 + (((2 * h + 1) ^ sw) << 4);

    // prologue, in the steady-state order A(s + 1) before W(s):  A0 | A1 W0 | A2 W1
    issue_a(0);
    issue_a(1); issue_w(0);
    issue_a(2); issue_w(1);
    bf16x8 af[3];
    f16x8 ah;
    auto take_a = [&](int st) {   // A fragment of stage st (this wave's own rows, landed) -> af / ah
        const char* slot = smem + (st % NA) * Geo::A_STAGE;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + a_off0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + a_off1);
        if constexpr (MODE == 0) {
            wl_split3(x0, x1, af);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ah[e] = (_Float16)(e < 4 ? x0[e] : x1[e - 4]);
        }
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Geo::G) : "memory");   // A0 has landed (behind it: A1 W0 A2 W1)
    take_a(0);
    constexpr int kTerm[6][2] = {{0, 2}, {2, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    for (int st = 0; st < n_stages; ++st) {
        // outstanding here: A(st+1) W(st) | A(st+2) W(st+1).  All but the youngest G: W(st) has landed — and so has
        // this wave's A(st+1), issued just before it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::G) : "memory");
        __builtin_amdgcn_s_barrier();   // everybody's W(st) is in; everybody is done with W(st-1): its slot is free
        const char* wbase = smem + Geo::A_RING + (st % NW) * Geo::W_STAGE + lane * 16;
        if constexpr (MODE == 0) {
            bf16x8 wf[4][3];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[b][pl] = *reinterpret_cast<const bf16x8*>(wbase + (b * PL + pl) * 1024);
            const bf16x8 ac[3] = {af[0], af[1], af[2]};
            // A(st+1), this wave's own rows, for the split below
            const char* nslot = smem + ((st + 1) % NA) * Geo::A_STAGE;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(nslot + a_off0);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(nslot + a_off1);
            // the coming stages' DMAs: destination and source of each of this wave's five instructions
            const int sa3 = st + 3 < n_stages ? st + 3 : n_stages - 1, sw2 = st + 2 < n_stages ? st + 2 : n_stages - 1;
            char* a_dst = smem + ((st + 3) % NA) * Geo::A_STAGE + wave * 32 * 64;
            char* w_dst = smem + Geo::A_RING + ((st + 2) % NW) * Geo::W_STAGE + wave * PL * 1024;
            __builtin_amdgcn_sched_barrier(0);
            // 24 MFMAs, one at a time, each followed by a slice of the other work — pinned: an MFMA occupies the
            // SIMD's issue port for 8 of its 32 cycles, what sits right behind it runs in its shadow
            float r1[8];
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int t = i >> 2, b = i & 3;
                acc[b] = RAGB_WL_MFMA_BF16(wf[b][kTerm[t][0]], ac[kTerm[t][1]], acc[b]);
                if (i == 1) glds16(a_src[0] + (size_t)sa3 * 64, a_dst);
                if (i == 3) glds16(a_src[1] + (size_t)sa3 * 64, a_dst + 1024);
                if (i == 5) glds16(w_src + ((size_t)sw2 * PL + 0) * 1024, w_dst);
                if (i == 7) glds16(w_src + ((size_t)sw2 * PL + 1) * 1024, w_dst + 1024);
                if (i == 9) glds16(w_src + ((size_t)sw2 * PL + 2) * 1024, w_dst + 2048);
                if (i >= 8) {   // elements 2 (i - 8) / 4 ... : two elements per pair of MFMAs, in two half-steps
                    const int e0 = ((i - 8) >> 1) * 1;   // element handled by this pair of slots: 0..7
                    const float x = e0 < 4 ? x0[e0] : x1[e0 - 4];
                    if (((i - 8) & 1) == 0) {
#if defined(RAGB_WL_NO_SPLIT)
                        af[0][e0] = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(x) >> 16));
                        r1[e0] = 0.f;
#else
                        const __bf16 hi = (__bf16)x;
                        af[0][e0] = hi;
                        r1[e0] = x - (float)hi;
#endif
                    } else {
#if defined(RAGB_WL_NO_SPLIT)
                        af[1][e0] = af[0][e0];
                        af[2][e0] = af[0][e0];
#else
                        const __bf16 mid = (__bf16)r1[e0];
                        af[1][e0] = mid;
                        af[2][e0] = (__bf16)(r1[e0] - (float)mid);
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f16x8 wf[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) wf[b] = *reinterpret_cast<const f16x8*>(wbase + b * PL * 1024);
            const f16x8 ac = ah;
            issue_a(st + 3);
            issue_w(st + 2);
            take_a(st + 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = RAGB_WL_MFMA_F16(wf[b], ac, acc[b]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.fold_s || p.ts_out) wl_epilogue_lnf(p, acc, smem, reinterpret_cast<const float2*>(smem + Geo::ROWST), m0, n0, wave, lane);
    else wl_epilogue(p, acc, smem, m0, n0, wave, lane);
}

// ---- small batches (the query encoder: 32 queries ~ 450 tokens) ------------------------------------------------------
// The same two-plane fp16 arithmetic on 64-row tiles, so that 450 tokens still spread over the chip: a workgroup is
// 64 rows x 64 NB columns (NB = 1: up to 252 workgroups for bge-base's QKV projection; NB = 2 when that would exceed the
// CU count), four waves as 2 (rows) x 2 (columns), each 32 rows x 32 NB columns.  The fp32 MFMA these GEMMs ran on
// needs 512 matrix cycles per 16-deep K-step of a 32 x 32 tile, the three fp16 products need 96; what is left is a
// stream of W (read once from HBM: every layer's weights exceed the Infinity Cache over a forward pass) and of A (L2),
// both by LDS-DMA, four K-steps per stage and barrier.  Split-K over blockIdx.z writes fp32 slabs exactly like
// gemm_nt_small_kernel (splitk_bias_res_ln_kernel reduces them), an unsplit launch applies bias / activation /
// residual itself.  K range of a split: a multiple of 64.
// KS_ = 4: the standalone form (96 / 144 KiB of LDS, sixteen waves).  KS_ = 1: the BACKGROUND form — one K-step per stage,
// four waves, a ring of four 8-KiB stages (32 KiB) — for a caller that runs the encoder on a side stream under a
// long-running kernel of another stream: the corpus scan holds one workgroup with 65-115 KiB of LDS on every CU for
// milliseconds, and only kernels that fit beside it (<= 45 KiB) run before it ends (bench.py, pipelined leg).
template <int NB, int KS_ = 4>
struct WsGeom {
    static constexpr int KS = KS_, NS = KS_ == 1 ? 4 : 3;     // (a fourth 32-KiB stage changes nothing: scripts/prof_encoder.py)
    static constexpr int KG = KS_;                            // wave groups: group g multiplies K-step g of every stage
    static constexpr int THREADS = 256 * KG;
    static constexpr int A_STAGE = KS * 64 * 64;              // KS K-steps x 64 rows x 16 fp32
    static constexpr int W_STAGE = KS * 2 * NB * 2 * 1024;    // KS K-steps x 2 NB column tiles x 2 planes
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int RED = (KG - 1) * 4 * NB * 16 * 64 * 4;   // partial tiles of groups 1 .. KG-1, parked for the sum (KG > 1)
    static constexpr int LDS = NS * STAGE > RED ? NS * STAGE : RED;
    static constexpr int G = 1 + NB;                          // LDS-DMA instructions per wave per stage
};

struct GemmWsParams {
    const float* A;
    const _Float16* Wimg;   // two fp16 planes, fragment order (pack_f16x2_frag_kernel)
    const float* bias;
    const float* R;
    float* C;               // [M][ldc]; with split-K: [splits][M][ldc] partial slabs
    int M, N, K;
    int lda, ldr, ldc;
    int act;
    int k_per_split;        // K range of one split, multiple of 64; == K when not split
    int splits;             // the C slabs written: [splits][M][ldc]
    uint32_t* range_flag;
    // ---- LayerNorm folded into its consumers (below); all null / 0 = the plain forms above
    const float2* ts_in;    // [M][nblk_in] (mean, M2) of 32-column blocks of the rows whose LayerNorm is pending
    int nblk_in;            // that LayerNorm's width / 32
    float ln_eps;
    const float* fold_s;    // form A: s[n] = sum_k W'[n][k] (W' = W gamma: the image), bias = b + W beta:
                            //         C = act(rstd (acc - mean s[n]) + bias[n])
    const float* Ry;        // form B: the residual BEFORE its LayerNorm, [M][ldr]: R = (Ry - mean) rstd ln_g[n] + ln_b[n]
    const float* ln_g;
    const float* ln_b;
    float* Yout;            // form B: [M][ldc] the finished pre-LayerNorm rows (C holds the split-K slabs)
    float2* ts_out;         // form B: [M][N / 32] (mean, M2) of the rows this GEMM writes (their LayerNorm is applied by
                            //         whoever reads them next)
    unsigned* tile_ctr;     // form B with split-K: one arrival counter per 64 x 64 NB output tile, zero between launches
};

// ---- LayerNorm folded into its consumers ------------------------------------------------------------------------------
// The query encoder at batch 32 is a chain of ~85 dependent launches of 5-25 us; a quarter of them (24 of bge-base's)
// were "sum the split-K slabs + bias + residual, LayerNorm" passes that do microseconds of work.  They are gone:
//   * a GEMM that produces a layer's pre-LayerNorm sum y (attention output / feed-forward output projection, split-K)
//     finishes itself: the LAST of a tile's K-splits to arrive (one atomic counter per tile; nobody waits for anybody)
//     adds the slabs in split order, the bias and the residual, stores y, and leaves per row the (mean, M2) of each
//     32-column block it holds — Chan's pairwise statistics, so no E[y^2] - E[y]^2 cancellation;
//   * whoever reads y next applies the LayerNorm on the fly.  A GEMM whose INPUT is LN(y) (QKV, feed-forward input)
//     runs on the raw y with gamma folded into its weight image: LN(y) W^T = rstd (y W'^T - mean s) + (b + W beta),
//     W' = W gamma, s[n] = sum_k W'[n][k] — the K loop is untouched, the epilogue applies two per-row scalars;
//   * a GEMM whose RESIDUAL is LN(y) recomputes it per element in its epilogue: (y - mean) rstd gamma[n] + beta[n].
// Per row, mean and rstd come from combining the H / 32 block statistics (lnf_row_stats): ~50 loads per lane, issued
// at the top of the epilogue.
// Sixteen waves per workgroup.  With four (one per SIMD) a stage ran strictly in order — DMA issues, LDS reads, the
// conversion, three MFMAs per K-step, four K-steps — about 1 us per stage whatever the ring depth or the L2 placement
// (kernel traces, scripts/prof_encoder.py): issue latency, not bandwidth.  Here wave (g, quadrant) takes K-step g of
// every stage for its 32 x 32 NB quadrant, so four waves per SIMD overlap each other's latencies and each issues a
// quarter of the DMAs; the four partial tiles of a quadrant are added through LDS at the end, in group order.
template <int NB, int KS_ = 4>
__global__ __launch_bounds__((WsGeom<NB, KS_>::THREADS)) void gemm_nt_ws_kernel(const GemmWsParams p) {
    using Geo = WsGeom<NB, KS_>;
    constexpr int KS = Geo::KS, NS = Geo::NS, KG = Geo::KG;
    static_assert(KS == KG, "one K-step of a stage per wave group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, quad = wave & 3;      // K-step of the stage; quadrant of the tile
    const int wm = quad >> 1, wn = quad & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware work order (1-D grid of ws_grid() workgroups; the hardware deals workgroup ids round-robin to the 8 XCDs,
    // each with its own 4 MiB L2).  A unit of work is (column tile, K split, row tile), numbered with the row tile
    // fastest, and XCD x takes the x-th eighth of that sequence: the row tiles that share a W slice sit on one XCD
    // (two at a boundary) and every XCD gets the same number of workgroups to within one — with one workgroup per CU
    // an XCD that holds 33 runs two rounds.
    const int mt = (p.M + 63) / 64, nct = (p.N + 64 * NB - 1) / (64 * NB);
    const int units = nct * p.splits * mt, per = (units + 7) / 8;
    const int slot = (int)blockIdx.x >> 3, unit = ((int)blockIdx.x & 7) * per + slot;
    if (slot >= per || unit >= units) return;
    const int cz = unit / mt, kz = cz % p.splits;
    const int m0 = (unit % mt) * 64, n0 = (cz / p.splits) * 64 * NB;
    const int nks = p.K / 16;
    const int ks_beg = kz * (p.k_per_split / 16);
    const int ks_end = min(nks, ks_beg + p.k_per_split / 16);
    const int n_stages = (ks_end - ks_beg) / KS;
    const bool split = p.splits > 1;

    // LDS-DMA sources.  Wave (g, q) brings, of K-step g of every stage: A rows 16 q .. 16 q + 15 (lane l -> row + l / 4,
    // slot l % 4, source chunk slot ^ ((row >> 2) & 3)), and W fragments q (and q + 4 when NB = 2) of the K-step's
    // 4 NB (column tile, plane) pairs.
    const char* a_src;
    {
        const int row = quad * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_src = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16 + (size_t)(ks_beg + kg) * 64;
    }
    const char* w_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int f = quad + 4 * j;                 // fragment of the K-step: column tile f >> 1, plane f & 1
        int nt = (n0 >> 5) + (f >> 1);
        nt = nt < (p.N >> 5) ? nt : (p.N >> 5) - 1;
        w_src[j] = reinterpret_cast<const char*>(p.Wimg) + (((size_t)nt * nks + ks_beg + kg) * 2 + (f & 1)) * 1024 + lane * 16;
    }
    auto issue_stage = [&](int st) {
        const int sc = st < n_stages ? st : n_stages - 1;
        char* slot_ = smem + (st % NS) * Geo::STAGE;
        glds16(a_src + (size_t)sc * KS * 64, slot_ + kg * 4096 + quad * 1024);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            glds16(w_src[j] + (size_t)sc * KS * 2 * 1024, slot_ + Geo::A_STAGE + (kg * 4 * NB + quad + 4 * j) * 1024);
    };

    f32x16 acc[NB], accx[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = accx[b][i] = 0.f;
    float amax = 0.f;
    const int my_row = wm * 32 + r;
    const int sw = (my_row >> 2) & 3;
    const int a_off0 = kg * 4096 + my_row * 64 + (((2 * h) ^ sw) << 4), a_off1 = kg * 4096 + my_row * 64 + (((2 * h + 1) ^ sw) << 4);

#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_stage(st);
    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::G * (NS - 2)) : "memory");   // own parts of stage st landed
        __builtin_amdgcn_s_barrier();                                              // everybody's; stage st - 1 is free
        issue_stage(st + NS - 1);
        const char* slot_ = smem + (st % NS) * Geo::STAGE;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot_ + a_off0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot_ + a_off1);
        f16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? x0[e] : x1[e - 4];
            const _Float16 hi = (_Float16)x;
            ah[e] = hi;
            al[e] = (_Float16)((x - (float)hi) * kX3Scale);
            amax = fmaxf(amax, fabsf(x));
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // fragment (column tile wn NB + b, plane) of K-step kg: column tile c = f >> 1 with f = 2 c + plane
            const char* wb = slot_ + Geo::A_STAGE + (kg * 4 * NB + 2 * (wn * NB + b)) * 1024 + lane * 16;
            const f16x8 wh = *reinterpret_cast<const f16x8*>(wb);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(wb + 1024);
            acc[b] = RAGB_WL_MFMA_F16(wh, ah, acc[b]);
            accx[b] = RAGB_WL_MFMA_F16(wl, ah, accx[b]);
            accx[b] = RAGB_WL_MFMA_F16(wh, al, accx[b]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last stages
    if (p.range_flag && amax >= kF16Max) *p.range_flag = 1u;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = __builtin_fmaf(accx[b][i], kX3Unscale, acc[b][i]);
    if constexpr (KG > 1) {
    __builtin_amdgcn_s_barrier();                       // every wave is done with the ring: it carries the partials now
    float* red = reinterpret_cast<float*>(smem);        // [KG - 1][4 quadrants][NB][16][64]
    if (kg > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[((((kg - 1) * 4 + quad) * NB + b) * 16 + i) * 64 + lane] = acc[b][i];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] += red[((((g - 1) * 4 + quad) * NB + b) * 16 + i) * 64 + lane];
    }
    float* Cz = p.C + (split ? (size_t)kz * p.M * p.ldc : 0);
    const int m = m0 + wm * 32 + r;
    if (!p.fold_s && !p.ts_out) {   // the plain forms
#pragma unroll
        for (int b = 0; b < NB; ++b)
            store_tile_rows(acc[b], m, n0 + (wn * NB + b) * 32, h, p.M, p.N, p.bias, p.R, p.ldr, Cz, p.ldc, p.act, split);
        return;
    }
    // ---- LayerNorm folded into its consumers: this row's pending statistics
    float mean = 0.f, rstd = 1.f;
    if (p.ts_in && m < p.M) lnf_row_stats(p.ts_in + (size_t)m * p.nblk_in, p.nblk_in, p.ln_eps, mean, rstd);
    if (p.fold_s) {   // form A: the input was LN(y), gamma is in the image: two per-row scalars finish it
        if (m >= p.M) return;
        const float ms = mean * rstd;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * NB + b) * 32 + 8 * g + 4 * h;
                if (n >= p.N) continue;   // (N % 32 == 0: a float4 is inside or outside)
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.fold_s + n);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = apply_act(__builtin_fmaf(rstd, acc[b][4 * g + e], __builtin_fmaf(-ms, s4[e], b4[e])), p.act);
                *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
            }
        return;
    }
    // form B: y = sum of the splits + bias + LN(Ry); the last split of the tile to arrive finishes it
    if (split) {
        // The slabs cross workgroups INSIDE a launch, and a device-scope fence is not the way: on this chip each XCD has
        // its own L2, so `__threadfence()` compiles to a write-back of the whole L2 (buffer_wbl2) and an invalidate — with
        // 252 workgroups doing that, a 7 us GEMM took 45 (measured).  Instead the slab words themselves are written and
        // read at agent scope (relaxed atomic stores / loads: plain global_store / global_load with the sc1 bit, served
        // by the memory side where all XCDs agree), and a workgroup-scope release (a wait for this wave's stores) in
        // front of the barrier orders them before the arrival count.
        if (m < p.M) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + (wn * NB + b) * 32 + 8 * g + 4 * h;
                    if (n >= p.N) continue;
                    float* dst = Cz + (size_t)m * p.ldc + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store(dst + e, acc[b][4 * g + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's slab stores have completed ...
        __syncthreads();                                         // ... every (remaining) wave's have, and the ring is free
        unsigned* word = reinterpret_cast<unsigned*>(smem + Geo::LDS - 16);
        const int tile = (cz / p.splits) * mt + (unit % mt);
        if (tid == 0) *word = __hip_atomic_fetch_add(p.tile_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*word != (unsigned)(p.splits - 1)) return;   // workgroup-uniform
        if (tid == 0) __hip_atomic_store(p.tile_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        if (m < p.M) {
            // the slabs in split order, whoever came last; split z + 1's loads are in flight while split z is added
            // (all of a lane's loads at once would take 8 x splits float4 registers)
            f32x4 cur[NB][4], nxt[NB][4];
            auto load_split = [&](f32x4 (&dst)[NB][4], int z) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + (wn * NB + b) * 32 + 8 * g + 4 * h;
                        const float* src = p.C + ((size_t)z * p.M + m) * p.ldc + (n < p.N ? n : 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[b][g][e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            };
            load_split(cur, 0);
            for (int z = 0; z < p.splits; ++z) {
                if (z + 1 < p.splits) load_split(nxt, z + 1);
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[b][4 * g + e] = z == 0 ? cur[b][g][e] : acc[b][4 * g + e] + cur[b][g][e];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) cur[b][g] = nxt[b][g];
            }
        }
    }
    if (m >= p.M) return;
    float* Y = p.Yout;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int nb0 = n0 + (wn * NB + b) * 32;
        if (nb0 >= p.N) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nb0 + 8 * g + 4 * h;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(p.Ry + (size_t)m * p.ldr + n);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.ln_g + n);
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(p.ln_b + n);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (acc[b][4 * g + e] + b4[e]) + __builtin_fmaf((y4[e] - mean) * rstd, g4[e], e4[e]);
                acc[b][4 * g + e] = v[e];
            }
            *reinterpret_cast<f32x4*>(Y + (size_t)m * p.ldc + n) = v;
        }
        const float2 st = lnf_block_stats(acc[b]);
        if (h == 0) p.ts_out[(size_t)m * (p.N >> 5) + (nb0 >> 5)] = st;
    }
}

}  // namespace ragb
