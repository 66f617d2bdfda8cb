// gemm_wt.hip.h — big-batch GEMM over activations stored in MFMA-fragment order ("tiled"), both operands through LDS-DMA.
//
//   C[M][N] = A[M][K] · W[N][K]ᵀ (+ bias[N]) (activation) (+ R[M][N])
//
// Replaces gemm_nt_wl_kernel<1, ...> (gemm_wl.hip.h: row-major fp32 activations, fp16 inputs) on the RAG_GEMM_F16 path —
// the precision the reference runs its reranker at on a GPU (src/pipeline/components/reranker.py:91-93, model.half()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

// Why a layout and not a bigger tile.  gemm_nt_wl_kernel's ablations put its bound at operand delivery,
// so the first attempt was fewer operand bytes per product: 256-row tiles with row-major fp16 activations
// (scripts/exp/gemm_wh_rowmajor.hip.h).  On hardware bigger tiles were SLOWER — 256 x 256, a third of the bytes per
// product, ran at 0.6 of the 128 x 128 kernel's speed — and the one shape that won (256 x 128) won only while two
// workgroups fit a CU.  The bound is not the byte count; it is how a row-major activation operand is fetched: 64 bytes
// from each of 16 rows per DMA instruction, half a 128-byte line per row and stage, the other half fetched again a
// stage later after the CU's vector cache has turned over.  With the SAME tile and the activations stored as below, the
// same GEMMs run 1.9-2.1x faster (scripts/exp/gemm_wh_bench.hip: N x K = 1152 x 384 at M = 178 405: 0.53 -> 0.25 ms).
//
// Here the ACTIVATIONS live in memory the way the matrix cores read them: a [M][F] matrix is stored as row blocks of
// 32 tokens x 16-feature steps, each (row block, step) one MFMA operand fragment in lane order — lane 32 h + r holds
// token 32 rb + r, features 16 ks + 8 h + 0..7.
//
//   T16 (fp16, RAG_GEMM_F16):  fragment = 1 KiB, lane l's eight halves at 16 l
//        element (m, f) at  (((m >> 5) F/16 + (f >> 4)) 64 + 32 ((f >> 3) & 1) + (m & 31)) 8 + (f & 7)          [halves]
//   T32 (fp32, RAG_GEMM_F32):  fragment = 2 KiB = two lane-linear halves: floats 0..3 of every lane, then floats 4..7
//        element (m, f) at  ((((m >> 5) F/16 + (f >> 4)) 2 + ((f >> 2) & 1)) 64 + 32 ((f >> 3) & 1) + (m & 31)) 4 + (f & 3)   [floats]
//
// A wave's A operand for a K-step is then 1 or 2 contiguous KiB — whole 128-byte lines, one DMA instruction each,
// read back lane-linear (no swizzle, no bank conflicts) — and the epilogue needs no LDS transpose: the accumulator of
// lane (r, h) holds token r, features 32 b + 8 g + 4 h + 0..3, which in the consumer's fragment (step 2 b + g / 2,
// lane half g & 1) is one contiguous piece per lane and 512 contiguous bytes per half wave: one store instruction per
// (b, g), every 128-byte line written whole.  Every producer and consumer of activations on the big-batch path
// (embedding, attention, LayerNorm, pooling: bert_tiled.hip.h) uses the layout; M is padded to a multiple of 32 rows
// in the buffers, and rows past M hold unspecified values that no valid row ever depends on.
//
// AK = 1 (T16): W one fp16 plane; fp32 accumulation; bias, activation and the residual join the fp32 sum, which is
//   rounded to fp16 once (a .half() model — reference reranker.py:91-93 — rounds `dense(x)` and the residual sum
//   separately: this is the same arithmetic with one rounding fewer).
// AK = 0 (T32): every fp32 operand as two fp16 planes, split in the wave's own registers — the arithmetic of
//   gemm_nt_wl_kernel<2> term for term, so the results are bit-identical to it; range flag as there.
// GELU for the fp16 path: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, one rcp and one exp; erff costs ~3x as
// many instructions and was a third of the FFN input projection's time).  The result is rounded to fp16 (2^-11
// relative) right after, so the approximation is invisible there; the fp32 paths keep erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float pz = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    pz = __builtin_fmaf(pz, t, 1.421413741f);
    pz = __builtin_fmaf(pz, t, -0.284496736f);
    pz = __builtin_fmaf(pz, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erfz = __builtin_fmaf(-pz * t, e, 1.0f);          // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + __builtin_copysignf(erfz, x));
}

template <typename E>
struct Tiled;
template <>
struct Tiled<_Float16> {
    static constexpr int VEC = 8;   // contiguous elements of one token (16 bytes)
    __host__ __device__ static size_t idx(long long m, int f, int F) {
        return ((((size_t)(m >> 5) * (size_t)(F >> 4) + (size_t)(f >> 4)) * 64) + 32 * ((f >> 3) & 1) + (size_t)(m & 31)) * 8 + (f & 7);
    }
};
template <>
struct Tiled<float> {
    static constexpr int VEC = 4;
    __host__ __device__ static size_t idx(long long m, int f, int F) {
        return (((((size_t)(m >> 5) * (size_t)(F >> 4) + (size_t)(f >> 4)) * 2 + ((f >> 2) & 1)) * 64) + 32 * ((f >> 3) & 1) +
                (size_t)(m & 31)) * 4 + (f & 3);
    }
};

struct GemmWtParams {
    const void* A;          // tiled [M][lda]
    const void* Wimg;       // fragment order: AK = 1 one fp16 plane (pack_f16_frag_kernel), AK = 0 two (pack_f16x2_frag_kernel)
    const float* bias;      // [N] or null
    const void* R;          // tiled [M][ldr] or null
    void* C;                // tiled [M][ldc]
    int M, N, K;            // N % 32 == 0, K % (16 KS) == 0
    int lda, ldr, ldc;      // feature counts of the three layouts (multiples of 16)
    int act;
    uint32_t* range_flag;   // AK = 0: set to 1 when an element of A is outside fp16's range; may be null
};

template <int AK, int NW, int NB, int KS, int NS>
struct WtGeom {
    static constexpr int THREADS = 64 * NW;
    static constexpr int TM = 32 * NW, TN = 32 * NB;
    static constexpr int AF = AK ? 1024 : 2048;               // bytes of an A fragment
    static constexpr int PL = AK ? 1 : 2;                     // W planes
    static constexpr int A_STAGE = NW * KS * AF;
    static constexpr int W_STAGE = NB * KS * PL * 1024;
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int LDS = NS * STAGE;
    static constexpr int W_WAVE = W_STAGE / NW;
    static constexpr int WI = (W_WAVE + 1023) / 1024;
    static constexpr int G = KS * (AF / 1024) + WI;
    static_assert(W_WAVE % 512 == 0 && W_WAVE >= 512, "a wave's W range is whole or half DMA instructions");
    static_assert(NS >= 3, "the ring needs a stage in flight beside the one being read and the one being refilled");
};

// ABL (experiment builds, scripts/exp/gemm_wh_bench.hip): 1 = every tile reads row blocks 0..NW-1 of A (always cache
// resident), 2 = no output stores, 4 = one W fragment read from LDS per K-step instead of NB (wrong values, same MFMAs and
// DMA traffic).  The product instantiates ABL = 0 only.
template <int AK, int NW, int NB, int KS, int NS, int OCC, int ABL = 0>
__global__ __launch_bounds__(64 * NW, OCC) void gemm_nt_wt_kernel(const GemmWtParams p) {
    using Geo = WtGeom<AK, NW, NB, KS, NS>;
    constexpr int PL = Geo::PL, AF = Geo::AF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, Geo::TM, Geo::TN, m0, n0)) return;
    const int nks = p.K / 16;
    const int n_stages = nks / KS;
    const int n_tiles32 = p.N >> 5;
    const int nrb = (p.M + 31) >> 5;
    const int rb = (m0 >> 5) + wave;                                 // this wave's row block
    const int rb_ld = (ABL & 1) ? wave : (rb < nrb ? rb : nrb - 1);  // past M: valid memory, results dropped

    const char* a_src = static_cast<const char*>(p.A) + (size_t)rb_ld * (p.lda >> 4) * AF + lane * 16;
    const char* w_src[Geo::WI];
#pragma unroll
    for (int i = 0; i < Geo::WI; ++i) {
        const int o = wave * Geo::W_WAVE + i * 1024 + lane * 16;
        const int f = o >> 10, b = f / (KS * PL), rem = f % (KS * PL);   // LDS fragment f = (b KS + ks) PL + pl
        int nt = (n0 >> 5) + b;
        nt = nt < n_tiles32 ? nt : n_tiles32 - 1;
        w_src[i] = static_cast<const char*>(p.Wimg) + ((size_t)nt * nks * PL + rem) * 1024 + (o & 1023);
    }
    constexpr bool kHalfLast = (Geo::W_WAVE % 1024) != 0;

    auto issue_stage = [&](int st) {
        char* slot = smem + (st % NS) * Geo::STAGE;
        const size_t aoff = (size_t)st * (KS * AF);
#pragma unroll
        for (int q = 0; q < KS * (AF / 1024); ++q) glds16(a_src + aoff + q * 1024, slot + wave * KS * AF + q * 1024);
        const size_t woff = (size_t)st * (KS * PL * 1024);
#pragma unroll
        for (int i = 0; i < Geo::WI; ++i) {
            char* dst = slot + Geo::A_STAGE + wave * Geo::W_WAVE + i * 1024;
            if (kHalfLast && i == Geo::WI - 1) {
                if (lane < 32) glds16(w_src[i] + woff, dst);
            } else {
                glds16(w_src[i] + woff, dst);
            }
        }
    };

    f32x16 acc[NB];
    f32x16 accx[AK ? 1 : NB];   // AK = 0: the cross terms (scaled by 2^11)
    float amax = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int b = 0; b < (AK ? 1 : NB); ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) accx[b][i] = 0.f;

#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_stage(st < n_stages ? st : n_stages - 1);

    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * Geo::G) : "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nx = st + NS - 1;
            issue_stage(nx < n_stages ? nx : n_stages - 1);
        }
        const char* slot = smem + (st % NS) * Geo::STAGE;
        const char* abase = slot + wave * KS * AF + lane * 16;
        const char* wbase = slot + Geo::A_STAGE + lane * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (AK == 1) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(abase + ks * 1024);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const f16x8 wf = *reinterpret_cast<const f16x8*>(wbase + (((ABL & 4) ? 0 : b) * KS + ks) * 1024);   // ABL 4: one W read per step
                    acc[b] = RAGB_WL_MFMA_F16(wf, af, acc[b]);
                }
            } else {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(abase + ks * 2048);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(abase + ks * 2048 + 1024);
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
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wbase + ((b * KS + ks) * PL + 0) * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wbase + ((b * KS + ks) * PL + 1) * 1024);
                    acc[b] = RAGB_WL_MFMA_F16(wh, ah, acc[b]);
                    accx[b] = RAGB_WL_MFMA_F16(wl, ah, accx[b]);
                    accx[b] = RAGB_WL_MFMA_F16(wh, al, accx[b]);
                }
            }
        }
    }
    if constexpr (AK == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = __builtin_fmaf(accx[b][i], kX3Unscale, acc[b][i]);
        if (p.range_flag && amax >= kF16Max) *p.range_flag = 1u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last stages (LDS is not reused)

    if (rb >= nrb) return;
    const size_t c_rb = (size_t)rb * (p.ldc >> 4), r_rb = (size_t)rb * (p.ldr >> 4);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (n0 + 32 * b >= p.N) break;                 // (wave-uniform: N % 32 == 0)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + 32 * b + 8 * g + 4 * h;
            f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (AK == 1 && p.act == ACT_GELU_ERF) ? gelu_erf_fast(v[e]) : apply_act(v[e], p.act);
            const size_t step = (size_t)((n0 >> 4) + 2 * b + (g >> 1));
            if constexpr (AK == 1) {
                const size_t in_frag = (size_t)(32 * (g & 1) + r) * 8 + 4 * h;
                if (p.R) {   // the residual joins the fp32 sum: one rounding to fp16, not two
                    const f16x4 rv = *reinterpret_cast<const f16x4*>(static_cast<const _Float16*>(p.R) + (r_rb + step) * 512 + in_frag);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                }
                const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                if ((ABL & 2) && hv[0] != (_Float16)123.25f) continue;   // (keeps the values live)
                *reinterpret_cast<f16x4*>(static_cast<_Float16*>(p.C) + (c_rb + step) * 512 + in_frag) = hv;
            } else {
                const size_t in_frag = (size_t)h * 256 + (size_t)(32 * (g & 1) + r) * 4;   // floats 4h..4h+3 of the lane's 8
                if (p.R) v += *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.R) + (r_rb + step) * 512 + in_frag);
                if ((ABL & 2) && v[0] != 123.25f) continue;
                *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (c_rb + step) * 512 + in_frag) = v;
            }
        }
    }
}

}  // namespace ragb
