// gemm_wh_rowmajor.hip.h — EXPERIMENT (not part of the product build): big-batch GEMM on 256-row tiles, eight waves,
// ROW-MAJOR activations, both operands through LDS-DMA.  Kept because its measurements are what sent the product to
// the fragment-tiled activation layout (csrc/gemm_wt.hip.h): see the header there and DESIGN.md §4.
//
//   C[M][N] = A[M][K] · W[N][K]ᵀ (+ bias[N]) (activation) (+ R[M][N])
//
// Why a second tile shape beside gemm_wl.hip.h (128 x 128, four waves): that kernel's ablations
// put its bound at the bytes a CU pulls from L2 into LDS — (TM + TN) · K · bytes per tile of TM x TN outputs — not at
// the matrix pipe: with its MFMAs removed it still took 60 % of its time, and ring depth, loader roles and software
// pipelining moved nothing.  The lever left is fewer operand bytes per product: a bigger tile, and a narrower operand.
//
//   AK = 1  "fp16 activations" (RAG_GEMM_F16, the precision the reference runs its reranker at on a GPU,
//           src/pipeline/components/reranker.py:91-93: model.half()): A, R and C are fp16 in memory, W one fp16 plane
//           in fragment order (pack_f16_frag_kernel), fp32 accumulation, bias and activation applied to the fp32 sum,
//           then rounded to fp16, residual added as fp16 + fp16 -> fp16 (exactly what `dense(x) + input_tensor` does on
//           a .half() model).  256 x (32 NB) tiles, NB = 8 / 6 / 4: 256 x 256 moves a third of the L2 -> LDS bytes per
//           product of the 128 x 128 fp32-activation kernel.
//   AK = 0  fp32 in memory, every operand as two fp16 planes (gemm_wl.hip.h MODE 2: hi = fp16(x), lo = fp16((x - hi) 2^11),
//           three products, fp32 accuracy, fp16 range with the range flag): 256 x (32 NB) tiles, NB = 4 or 6.
//
// Structure (as gemm_wl.hip.h where not said otherwise): a wave owns 32 token rows x all 32 NB columns; its A rows are
// loaded by itself (LDS-DMA, 64 bytes per row and stage, 16-byte chunks XOR-swizzled by (row >> 2) & 3 on the source
// side), read by itself, and for AK = 0 split in its own registers.  The W stage image is one contiguous run of
// fragments, cut into eight equal byte ranges, one per wave (1.5 KiB ranges are one full and one half-masked DMA
// instruction).  Ring of NS stages, one raw s_barrier per stage behind a counted vmcnt.  A stage is 32 K-columns for
// AK = 1 (two 16-deep MFMA steps, 16 MFMAs per wave and barrier at NB = 8), 16 for AK = 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ragb {

struct GemmWhParams {
    const void* A;       // [M][lda]  AK = 1: fp16, AK = 0: fp32   (lda in elements; rows 16-byte aligned)
    const void* Wimg;    // fragment-order image: AK = 1 one fp16 plane, AK = 0 two fp16 planes
    const float* bias;   // [N] or null
    const void* R;       // [M][ldr] residual (type of C) or null
    void* C;             // [M][ldc]  AK = 1: fp16, AK = 0: fp32
    int M, N, K;         // N % 32 == 0; K % 32 == 0 (AK = 1) / K % 16 == 0 (AK = 0)
    int lda, ldr, ldc;
    int act;
    uint32_t* range_flag;   // AK = 0: set to 1 when an element of A is outside fp16's range; may be null
};

template <int AK, int NB, int NS>
struct WhGeom {
    static constexpr int WAVES = 8, THREADS = 512;
    static constexpr int TM = 256, TN = 32 * NB;
    static constexpr int KS = AK ? 2 : 1;                     // 16-deep MFMA steps per stage
    static constexpr int PL = AK ? 1 : 2;                     // W planes
    static constexpr int KSTAGE = 16 * KS;                    // K columns per stage
    static constexpr int A_STAGE = TM * 64;                   // bytes: 64 per row (32 fp16 or 16 fp32)
    static constexpr int W_STAGE = NB * KS * PL * 1024;       // bytes: fragments [b][ks][pl] of 1 KiB
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int RING = NS * STAGE;
    static constexpr int W_WAVE = W_STAGE / WAVES;            // this wave's byte range of the W stage image
    static constexpr int WI = (W_WAVE + 1023) / 1024;         // DMA instructions for it (the last may be half-masked)
    static constexpr int G = 2 + WI;                          // LDS-DMA instructions per wave per stage
    static constexpr int CSZ = AK ? 2 : 4;                    // bytes of an output element
    static constexpr int LD = TN + (AK ? 8 : 4);              // epilogue strip row, elements (16-byte aligned rows)
    static constexpr int EPI = WAVES * 16 * LD * CSZ;
    static constexpr int LDS = RING > EPI ? RING : EPI;
    static_assert(W_WAVE % 512 == 0 && W_WAVE >= 512, "a wave's W range is whole or half DMA instructions");
    static_assert(NS >= 3, "the ring needs a stage in flight beside the one being read and the one being refilled");
};

template <int AK, int NB, int NS>
__global__ __launch_bounds__(512) void gemm_nt_wh_kernel(const GemmWhParams p) {
    using Geo = WhGeom<AK, NB, NS>;
    constexpr int KS = Geo::KS, PL = Geo::PL, TN = Geo::TN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, Geo::TM, TN, m0, n0)) return;
    const int nks = p.K / 16;
    const int n_stages = nks / KS;
    const int n_tiles32 = p.N >> 5;

    // ---- LDS-DMA sources.  A: two instructions of 16 rows x 64 bytes per stage for this wave's 32 rows.
    const char* a_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;                               // rows past M: valid memory, results dropped
        a_src[q] = static_cast<const char*>(p.A) + (size_t)am * p.lda * (AK ? 2 : 4) + chunk * 16;
    }
    // W: LDS byte o of the stage image is fragment f = o / 1024 = (b KS + ks) PL + pl; in the global image that
    // fragment of stage st sits at (((n0/32 + b) nks + st KS) PL + ks PL + pl) KiB.
    const char* w_src[Geo::WI];
#pragma unroll
    for (int i = 0; i < Geo::WI; ++i) {
        const int o = wave * Geo::W_WAVE + i * 1024 + lane * 16;
        const int f = o >> 10, b = f / (KS * PL), rem = f % (KS * PL);
        int nt = (n0 >> 5) + b;
        nt = nt < n_tiles32 ? nt : n_tiles32 - 1;                   // column tiles past N: valid memory, results dropped
        w_src[i] = static_cast<const char*>(p.Wimg) + ((size_t)nt * nks * PL + rem) * 1024 + (o & 1023);
    }
    constexpr bool kHalfLast = (Geo::W_WAVE % 1024) != 0;           // the last W instruction covers 512 bytes

    auto issue_stage = [&](int st) {   // stage number st (clamped by the caller) into slot st % NS
        char* slot = smem + (st % NS) * Geo::STAGE;
        const size_t koff = (size_t)st * 64;                        // bytes along K in a row of A
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(a_src[q] + koff, slot + (wave * 32 + q * 16) * 64);
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

    const int my_row = wave * 32 + r;
    const int sw = (my_row >> 2) & 3;

#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_stage(st < n_stages ? st : n_stages - 1);

    for (int st = 0; st < n_stages; ++st) {
        // own DMAs of all but the NS - 2 youngest stages have landed; after the barrier everybody's have, and every
        // wave has finished reading stage st - 1, whose slot is refilled next
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * Geo::G) : "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nx = st + NS - 1;
            issue_stage(nx < n_stages ? nx : n_stages - 1);   // past the end: a harmless re-load keeps the counts fixed
        }
        const char* slot = smem + (st % NS) * Geo::STAGE;
        const char* wbase = slot + Geo::A_STAGE + lane * 16;
        if constexpr (AK == 1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(slot + my_row * 64 + (((2 * ks + h) ^ sw) << 4));
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const f16x8 wf = *reinterpret_cast<const f16x8*>(wbase + (b * KS + ks) * 1024);
                    acc[b] = RAGB_WL_MFMA_F16(wf, af, acc[b]);
                }
            }
        } else {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + my_row * 64 + (((2 * h) ^ sw) << 4));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + my_row * 64 + (((2 * h + 1) ^ sw) << 4));
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
                const f16x8 wh = *reinterpret_cast<const f16x8*>(wbase + (b * PL + 0) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(wbase + (b * PL + 1) * 1024);
                acc[b] = RAGB_WL_MFMA_F16(wh, ah, acc[b]);
                accx[b] = RAGB_WL_MFMA_F16(wl, ah, accx[b]);
                accx[b] = RAGB_WL_MFMA_F16(wh, al, accx[b]);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last stages
    __builtin_amdgcn_s_barrier();                       // every wave is done with the ring: it carries the output now

    // ---- epilogue.  Lane (r, h) holds, for its token row, features 32 b + 8 g + 4 h + 0..3 in acc[b][4g..4g+3].
    // Each wave transposes its own 32 x TN tile through its own 16-row LDS strip (wave-private: no barrier), two
    // halves of 16 rows; bias and activation are applied to the fp32 sums on the way in, the residual on the way
    // out, where every store instruction writes whole rows.
    constexpr int LD = Geo::LD;
    using CT = typename std::conditional<AK == 1, _Float16, float>::type;
    CT* Cs = reinterpret_cast<CT*>(smem) + wave * 16 * LD;
    constexpr int EPC = 16 / Geo::CSZ;                 // elements per 16-byte piece
    constexpr int PPR = TN / EPC;                       // pieces per row
    constexpr int ITER = (16 * PPR + 63) / 64;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((r >> 4) == hh) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = 32 * b + 8 * g + 4 * h;
                    const int n = n0 + nl;
                    f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                    if (p.bias && n < p.N) v += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                    if constexpr (AK == 1) {
                        const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *reinterpret_cast<f16x4*>(&Cs[(r & 15) * LD + nl]) = hv;
                    } else {
                        *reinterpret_cast<f32x4*>(&Cs[(r & 15) * LD + nl]) = v;
                    }
                }
        }
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + lane;
            const int lr = idx / PPR, c = idx - lr * PPR;
            const int m = m0 + wave * 32 + hh * 16 + lr;
            const int n = n0 + c * EPC;
            if (lr < 16 && m < p.M && n < p.N) {
                if constexpr (AK == 1) {
                    f16x8 v = *reinterpret_cast<const f16x8*>(&Cs[lr * LD + c * EPC]);
                    if (p.R) v += *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(p.R) + (size_t)m * p.ldr + n);
                    *reinterpret_cast<f16x8*>(static_cast<_Float16*>(p.C) + (size_t)m * p.ldc + n) = v;
                } else {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * LD + c * EPC]);
                    if (p.R) v += *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
                    *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
                }
            }
        }
    }
}

}  // namespace ragb
