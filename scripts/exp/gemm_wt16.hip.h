// gemm_wt16.hip.h — EXPERIMENT: the fragment-order fp16 GEMM (csrc/gemm_wt.hip.h, AK = 1) on a 256 x 256 tile with
// SIXTEEN waves (8 row groups x 2 column halves; wave (rg, cg) owns rows 32 rg, columns 128 cg ..).  Why: the counters
// say the texture addresser is 74 % busy at ~36 cycles per 1-KiB LDS-DMA instruction, so the lever is bytes per MFMA:
// 256 x 256 moves 16 KiB per 16-deep K-step for 512 matrix cycles per CU where 256 x 128 moves 12 KiB for 256.  Sixteen
// waves of 93 registers fit a CU once (four per SIMD), so this runs ONE workgroup per CU.
#pragma once
#include <hip/hip_runtime.h>

namespace ragb {

template <int KS, int NS>
struct Wt16Geom {
    static constexpr int THREADS = 1024, TM = 256, TN = 256;
    static constexpr int A_STAGE = 8 * KS * 1024, W_STAGE = 8 * KS * 1024, STAGE = A_STAGE + W_STAGE;
    static constexpr int LDS = NS * STAGE;
    static constexpr int PER_WAVE = KS / 2;              // A fragments (and W fragments) per wave per stage
    static constexpr int G = 2 * PER_WAVE;
    static_assert(KS % 2 == 0, "16 K fragments per operand and stage over 16 waves");
};

template <int KS, int NS>
__global__ __launch_bounds__(1024, 1) void gemm_nt_wt16_kernel(const GemmWtParams p) {
    using Geo = Wt16Geom<KS, NS>;
    constexpr int PW = Geo::PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rg = wave & 7, cg = wave >> 3;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, Geo::TM, Geo::TN, m0, n0)) return;
    const int nks = p.K / 16;
    const int n_stages = nks / KS;
    const int n_tiles32 = p.N >> 5;
    const int nrb = (p.M + 31) >> 5;

    // this wave's DMA duty: fragment f = wave * PW + i of the stage's 8 KS A fragments [rb][ks] and W fragments [b][ks]
    const char* a_src[PW];
    const char* w_src[PW];
    int lds_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int f = wave * PW + i, blk = f / KS, ks = f % KS;
        int rbl = (m0 >> 5) + blk;
        rbl = rbl < nrb ? rbl : nrb - 1;
        int nt = (n0 >> 5) + blk;
        nt = nt < n_tiles32 ? nt : n_tiles32 - 1;
        a_src[i] = static_cast<const char*>(p.A) + ((size_t)rbl * (p.lda >> 4) + ks) * 1024 + lane * 16;
        w_src[i] = static_cast<const char*>(p.Wimg) + ((size_t)nt * nks + ks) * 1024 + lane * 16;
        lds_off[i] = f * 1024;
    }
    auto issue_stage = [&](int st) {
        char* slot = smem + (st % NS) * Geo::STAGE;
        const size_t off = (size_t)st * (KS * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            glds16(a_src[i] + off, slot + lds_off[i]);
            glds16(w_src[i] + off, slot + Geo::A_STAGE + lds_off[i]);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
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
        const char* abase = slot + rg * KS * 1024 + lane * 16;
        const char* wbase = slot + Geo::A_STAGE + cg * 4 * KS * 1024 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 af = *reinterpret_cast<const f16x8*>(abase + ks * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f16x8 wf = *reinterpret_cast<const f16x8*>(wbase + (b * KS + ks) * 1024);
                acc[b] = RAGB_WL_MFMA_F16(wf, af, acc[b]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int rb = (m0 >> 5) + rg;
    if (rb >= nrb) return;
    const int nw0 = n0 + 128 * cg;
    const size_t c_rb = (size_t)rb * (p.ldc >> 4), r_rb = (size_t)rb * (p.ldr >> 4);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (nw0 + 32 * b >= p.N) break;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nw0 + 32 * b + 8 * g + 4 * h;
            f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = p.act == ACT_GELU_ERF ? gelu_erf_fast(v[e]) : apply_act(v[e], p.act);
            const size_t step = (size_t)((nw0 >> 4) + 2 * b + (g >> 1));
            const size_t in_frag = (size_t)(32 * (g & 1) + r) * 8 + 4 * h;
            if (p.R) {
                const f16x4 rv = *reinterpret_cast<const f16x4*>(static_cast<const _Float16*>(p.R) + (r_rb + step) * 512 + in_frag);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<f16x4*>(static_cast<_Float16*>(p.C) + (c_rb + step) * 512 + in_frag) = hv;
        }
    }
}

}  // namespace ragb
