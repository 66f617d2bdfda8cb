// gemm_w6.hip.h — the default mode's big-batch GEMM on 128 x 192 tiles (round 4).
//
//   C[M][N] = A[M][K] · W[N][K]ᵀ (+ bias) (activation) (+ R)      A, C, R fp32 row-major; fp32 accuracy (two fp16 planes)
//
// Same arithmetic and same per-element summation order as gemm_nt_wl_kernel<2, 1, 3> (gemm_wl.hip.h) — reference
// src/pipeline/components/reranker.py:248-252, the cross-encoder's GEMMs — on a tile half again as wide.  Why: that kernel
// is bound by how many LDS-DMA instructions the texture addresser takes per MFMA, not by the matrix pipe or the byte
// count (profiles/r03_gemm_two_plane_counters.txt: addresser busy 60 % at ~36 cycles per 1-KiB instruction, matrix pipe
// 35 %).  A 128 x 128 tile issues 8 (A) + 8 (W) DMA instructions per 16-deep K-step for 48 MFMAs; a 128 x 192 tile issues
// 8 + 12 for 72: 0.28 instead of 0.33 per MFMA, at the SAME two workgroups per CU — the accumulators (6 + 6 tiles of 16
// registers per wave) still fit 256 registers, and LDS is 61 KiB per workgroup.  (Round 3's bigger tiles lost because
// they cost a workgroup per CU; this one does not.)  192 divides every N of the BERT-family shapes here (384, 768,
// 1152, 1536, 2304, 3072); other N keep the 128-wide kernel.
//
// The epilogue is one function for the plain forms and for "LayerNorm folded into its consumers" (gemm_wl.hip.h): after
// the LDS transposition a row of the tile is visited in a 128-column chunk (32 lanes per row, two rows per
// wave-instruction) and a 64-column chunk (16 lanes per row, four rows per instruction); statistics are per (row,
// 64-column block) — a DPP row of 16 lanes is exactly one block in either chunk, so they need four DPP steps and no
// cross-row shuffle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

struct W6Geom {
    static constexpr int NB = 6, TM = 128, TN = 192, NS = 3;
    static constexpr int A_STAGE = 128 * 64;                  // 128 rows x 16 fp32
    static constexpr int W_STAGE = NB * 2 * 1024;             // 6 column tiles x 2 planes
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int RING = NS * STAGE;
    static constexpr int CLD = TN + 4;                        // floats per row of a wave's epilogue strip
    static constexpr int EPI = 4 * 16 * CLD * 4;
    static constexpr int ROWST = RING > EPI ? RING : EPI;
    static constexpr int LDS = ROWST + 4 * 32 * 8;
    static constexpr int G = 2 + 3;                           // LDS-DMA instructions per wave per stage
};

__device__ __forceinline__ float row16_sum(float v) {   // over the 16 lanes of this lane's DPP row; every lane gets the sum
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}

// One 16-row half of a wave's 32 x 192 tile is in its strip `Cs` (row stride CLD).  Chunk of CW columns starting at c0:
// LPR = CW / 4 lanes per row, 64 / LPR rows per wave-instruction.
template <int CW>
__device__ __forceinline__ void w6_epilogue_chunk(const GemmWlParams& p, const float* Cs, const float2* rowst, int c0, int m_base,
                                                  int n0, int lane) {
    constexpr int LPR = CW / 4, RPI = 64 / LPR, ITER = 16 / RPI;
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int n = n0 + c0 + 4 * c4;
    const bool n_ok = n < p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = bv, gv = bv, ev = bv;
    if (p.bias && n_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.fold_s && n_ok) sv = *reinterpret_cast<const f32x4*>(p.fold_s + n);
    if (p.Ry && n_ok) {
        gv = *reinterpret_cast<const f32x4*>(p.ln_g + n);
        ev = *reinterpret_cast<const f32x4*>(p.ln_b + n);
    }
    f32x4 vals[ITER];
    float rsum[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int lr = rsub + RPI * i;                     // row of the strip
        const int m = m_base + lr;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * W6Geom::CLD + c0 + 4 * c4]);
        const bool ok = m < p.M && n_ok;
        float2 st = make_float2(0.f, 1.f);
        if (p.rs_in) st = rowst[lr];
        if (p.fold_s) {   // form A: the input rows' LayerNorm, gamma folded into the image
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = apply_act(__builtin_fmaf(st.y, v[e], __builtin_fmaf(-st.x * st.y, sv[e], bv[e])), p.act);
        } else {
            v += bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
            if (ok) {
                if (p.Ry) {   // form B: the residual's LayerNorm recomputed per element
                    const f32x4 y4 = *reinterpret_cast<const f32x4*>(p.Ry + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += __builtin_fmaf((y4[e] - st.x) * st.y, gv[e], ev[e]);
                } else if (p.R) {
                    v += *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
                }
            }
        }
        if (ok) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
        vals[i] = v;
        rsum[i] = (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (p.ts_out) {   // (mean, M2) of every 64-column block of the rows just written: one DPP row of 16 lanes each
#pragma unroll
        for (int i = 0; i < ITER; ++i) rsum[i] = row16_sum(rsum[i]) * (1.f / 64.f);
        float m2[ITER];
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dl = vals[i][e] - rsum[i];
                a = __builtin_fmaf(dl, dl, a);
            }
            m2[i] = a;
        }
#pragma unroll
        for (int i = 0; i < ITER; ++i) m2[i] = row16_sum(m2[i]);
        if ((lane & 15) == 0 && n_ok) {
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int m = m_base + rsub + RPI * i;
                if (m < p.M) p.ts_out[(size_t)m * (p.N >> 6) + (n >> 6)] = make_float2(rsum[i], m2[i]);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_w6_kernel(const GemmWlParams p) {
    using Geo = W6Geom;
    constexpr int NB = Geo::NB, NS = Geo::NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ALL of the kernel's LDS (see gemm_nt_wl_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!xcd_tile(p.M, p.N, Geo::TM, Geo::TN, m0, n0)) return;
    const int nks = p.K / 16;

    // LDS-DMA sources.  A: two instructions of 16 rows x 64 bytes per K-step for THIS wave's 32 rows (lane l -> row 16 q + l / 4,
    // LDS slot l % 4, source chunk slot ^ ((row >> 2) & 3)).  W: the K-step's twelve fragments (column tile b, plane pl) = f =
    // 2 b + pl; wave w brings fragments 3 w .. 3 w + 2.
    const char* a_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;                               // rows past M: valid memory, results dropped
        a_src[q] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16;
    }
    const char* w_src[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = wave * 3 + j;
        int nt = (n0 >> 5) + (f >> 1);
        nt = nt < (p.N >> 5) ? nt : (p.N >> 5) - 1;
        w_src[j] = static_cast<const char*>(p.Wimg) + ((size_t)nt * nks * 2 + (f & 1)) * 1024 + lane * 16;
    }
    auto issue_stage = [&](int st) {   // K-step st (clamped by the caller) into slot st % NS
        char* slot = smem + (st % NS) * Geo::STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(a_src[q] + (size_t)st * 64, slot + (wave * 32 + q * 16) * 64);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16(w_src[j] + (size_t)st * 2048, slot + Geo::A_STAGE + (wave * 3 + j) * 1024);
    };

    f32x16 acc[NB], accx[NB];   // accx: the cross terms (scaled by 2^11)
    float amax = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = accx[b][i] = 0.f;

    const int my_row = wave * 32 + r;
    const int sw = (my_row >> 2) & 3;
    const int a_off0 = my_row * 64 + (((2 * h) ^ sw) << 4), a_off1 = my_row * 64 + (((2 * h + 1) ^ sw) << 4);

    if (p.rs_in) {   // (mean, rstd) of this wave's 32 rows: one 4-byte DMA per lane, the oldest of the wave's DMAs
        int mrow = m0 + wave * 32 + (lane >> 1);
        mrow = mrow < p.M ? mrow : p.M - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(p.rs_in + mrow) + (lane & 1)),
                                         (__attribute__((address_space(3))) void*)(smem + Geo::ROWST + wave * 256), 4, 0, 0);
    }
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_stage(st < nks ? st : nks - 1);

    for (int st = 0; st < nks; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Geo::G) : "memory");   // own parts of K-step st have landed ...
        __builtin_amdgcn_s_barrier();                                    // ... everybody's have, and K-step st - 1's slot is free
        {
            const int nx = st + NS - 1;
            issue_stage(nx < nks ? nx : nks - 1);
        }
        const char* slot = smem + (st % NS) * Geo::STAGE;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + a_off0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + a_off1);
        f16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? x0[e] : x1[e - 4];
            const _Float16 hi = (_Float16)x;
            ah[e] = hi;
            al[e] = (_Float16)((x - (float)hi) * kX3Scale);
            amax = fmaxf(amax, fabsf(x));
        }
        const char* wbase = slot + Geo::A_STAGE + lane * 16;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(wbase + (2 * b) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(wbase + (2 * b + 1) * 1024);
            acc[b] = RAGB_WL_MFMA_F16(wh, ah, acc[b]);
            accx[b] = RAGB_WL_MFMA_F16(wl, ah, accx[b]);
            accx[b] = RAGB_WL_MFMA_F16(wh, al, accx[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = __builtin_fmaf(accx[b][i], kX3Unscale, acc[b][i]);
    if (p.range_flag && amax >= kF16Max) *p.range_flag = 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: each wave transposes its own 32 x 192 tile through its own 16-row strip, two halves of 16 rows
    float* Cs = reinterpret_cast<float*>(smem) + wave * 16 * Geo::CLD;
    const float2* rowst = reinterpret_cast<const float2*>(smem + Geo::ROWST) + wave * 32;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((r >> 4) == hh) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(&Cs[(r & 15) * Geo::CLD + b * 32 + 8 * g + 4 * h]) = v;
                }
        }
        const int m_base = m0 + wave * 32 + hh * 16;
        w6_epilogue_chunk<128>(p, Cs, rowst + hh * 16, 0, m_base, n0, lane);
        w6_epilogue_chunk<64>(p, Cs, rowst + hh * 16, 128, m_base, n0, lane);
    }
}

}  // namespace ragb
