// ffn_fused_t16.hip.h — the feed-forward block of a layer as ONE kernel (RAG_GEMM_F16 big-batch path, hidden <= 384):
//
//   Y[M][H] = GELU(X[M][H] · W1[I][H]ᵀ + b1) · W2[H][I]ᵀ + b2 + X        X, Y fp16 in MFMA-fragment order (gemm_wt.hip.h)
//
// Replaces two gemm_nt_wt_kernel launches of the cross-encoder's fp16 mode (the precision the reference runs its
// reranker at on a GPU: src/pipeline/components/reranker.py:91-93, model(**inputs) at :248-252).  Why: at hidden 384
// that mode runs at half of its own HBM floor, and a quarter of the floor is the 4 x wider intermediate — 178 k tokens
// x 1536 halves written by one GEMM and read back by the next, 1.1 GB per layer (profiles/r03_rerank_f16_minilm_pass.json).
// Here it never leaves the registers:
//
//   * a workgroup is 4 waves, ONE per SIMD, each with the whole register file (up to 512 VGPRs): a wave owns 32 token
//     rows end to end — its X fragments (24 k-steps x 4 registers) stay in registers for the whole kernel, and so does
//     its 32 x 384 fp32 output tile (12 accumulators of 16 registers);
//   * the intermediate is walked in chunks of 128 columns.  Phase 1 of a chunk: H1 = X · W1[chunk]ᵀ, 4 accumulators,
//     24 k-steps; bias (pre-loaded: the accumulators start at b1) and GELU in registers, rounded to fp16 where they sit.
//     Phase 2: Y += H1 · W2[:, chunk]ᵀ with H1 taken FROM THE ACCUMULATOR REGISTERS as the MFMA's token operand — registers
//     8 s .. 8 s + 7 of an H1 tile are "k-step s" with element j of lane half h = column 16 s + 8 (j >> 2) + 4 h + (j & 3);
//     W2's fragment image is packed in that same column order (pack_f16_accorder_kernel), so the contraction pairs the
//     right columns and H1 needs no transpose, no LDS round trip and no HBM round trip;
//   * only the weights move: every stage of the LDS-DMA ring is 12 fragments of 1 KiB (phase 1: 3 k-steps x 4 column
//     blocks of W1; phase 2: one k-step x 12 column blocks of W2), three DMA instructions per wave, twelve MFMAs per wave
//     — 32 bytes per clock and CU at full matrix rate, which is what the addresser delivers.
//     One raw s_barrier per stage behind a counted vmcnt, as gemm_nt_wt_kernel.
// No plain global load sits inside the ring's loop (hipcc would drain the ring at its use): X, and b1 (into LDS), are
// loaded before the ring starts, b2 and the residual after it has drained.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

// W2 fp32 [N][K] -> fp16 image whose K order inside every 32-column block matches the accumulator layout of the H1 tile:
// fragment (nt, T), T = 2 B + s over the K axis (B = 32-column block, s = half of it): lane (r, h) element j holds
// W2[32 nt + r][32 B + 16 s + 8 (j >> 2) + 4 h + (j & 3)], at ((nt K/16 + T) 64 + 32 h + r) 8 + j.
__global__ void pack_f16_accorder_kernel(const float* W, int N, int K, int ldw, _Float16* out) {   // N % 32 == 0, K % 32 == 0
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    const int nt = n >> 5, r = n & 31;
    const int B = k >> 5, w = k & 31, s = w >> 4, u = w & 15;      // u = 8 (j >> 2) + 4 h + (j & 3)
    const int h = (u >> 2) & 1, j = ((u >> 3) << 2) | (u & 3);
    out[(((size_t)nt * (K / 16) + 2 * B + s) * 64 + 32 * h + r) * 8 + j] = (_Float16)W[(size_t)n * ldw + k];
}

struct FfnFusedParams {
    const _Float16* X;      // tiled [M][H]: the layer's (LayerNormed) input, also the residual
    const _Float16* W1img;  // [I][H] one fp16 plane, fragment order (pack_f16_frag_kernel)
    const float* b1;        // [I]
    const _Float16* W2img;  // [H][I] fp16, accumulator column order (pack_f16_accorder_kernel)
    const float* b2;        // [H]
    _Float16* Y;            // tiled [M][H]: pre-LayerNorm output
    int M, I;               // I % 128 == 0
};

template <int HB>   // hidden / 32
struct FfnFusedGeom {
    static constexpr int NS = 6;                         // ring stages
    static constexpr int STAGE = 12 * 1024;
    static constexpr int RING = NS * STAGE;
    static constexpr int lds(int I) { return RING + I * 4; }
};

// ABL (experiment builds only; the product instantiates 0): 1 = no DMA in the loop, 2 = no MFMAs, 4 = no LDS reads in the loop.
// Round 4's measurement with them: the DMA stream is 0.69 of the kernel's 0.73 ms per layer, the rest barely shows.
template <int HB, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_t16_kernel(const FfnFusedParams p) {
    static_assert(HB == 12, "stage shapes below are written for hidden = 384 (12 column blocks, 24 k-steps)");
    using Geo = FfnFusedGeom<HB>;
    constexpr int NS = Geo::NS, STAGE = Geo::STAGE, KS = 2 * HB;       // 24 k-steps over the hidden size
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + Geo::RING);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nrb = (p.M + 31) >> 5;
    const int rb = blockIdx.x * 4 + wave;
    const int rb_ld = rb < nrb ? rb : nrb - 1;                          // past M: valid memory, results dropped
    const int n_chunks = p.I >> 7;
    const int n_stages = n_chunks * 16;
    const int kt2 = p.I >> 4;                                           // fragments per column block of W2's image

    // ---- before the ring: b1 into LDS, this wave's X fragments into registers
    for (int i = tid; i < p.I; i += 256) b1s[i] = p.b1[i];
    f16x8 xf[KS];
    {
        const char* xsrc = reinterpret_cast<const char*>(p.X) + (size_t)rb_ld * KS * 1024 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(xsrc + ks * 1024);
    }

    // stage g of the walk: chunk c = g / 16, phase 1 for g % 16 < 8 (sub-stage q: k-steps 3 q .. 3 q + 2 of W1's four
    // column blocks 4 c .. 4 c + 3), phase 2 otherwise (sub-stage t: W2's twelve column blocks at k-fragment 8 c + t).
    // Wave w issues fragments 3 w .. 3 w + 2 of the stage's twelve.
    // Every workgroup streams the SAME 2.4 MB of weights.  Started together and walking the chunks in the same order,
    // all 256 CUs ask the L2 for the same lines at the same moment and the ring runs at a quarter of its rate (measured:
    // the DMA stream, not the MFMAs or the LDS reads, was 0.69 of 0.73 ms per layer).  So each workgroup starts its walk
    // at a different chunk; the sum over chunks does not care (fp32 accumulation order differs per row block only).
    const int c_rot = (int)((blockIdx.x * 5u) % (unsigned)n_chunks);
    const char* w1 = reinterpret_cast<const char*>(p.W1img) + lane * 16;
    const char* w2 = reinterpret_cast<const char*>(p.W2img) + lane * 16;
    auto issue_stage = [&](int g) {
        const int gc = g < n_stages ? g : n_stages - 1;                 // past the end: a harmless re-load keeps the counts fixed
        int c = (gc >> 4) + c_rot;                                      // this workgroup's chunk order (see c_rot)
        c = c >= n_chunks ? c - n_chunks : c;
        const int sub = gc & 15;
        char* slot = smem + (g % NS) * STAGE;
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            const int i = wave * 3 + i3;                                // fragment of the stage, 0..11
            const char* src;
            if (sub < 8) {
                const int kk = i >> 2, b = i & 3;                       // (k-step within the stage, column block of the chunk)
                src = w1 + ((size_t)(4 * c + b) * KS + 3 * sub + kk) * 1024;
            } else {
                src = w2 + ((size_t)i * kt2 + 8 * c + (sub - 8)) * 1024;
            }
            if (!(ABL & 1) || g < NS - 1) glds16(src, slot + i * 1024);
        }
    };

    f32x16 accy[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) accy[b][i] = 0.f;

#pragma unroll
    for (int g = 0; g < NS - 1; ++g) issue_stage(g);
    __syncthreads();   // b1s is complete (the ring's own barriers follow)

    // A stage's twelve fragments are read from LDS into registers ONE STAGE AHEAD, under the previous stage's MFMAs
    // (with one wave per SIMD nothing else would cover the ds_read latency: read-then-multiply, fragment by fragment,
    // left the matrix pipe idle two thirds of the time).  So at the top of stage g the ring must hold stage g + 1 as
    // well: own DMAs of all but the NS - 3 youngest stages have landed, and after the barrier everybody's have; stage
    // g - 1's slot was emptied into registers a stage ago and is refilled with stage g + NS - 1.
    f16x8 wf[2][12];
    auto read_stage = [&](f16x8 (&dst)[12], int g) {
        const char* slot = smem + (g % NS) * STAGE + lane * 16;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (!(ABL & 4) || g == 0) dst[i] = *reinterpret_cast<const f16x8*>(slot + i * 1024);
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 3) : "memory");
    __builtin_amdgcn_s_barrier();
    read_stage(wf[0], 0);

    for (int cw = 0; cw < n_chunks; ++cw) {
        const int c = cw + c_rot >= n_chunks ? cw + c_rot - n_chunks : cw + c_rot;   // the chunk this step of the walk works on
        f32x16 acc1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {   // the accumulators start at the bias: lane (r, h) holds columns 8 g + 4 h + e
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + 128 * c + 32 * b + 8 * gq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[b][4 * gq + e] = bv[e];
            }
        // ---- phase 1: H1 = X · W1[chunk]ᵀ
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int g = cw * 16 + q;
            if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * 3) : "memory");
            __builtin_amdgcn_s_barrier();
            issue_stage(g + NS - 1);
            read_stage(wf[(q + 1) & 1], g + 1);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (!(ABL & 2)) acc1[b] = RAGB_WL_MFMA_F16(wf[q & 1][kk * 4 + b], xf[3 * q + kk], acc1[b]);
        }
        // GELU, rounded to fp16 where the values sit: registers 8 s .. 8 s + 7 of block b are phase 2's k-step (b, s)
        f16x8 hf[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) hf[b][s][j] = (_Float16)gelu_erf_fast(acc1[b][8 * s + j]);
        // ---- phase 2: Y += H1 · W2[:, chunk]ᵀ
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int g = cw * 16 + 8 + t;
            if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * 3) : "memory");
            __builtin_amdgcn_s_barrier();
            issue_stage(g + NS - 1);
            read_stage(wf[(t + 1) & 1], g + 1);
#pragma unroll
            for (int n2 = 0; n2 < HB; ++n2)
                if (!(ABL & 2)) accy[n2] = RAGB_WL_MFMA_F16(wf[t & 1][n2], hf[t >> 1][t & 1], accy[n2]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last stages

    // ---- epilogue: + b2 + X (residual), one rounding to fp16, stored in fragment order (as gemm_nt_wt_kernel)
    if (rb >= nrb) return;
    const size_t rb_off = (size_t)rb * (2 * HB);        // fragments per row block
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * b + 8 * g + 4 * h;
            f32x4 v = {accy[b][4 * g], accy[b][4 * g + 1], accy[b][4 * g + 2], accy[b][4 * g + 3]};
            v += *reinterpret_cast<const f32x4*>(p.b2 + n);
            const size_t step = (size_t)(2 * b + (g >> 1));
            const size_t in_frag = (size_t)(32 * (g & 1) + r) * 8 + 4 * h;
            const f16x4 rv = *reinterpret_cast<const f16x4*>(p.X + (rb_off + step) * 512 + in_frag);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<f16x4*>(p.Y + (rb_off + step) * 512 + in_frag) = hv;
        }
}

}  // namespace ragb
