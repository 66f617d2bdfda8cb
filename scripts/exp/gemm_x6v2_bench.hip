// Experiment: a role-specialised, persistent version of the split-bf16 GEMM (gemm_nt_x6_kernel).
//
// The product kernel runs 2 workgroups x 4 waves per CU, every wave doing everything (A loads, split, LDS
// stores, W loads, fragment reads, MFMAs) with one barrier per K-tile: 41 % MFMA utilisation, and two
// co-resident workgroups each take twice as long as one alone.  Here one workgroup of EIGHT waves per CU:
//   waves 0-3  consumers: W fragments from L2, A fragments from LDS, MFMAs, epilogue
//   waves 4-7  producers: A rows from global, three-way split, LDS ring
// meeting only through LDS counters (a ring of S stages: `full[s]` counts producer fills, `freed[s]` consumer
// releases), no workgroup barrier after start-up; the workgroup is persistent (tiles id, id + grid, ...) so
// the producers run ahead across tile boundaries while the consumers are in an epilogue.
// Output must be bit-identical to gemm_nt_x6_kernel (same products, same order per accumulator).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../rag_inference_pipeline_amd/csrc/bert_kernels.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

namespace v2 {
using namespace ragb;

constexpr int S = 3;                       // ring stages
constexpr int STAGE = 3 * 128 * XLD;       // bf16 elements per stage (three planes of 128 rows x 40)
constexpr int EPLD = 68;                   // epilogue rows: 64 floats + 4 (conflict-free 16-byte writes)

__device__ unsigned g_stuck;  // a wait that gave up (the experiment must never hang the GPU)
__device__ __forceinline__ void wait_ge(volatile unsigned* w, unsigned target) {
    for (int i = 0; *w < target; ++i) {
        __builtin_amdgcn_s_sleep(1);
        if (i > (1 << 20)) {  // ~30 ms: something is wrong; leave with wrong results rather than spin
            g_stuck = 1;
            break;
        }
    }
}

__global__ __launch_bounds__(512) void gemm_nt_x6v2_kernel(const GemmX6Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* ring = reinterpret_cast<__bf16*>(smem);                       // [S][3][128 * XLD]
    float* ep = reinterpret_cast<float*>(smem + (size_t)S * STAGE * 2);  // [4 consumer waves][32 * EPLD]
    __shared__ unsigned full[S], freed[S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < S) { full[tid] = 0; freed[tid] = 0; }
    __syncthreads();

    const int tm = (p.M + 127) / 128, tn = (p.N + 127) / 128;
    const int n_ids = (tm + kXcds - 1) / kXcds * kXcds * tn;              // virtual tile ids (xcd_grid)
    const int nk = p.K / XBK, nks = p.K / 16;
    auto tile_of = [&](int id, int& m0, int& n0) -> bool {                 // xcd_tile for a virtual id
        const int xcd = id % kXcds, j = id / kXcds;
        const int mt = (j / tn) * kXcds + xcd;
        m0 = mt * 128;
        n0 = (j % tn) * 128;
        return m0 < p.M;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const int prow = (lane >> 3), pcol = (lane & 7) * 4;               // rows 32 pw + prow + 8 j, j < 4
        unsigned fill = 0;
        for (int id = blockIdx.x; id < n_ids; id += gridDim.x) {
            int m0, n0;
            if (!tile_of(id, m0, n0)) continue;
            const float* ag[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int am = m0 + 32 * pw + prow + 8 * j;
                am = am < p.M ? am : p.M - 1;
                ag[j] = p.A + (size_t)am * p.lda + pcol;
            }
            f32x4 ra[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ag[j]);
            for (int kt = 0; kt < nk; ++kt, ++fill) {
                f32x4 rn[4];
                const int ktn = kt + 1 < nk ? kt + 1 : kt;
#pragma unroll
#ifdef V2_FAKE_PRODUCER
                for (int j = 0; j < 4; ++j) rn[j] = ra[j];
#else
                for (int j = 0; j < 4; ++j) rn[j] = *reinterpret_cast<const f32x4*>(ag[j] + (size_t)ktn * XBK);
#endif
                const unsigned s = fill % S, use = fill / S;
                if (use > 0) wait_ge(&freed[s], 4 * use);                  // the consumers have left this stage
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __bf16* st = ring + (size_t)s * STAGE;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 hi, mid, lo;
#ifdef V2_NO_SPLIT
                    hi = __builtin_bit_cast(bf16x4, ((const uint2*)&ra[j])[0]);   // wrong values, no vector work
                    mid = __builtin_bit_cast(bf16x4, ((const uint2*)&ra[j])[1]);
                    lo = hi;
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = ra[j][e];
                        hi[e] = (__bf16)x;
                        const float r1 = x - (float)hi[e];
                        mid[e] = (__bf16)r1;
                        lo[e] = (__bf16)(r1 - (float)mid[e]);
                    }
#endif
                    const int o = (32 * pw + prow + 8 * j) * XLD + pcol;
                    *reinterpret_cast<bf16x4*>(st + o) = hi;
                    *reinterpret_cast<bf16x4*>(st + 128 * XLD + o) = mid;
                    *reinterpret_cast<bf16x4*>(st + 2 * 128 * XLD + o) = lo;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&full[s], 1u);
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[j] = rn[j];
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    constexpr int kTerm[6][2] = {{0, 2}, {2, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    unsigned cons = 0;
    float* my_ep = ep + (size_t)wave * 32 * EPLD;
    for (int id = blockIdx.x; id < n_ids; id += gridDim.x) {
        int m0, n0;
        if (!tile_of(id, m0, n0)) continue;
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
        // Software pipeline: everything a step consumes was requested a step (A fragments, 768 matrix cycles)
        // or a whole K-tile (W fragments, 1536) before.  Two W register sets alternate per K-tile.
        bf16x8 wrA[2][3][2], wrB[2][3][2], af0[3][2], af1[3][2];
        auto load_w = [&](bf16x8 (&wr)[2][3][2], int kt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int b = 0; b < 2; ++b) wr[ks][pl][b] = wfrag[b][(size_t)((kt * 2 + ks) * 3 + pl) * 64];
        };
        auto read_af = [&](bf16x8 (&af)[3][2], const __bf16* st, int ks) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    af[pl][a] = *reinterpret_cast<const bf16x8*>(st + pl * 128 * XLD + (wm * 64 + a * 32 + r) * XLD + 16 * ks + 8 * h);
        };
        auto mfma24 = [&](const bf16x8 (&wr)[3][2], const bf16x8 (&af)[3][2]) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#ifdef V2_NO_MFMA
                    { asm volatile("" ::"v"(wr[kTerm[t][0]][b]), "v"(af[kTerm[t][1]][a])); }
#else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[kTerm[t][0]][b], af[kTerm[t][1]][a], acc[a][b], 0, 0, 0);
#endif
        };
        auto stage_ptr = [&](unsigned c) -> const __bf16* { return ring + (size_t)(c % S) * STAGE; };
        // prologue of the tile: W of K-tile 0, first A fragments
        load_w(wrA, 0);
        wait_ge(&full[cons % S], 4 * (cons / S + 1));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        read_af(af0, stage_ptr(cons), 0);
        auto ktile = [&](int kt, const bf16x8 (&wr)[2][3][2], bf16x8 (&wr_next)[2][3][2]) {
            const bool more = kt + 1 < nk;
            load_w(wr_next, more ? kt + 1 : kt);                        // a whole K-tile ahead
            read_af(af1, stage_ptr(cons), 1);                           // second step of this K-tile
            mfma24(wr[0], af0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // af1 has landed: this stage is read out
            if (lane == 0) atomicAdd(&freed[cons % S], 1u);
            ++cons;
            if (more) {                                                 // first step of the next K-tile
                wait_ge(&full[cons % S], 4 * (cons / S + 1));
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                read_af(af0, stage_ptr(cons), 0);
            }
            mfma24(wr[1], af1);
        };
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(kt, wrA, wrB);
            if (kt + 1 < nk) ktile(kt + 1, wrB, wrA);
        }
        // epilogue, one wave on its own: a 32-row half of its 64 x 64 tile goes through the wave's LDS patch, rows
        // leave as 256-byte pieces (bias / activation / residual applied on that side)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(&my_ep[r * EPLD + b * 32 + 8 * g + 4 * h]) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c4 = lane & 15;                                      // float4 column of the 64-wide row
            const int n = n0 + wn * 64 + 4 * c4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && n + 3 < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lr = (lane >> 4) + 4 * i;                        // 0..31
                const int m = m0 + wm * 64 + a * 32 + lr;
                if (m >= p.M || n >= p.N) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(&my_ep[lr * EPLD + 4 * c4]);
                if (n + 3 < p.N) {
                    v += bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                    if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
                    *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) {
                            float x = apply_act(v[e] + (p.bias ? p.bias[n + e] : 0.f), p.act);
                            if (p.R) x += p.R[(size_t)m * p.ldr + n + e];
                            p.C[(size_t)m * p.ldc + n + e] = x;
                        }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the patch is read before the other half lands in it
        }
    }
}
}  // namespace v2

static void fill(float* d, size_t n) {
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

static void run(int M, int N, int K, int act, bool res, const char* name) {
    float *A, *W, *C, *C2, *R, *b; __bf16* Wx;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&C2, (size_t)M * N * 4));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&b, (size_t)N * 4)); CK(hipMalloc(&Wx, (size_t)N * K * 6));
    fill(A, (size_t)M * K); fill(W, (size_t)N * K); fill(R, (size_t)M * N); fill(b, N);
    ragb::pack_x6_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(W, N, K, K, Wx);
    ragb::GemmX6Params g{A, Wx, b, res ? R : nullptr, C, M, N, K, K, N, N, act};
    ragb::GemmX6Params g2 = g; g2.C = C2;
    dim3 grid(ragb::xcd_grid(M, N, 128, 128), 1, 1);
    const size_t lds = (size_t)v2::S * v2::STAGE * 2 + (size_t)4 * 32 * v2::EPLD * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v2::gemm_nt_x6v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int n_cus = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); n_cus = pr.multiProcessorCount; }
    const int pgrid = (int)grid.x < n_cus ? (int)grid.x : n_cus;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) ragb::gemm_nt_x6_kernel<<<grid, 256>>>(g);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms1, e0, e1)); ms1 /= 5;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) v2::gemm_nt_x6v2_kernel<<<dim3(pgrid), 512, lds>>>(g2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms2, e0, e1)); ms2 /= 5;
    }
    std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
    CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
    unsigned stuck = 0;
    CK(hipMemcpyFromSymbol(&stuck, HIP_SYMBOL(v2::g_stuck), sizeof stuck));
    if (stuck) printf("  !! a ring wait gave up (g_stuck)\n");
    size_t diff = 0;
    for (size_t i = 0; i < h1.size(); ++i) diff += std::memcmp(&h1[i], &h2[i], 4) != 0;
    printf("%-22s M=%6d N=%5d K=%5d: product %7.3f ms %6.1f TF/s | v2 %7.3f ms %6.1f TF/s  (%+.1f %%)  differing elements %zu\n", name, M, N, K,
           ms1, 2.0 * M * N * K / ms1 / 1e9, ms2, 2.0 * M * N * K / ms2 / 1e9, (ms1 / ms2 - 1) * 100, diff);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(C2); (void)hipFree(R); (void)hipFree(b); (void)hipFree(Wx);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 178405;
    run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM)");
    run(M, 384, 384, ragb::ACT_NONE, true, "attn out + residual");
    run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
    run(M, 384, 1536, ragb::ACT_NONE, true, "ffn2 + residual");
    run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
    run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
    return 0;
}
