// Experiment: gemm_nt_kernel<2,2> / gemm_nt_small_kernel timing at transformer shapes and at a long-K shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rag_inference_pipeline_amd/csrc/bert_kernels.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)


// Variant under test: same 128x128 tile, BK = 64 (half the barriers per flop), rows padded to 68.
__global__ __launch_bounds__(256) void gemm_bk64(const ragb::GemmParams p) {
    using ragb::f32x4; using ragb::f32x16;
    constexpr int BK = 64, LD = 68;
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Ws[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    const float* ag[8]; const float* wg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int am = m0 + srow + 16 * j; am = am < p.M ? am : p.M - 1;
        int wr = n0 + srow + 16 * j; wr = wr < p.N ? wr : p.N - 1;
        ag[j] = p.A + (size_t)am * p.lda + scol; wg[j] = p.W + (size_t)wr * p.ldw + scol;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    f32x4 ra[8], rw[8];
    const int nk = p.K / BK;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ra[j] = *(const f32x4*)(ag[j]); rw[j] = *(const f32x4*)(wg[j]); }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { *(f32x4*)&As[(srow + 16 * j) * LD + scol] = ra[j]; *(f32x4*)&Ws[(srow + 16 * j) * LD + scol] = rw[j]; }
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ra[j] = *(const f32x4*)(ag[j] + (size_t)(kt + 1) * BK); rw[j] = *(const f32x4*)(wg[j] + (size_t)(kt + 1) * BK); }
        }
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const f32x4*)&As[(wm * 64 + a * 32 + r) * LD + kg * 8 + 4 * h];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *(const f32x4*)&Ws[(wn * 64 + b * 32 + r) * LD + kg * 8 + 4 * h];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + r;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < p.M) {
                    float v = ragb::apply_act(acc[a][b][i] + bias, p.act);
                    if (p.R) v += p.R[(size_t)m * p.ldr + n];
                    p.C[(size_t)m * p.ldc + n] = v;
                }
            }
    }
}

// Variant under test: persistent 128x128x32 kernel — each workgroup walks output tiles (n fastest) and
// fetches the first K-tile of its NEXT output tile while it multiplies the last K-tile of the current one.
__global__ __launch_bounds__(256) void gemm_persist(const ragb::GemmParams p, int tiles_n, int n_tiles) {
    using ragb::f32x4; using ragb::f32x16;
    constexpr int BK = 32, LD = 36;
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Ws[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nk = p.K / BK;
    f32x4 ra[4], rw[4];
    auto load_tile = [&](int tile, int kt) {
        const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int am = m0 + srow + 32 * j; am = am < p.M ? am : p.M - 1;
            int wr = n0 + srow + 32 * j; wr = wr < p.N ? wr : p.N - 1;
            ra[j] = *(const f32x4*)(p.A + (size_t)am * p.lda + scol + kt * BK);
            rw[j] = *(const f32x4*)(p.W + (size_t)wr * p.ldw + scol + kt * BK);
        }
    };
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    load_tile(tile, 0);
    for (; tile < n_tiles; tile += gridDim.x) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) { *(f32x4*)&As[(srow + 32 * j) * LD + scol] = ra[j]; *(f32x4*)&Ws[(srow + 32 * j) * LD + scol] = rw[j]; }
            __syncthreads();
            if (kt + 1 < nk) load_tile(tile, kt + 1);
            else if (tile + (int)gridDim.x < n_tiles) load_tile(tile + gridDim.x, 0);
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg) {
                f32x4 af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = *(const f32x4*)&As[(wm * 64 + a * 32 + r) * LD + kg * 8 + 4 * h];
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = *(const f32x4*)&Ws[(wn * 64 + b * 32 + r) * LD + kg * 8 + 4 * h];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[b][t], af[a][t], acc[a][b], 0, 0, 0);
            }
        }
        const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                ragb::store_tile_rows(acc[a][b], m0 + wm * 64 + a * 32 + r, n0 + wn * 64 + b * 32, h, p.M, p.N, p.bias, p.R, p.ldr, p.C, p.ldc, p.act, false);
    }
}

static void fill(float* d, size_t n) {
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

static void run(int M, int N, int K, int act, bool res, const char* name) {
    float *A, *W, *C, *R, *b;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&b, (size_t)N * 4));
    fill(A, (size_t)M * K); fill(W, (size_t)N * K); fill(R, (size_t)M * N); fill(b, N);
    ragb::GemmParams g{A, W, b, res ? R : nullptr, C, M, N, K, K, K, N, N, act, K};
    dim3 grid(ragb::xcd_grid(M, N, 128, 128), 1, 1);  // gemm_nt_kernel<2,2> takes the 1-D XCD-aware grid
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) ragb::gemm_nt_kernel<2, 2><<<grid, 256>>>(g);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        if (rep) printf("%-22s M=%6d N=%5d K=%5d: %7.3f ms %6.1f TF/s", name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
    }
    const int tiles_n = (N + 127) / 128, n_tiles = tiles_n * ((M + 127) / 128);
    for (int pg : {768, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) gemm_persist<<<pg, 256>>>(g, tiles_n, n_tiles);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            if (rep) printf("   | persist(%d): %7.3f ms %6.1f TF/s", pg, ms, 2.0 * M * N * K / ms / 1e9);
        }
    }
    printf("\n");
    hipFree(A); hipFree(W); hipFree(C); hipFree(R); hipFree(b);
}

int main() {
    const int M = 178405;
    run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM)");
    run(M, 384, 384, ragb::ACT_NONE, true, "attn out + residual");
    run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
    run(M, 1536, 384, ragb::ACT_NONE, false, "ffn1 no act");
    run(M, 384, 1536, ragb::ACT_NONE, true, "ffn2 + residual");
    run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
    run(M, 3072, 768, ragb::ACT_GELU_ERF, false, "ffn1 (base) + gelu");
    run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
    run(4096, 4096, 4096, ragb::ACT_NONE, false, "4096^3");
    run(8192, 8192, 1024, ragb::ACT_NONE, false, "8192x8192x1024");
    return 0;
}
