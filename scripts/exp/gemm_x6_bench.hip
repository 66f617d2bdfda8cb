// Experiment: gemm_nt_x6_kernel at the cross-encoder's shapes, with ablation builds
// (-DRAGB_X6_NO_MFMA / _NO_STAGE / _NO_WLOAD) to see which resource bounds it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rag_inference_pipeline_amd/csrc/bert_kernels.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static void fill(float* d, size_t n) {
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

static void run(int M, int N, int K, int act, bool res, const char* name) {
    float *A, *W, *C, *R, *b; __bf16* Wx;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&b, (size_t)N * 4)); CK(hipMalloc(&Wx, (size_t)N * K * 6));
    fill(A, (size_t)M * K); fill(W, (size_t)N * K); fill(R, (size_t)M * N); fill(b, N);
    ragb::pack_x6_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(W, N, K, K, Wx);
    ragb::GemmX6Params g{A, Wx, b, res ? R : nullptr, C, M, N, K, K, N, N, act};
    dim3 grid(ragb::xcd_grid(M, N, 128, 128), 1, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) ragb::gemm_nt_x6_kernel<<<grid, 256>>>(g);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        if (rep) printf("%-22s M=%6d N=%5d K=%5d: %7.3f ms %6.1f TF/s (fp32-equivalent)\n", name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
    }
#if defined(RAGB_X6_TRACE)
    {
        static unsigned long long h[4 * 128 * 5];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ragb::g_x6_trace), sizeof h));
        const int nk = K / 32 < 128 ? K / 32 : 128;
        printf("  trace of workgroup %d (cycles; per wave and K-tile: step 0, barrier wait, step 1 | whole iteration)\n", RAGB_X6_TRACE);
        for (int w = 0; w < 4; ++w) {
            printf("  wave %d:", w);
            for (int kt = 0; kt < nk && kt < 12; ++kt) {
                const unsigned long long* t = h + (w * 128 + kt) * 5;
                const unsigned long long next0 = kt + 1 < nk ? h[(w * 128 + kt + 1) * 5] : t[3];
                printf(" [%llu %llu %llu | %llu]", t[1] - t[0], t[2] - t[1], t[3] - t[2], next0 - t[0]);
            }
            printf("\n");
        }
    }
#endif
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(R); (void)hipFree(b); (void)hipFree(Wx);
}

int main() {
    const int M = 178405;
    run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM)");
    run(M, 384, 384, ragb::ACT_NONE, true, "attn out + residual");
    run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
    run(M, 384, 1536, ragb::ACT_NONE, true, "ffn2 + residual");
    run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
    run(M, 3072, 768, ragb::ACT_GELU_ERF, false, "ffn1 (base) + gelu");
    run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
    run(8192, 8192, 1024, ragb::ACT_NONE, false, "8192x8192x1024");
    return 0;
}
