// Experiment / regression harness: gemm_nt_wl_kernel (both operands through LDS-DMA) against the kernels it
// replaces, at the cross-encoder's shapes: time, TF/s and a bitwise comparison of the outputs.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o gemm_wl_bench scripts/exp/gemm_wl_bench.hip && ./gemm_wl_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../rag_inference_pipeline_amd/csrc/bert_kernels.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static void fill(float* d, size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

template <class F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static size_t diff_count(const float* a, const float* b, size_t n, float* maxabs) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; float mx = 0.f;
    for (size_t i = 0; i < n; ++i) {
        if (memcmp(&ha[i], &hb[i], 4)) ++bad;
        const float d = fabsf(ha[i] - hb[i]);
        if (d > mx || d != d) mx = d;
    }
    *maxabs = mx;
    return bad;
}

template <int MODE, int KS, int NS, bool PIPE = false>
static void launch_wl(const ragb::GemmWlParams& g) {
    using Geo = ragb::WlGeom<MODE, KS, NS>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ragb::gemm_nt_wl_kernel<MODE, KS, NS, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, 128, 128), 1, 1);
    hipLaunchKernelGGL((ragb::gemm_nt_wl_kernel<MODE, KS, NS, PIPE>), grid, dim3(256), Geo::LDS, 0, g);
}

template <int MODE, int NA, int NW>
static void launch_wl3(const ragb::GemmWlParams& g) {
    using Geo = ragb::Wl3Geom<MODE, NA, NW>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ragb::gemm_nt_wl3_kernel<MODE, NA, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, 128, 128), 1, 1);
    hipLaunchKernelGGL((ragb::gemm_nt_wl3_kernel<MODE, NA, NW>), grid, dim3(256), Geo::LDS, 0, g);
}

template <int MODE>
static void launch_wl4(const ragb::GemmWlParams& g) {
    using Geo = ragb::Wl4Geom<MODE>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ragb::gemm_nt_wl4_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, 128, 128), 1, 1);
    hipLaunchKernelGGL((ragb::gemm_nt_wl4_kernel<MODE>), grid, dim3(256), Geo::LDS, 0, g);
}

template <class F>
static void try_variant(const char* tag, F f, const float* ref, float* out, size_t n, double flop) {
    CK(hipMemset(out, 0, n * 4));
    const float t = time_ms(f);
    float mx; const size_t bad = diff_count(ref, out, n, &mx);
    printf("      %-14s %7.3f ms %6.1f TF diff %zu\n", tag, t, flop / t / 1e9, bad);
}

static void run(int M, int N, int K, int act, bool res, const char* name) {
    float *A, *W, *C0, *C1, *R, *b; __bf16* Wx; _Float16 *W16, *W16f;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C0, (size_t)M * N * 4));
    CK(hipMalloc(&C1, (size_t)M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&b, (size_t)N * 4));
    CK(hipMalloc(&Wx, (size_t)N * K * 6)); CK(hipMalloc(&W16, (size_t)N * K * 2)); CK(hipMalloc(&W16f, (size_t)N * K * 2));
    fill(A, (size_t)M * K, 1); fill(W, (size_t)N * K, 2); fill(R, (size_t)M * N, 3); fill(b, N, 4);
    const unsigned pg = (unsigned)(((size_t)N * K + 255) / 256);
    ragb::pack_x6_kernel<<<pg, 256>>>(W, N, K, K, Wx);
    ragb::pack_f16_frag_kernel<<<pg, 256>>>(W, N, K, K, W16f);
    {   // plain [N][K] fp16 copy for the old kernel (same rounding)
        std::vector<float> hw((size_t)N * K); std::vector<_Float16> h16((size_t)N * K);
        CK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hw.size(); ++i) h16[i] = (_Float16)hw[i];
        CK(hipMemcpy(W16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
    }
    const double flop = 2.0 * M * N * K;
    float mx;
    // ---- split-bf16
    ragb::GemmX6Params g6{A, Wx, b, res ? R : nullptr, C0, M, N, K, K, N, N, act};
    const float t_old6 = time_ms([&] { ragb::gemm_nt_x6_kernel<<<dim3(ragb::xcd_grid(M, N, 128, 128)), 256>>>(g6); });
    ragb::GemmWlParams w6{A, Wx, b, res ? R : nullptr, C1, M, N, K, K, N, N, act, nullptr};
    const float t_13 = time_ms([&] { launch_wl<0, 1, 3>(w6); });
    const size_t bad13 = diff_count(C0, C1, (size_t)M * N, &mx);
    CK(hipMemset(C1, 0, (size_t)M * N * 4));
    const float t_14 = time_ms([&] { launch_wl<0, 1, 3, true>(w6); });
    float mx4; const size_t bad14 = diff_count(C0, C1, (size_t)M * N, &mx4);
    printf("%-20s M=%6d N=%5d K=%5d | x6 old %7.3f ms %6.1f TF | wl<1,3> %7.3f ms %6.1f TF diff %zu (max %.3g) | wl<1,3,pipe> %7.3f ms %6.1f TF diff %zu\n",
           name, M, N, K, t_old6, flop / t_old6 / 1e9, t_13, flop / t_13 / 1e9, bad13, mx, t_14, flop / t_14 / 1e9, bad14);
    try_variant("wl3<x6,4,3>", [&] { launch_wl3<0, 4, 3>(w6); }, C0, C1, (size_t)M * N, flop);
    try_variant("wl4<x6>", [&] { launch_wl4<0>(w6); }, C0, C1, (size_t)M * N, flop);
    {   // two-plane fp16 split (MODE 2): not bit-identical by construction; distance from the split-bf16 result
        _Float16* W2; CK(hipMalloc(&W2, (size_t)N * K * 4));
        ragb::pack_f16x2_frag_kernel<<<pg, 256>>>(W, N, K, K, W2, nullptr);
        ragb::GemmWlParams w2{A, W2, b, res ? R : nullptr, C1, M, N, K, K, N, N, act, nullptr};
        CK(hipMemset(C1, 0, (size_t)M * N * 4));
        const float t23 = time_ms([&] { launch_wl<2, 1, 3>(w2); });
        float mx2; const size_t bad2 = diff_count(C0, C1, (size_t)M * N, &mx2);
        CK(hipMemset(C1, 0, (size_t)M * N * 4));
        const float t24 = time_ms([&] { launch_wl<2, 1, 4>(w2); });
        printf("      f16x2 split: wl<2;1,3> %7.3f ms %6.1f TF | wl<2;1,4> %7.3f ms %6.1f TF | vs x6: %zu differ, max |diff| %.3g\n",
               t23, flop / t23 / 1e9, t24, flop / t24 / 1e9, bad2, mx2);
        if (M < 10000) {   // small shape: both against a float64 evaluation (first 64 rows)
            std::vector<float> hA((size_t)64 * K), hW((size_t)N * K), hb(N), hR((size_t)64 * N), c6((size_t)64 * N), c3((size_t)64 * N);
            CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), b, hb.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c6.data(), C0, c6.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c3.data(), C1, c3.size() * 4, hipMemcpyDeviceToHost));
            double e6 = 0, e3 = 0;
            for (int m = 0; m < 64; ++m)
                for (int n = 0; n < N; ++n) {
                    double acc = 0;
                    for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
                    acc += hb[n]; if (res) acc += hR[(size_t)m * N + n];
                    e6 = fmax(e6, fabs(c6[(size_t)m * N + n] - acc)); e3 = fmax(e3, fabs(c3[(size_t)m * N + n] - acc));
                }
            printf("      max |error| against float64 (64 rows): split-bf16 x6 %.3g, f16x2 %.3g\n", e6, e3);
        }
        (void)hipFree(W2);
    }
    // ---- fp16
    ragb::GemmF16Params gf{A, W16, b, res ? R : nullptr, C0, M, N, K, K, K, N, N, act};
    const float t_oldf = time_ms([&] { ragb::gemm_nt_f16_kernel<<<dim3(ragb::xcd_grid(M, N, 128, 128)), 256>>>(gf); });
    ragb::GemmWlParams wf{A, W16f, b, res ? R : nullptr, C1, M, N, K, K, N, N, act, nullptr};
    CK(hipMemset(C1, 0, (size_t)M * N * 4));
    const float t_f23 = time_ms([&] { launch_wl<1, 2, 3>(wf); });
    const size_t badf = diff_count(C0, C1, (size_t)M * N, &mx);
    CK(hipMemset(C1, 0, (size_t)M * N * 4));
    const float t_f14 = time_ms([&] { launch_wl<1, 1, 4, true>(wf); });
    float mxb; const size_t badf2 = diff_count(C0, C1, (size_t)M * N, &mxb);
    printf("%-20s                          | f16 old %6.3f ms %6.1f TF | wl<2,3> %7.3f ms %6.1f TF diff %zu (max %.3g) | wl<1,4,pipe> %7.3f ms %6.1f TF diff %zu\n",
           "", t_oldf, flop / t_oldf / 1e9, t_f23, flop / t_f23 / 1e9, badf, mx, t_f14, flop / t_f14 / 1e9, badf2);
    CK(hipMemset(C0, 0, (size_t)M * N * 4));
    ragb::gemm_nt_f16_kernel<<<dim3(ragb::xcd_grid(M, N, 128, 128)), 256>>>(gf);
    try_variant("wl3<f16,4,3>", [&] { launch_wl3<1, 4, 3>(wf); }, C0, C1, (size_t)M * N, flop);
    try_variant("wl4<f16>", [&] { launch_wl4<1>(wf); }, C0, C1, (size_t)M * N, flop);
    fflush(stdout);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C0); (void)hipFree(C1); (void)hipFree(R); (void)hipFree(b);
    (void)hipFree(Wx); (void)hipFree(W16); (void)hipFree(W16f);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 178405;
    run(4096 + 37, 384, 384, ragb::ACT_NONE, true, "small check");
    run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM)");
    run(M, 384, 384, ragb::ACT_NONE, true, "attn out + residual");
    run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
    run(M, 384, 1536, ragb::ACT_NONE, true, "ffn2 + residual");
    run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
    run(M, 3072, 768, ragb::ACT_GELU_ERF, false, "ffn1 (base) + gelu");
    run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
    return 0;
}
