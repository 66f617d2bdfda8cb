// Experiment / regression harness: gemm_nt_wh_kernel (256-row tiles, eight waves; gemm_wh.hip.h) against
// gemm_nt_wl_kernel at the cross-encoder's shapes: time, TF/s and a bitwise comparison of the outputs.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o gemm_wh_bench scripts/exp/gemm_wh_bench.hip && ./gemm_wh_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../rag_inference_pipeline_amd/csrc/bert_kernels.hip.h"
#include "gemm_wh_rowmajor.hip.h"
#include "gemm_wt16.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static std::vector<float> rnd(size_t n, unsigned seed, bool to_f16) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        float x = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22));
        if (to_f16) x = (float)(_Float16)x;
        h[i] = x;
    }
    return h;
}

template <class F>
static float time_ms(F f, int reps = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm up for ~150 ms of GPU time: after a host-side check the clocks are down, and five launches do not bring them back
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float one; CK(hipEventElapsedTime(&one, e0, e1));
    const int warm = (int)(150.0f / (one > 0.01f ? one : 0.01f)) + 1;
    for (int i = 0; i < warm; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int MODE, int KS, int NS, bool PIPE = false>
static void launch_wl(const ragb::GemmWlParams& g) {
    using Geo = ragb::WlGeom<MODE, KS, NS>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ragb::gemm_nt_wl_kernel<MODE, KS, NS, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, 128, 128), 1, 1);
    hipLaunchKernelGGL((ragb::gemm_nt_wl_kernel<MODE, KS, NS, PIPE>), grid, dim3(256), Geo::LDS, 0, g);
}

template <int AK, int NB, int NS>
static void launch_wh(const ragb::GemmWhParams& g) {
    using Geo = ragb::WhGeom<AK, NB, NS>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ragb::gemm_nt_wh_kernel<AK, NB, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, Geo::TM, Geo::TN), 1, 1);
    hipLaunchKernelGGL((ragb::gemm_nt_wh_kernel<AK, NB, NS>), grid, dim3(Geo::THREADS), Geo::LDS, 0, g);
}

template <int AK, int NW, int NB, int KS, int NS, int OCC, int ABL = 0>
static void launch_wt(const ragb::GemmWtParams& g) {
    using Geo = ragb::WtGeom<AK, NW, NB, KS, NS>;
    static bool once = false;
    auto fn = &ragb::gemm_nt_wt_kernel<AK, NW, NB, KS, NS, OCC, ABL>;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, Geo::TM, Geo::TN), 1, 1);
    hipLaunchKernelGGL(fn, grid, dim3(Geo::THREADS), Geo::LDS, 0, g);
}

template <int KS, int NS>
static void launch_wt16(const ragb::GemmWtParams& g) {
    using Geo = ragb::Wt16Geom<KS, NS>;
    static bool once = false;
    auto fn = &ragb::gemm_nt_wt16_kernel<KS, NS>;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS)); once = true; }
    dim3 grid(ragb::xcd_grid(g.M, g.N, Geo::TM, Geo::TN), 1, 1);
    hipLaunchKernelGGL(fn, grid, dim3(Geo::THREADS), Geo::LDS, 0, g);
}

static int g_mode = 3;   // 1: fp16 (T16) variants, 2: two-plane fp32 (T32) variants, 4: row-major 256-row kernels, 8: ablations

static void run(int M, int N, int K, int act, bool res, const char* name) {
    const size_t mk = (size_t)M * K, nk = (size_t)N * K, mn = (size_t)M * N;
    const int Mp = (M + 31) / 32 * 32;
    std::vector<float> hA = rnd(mk, 1, true), hW = rnd(nk, 2, false), hR = rnd(mn, 3, true), hb = rnd(N, 4, false);
    float *A, *W, *C0, *R, *b; _Float16 *W16f, *W2;
    CK(hipMalloc(&A, mk * 4)); CK(hipMalloc(&W, nk * 4)); CK(hipMalloc(&C0, mn * 4)); CK(hipMalloc(&R, mn * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMalloc(&W16f, nk * 2)); CK(hipMalloc(&W2, nk * 4));
    CK(hipMemcpy(A, hA.data(), mk * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), nk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, hR.data(), mn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
    const unsigned pg = (unsigned)((nk + 255) / 256);
    ragb::pack_f16_frag_kernel<<<pg, 256>>>(W, N, K, K, W16f);
    ragb::pack_f16x2_frag_kernel<<<pg, 256>>>(W, N, K, K, W2, nullptr);
    const double flop = 2.0 * M * N * K;
    printf("%-22s M=%6d N=%5d K=%5d\n", name, M, N, K);
    std::vector<float> c0(mn);

    if (g_mode & (1 | 4 | 8 | 32 | 512)) {
        // ---- fp16 activations against the fp32-activation fp16-input kernel (no residual there: it is added here in fp16)
        std::vector<_Float16> hA16(mk), hR16(mn);
        for (size_t i = 0; i < mk; ++i) hA16[i] = (_Float16)hA[i];
        for (size_t i = 0; i < mn; ++i) hR16[i] = (_Float16)hR[i];
        ragb::GemmWlParams wf{A, W16f, b, nullptr, C0, M, N, K, K, N, N, act, nullptr};
        const float t_ref = time_ms([&] { launch_wl<1, 2, 3>(wf); });
        CK(hipMemcpy(c0.data(), C0, mn * 4, hipMemcpyDeviceToHost));
        std::vector<_Float16> want(mn);
        for (size_t i = 0; i < mn; ++i) { _Float16 v = (_Float16)c0[i]; if (res) v = v + hR16[i]; want[i] = v; }
        printf("    row-major fp32 activations wl<1;2,3> 128x128  %7.3f ms %6.1f TF\n", t_ref, flop / t_ref / 1e9);
        if (g_mode & 4) {
            _Float16 *A16, *R16, *C16;
            CK(hipMalloc(&A16, mk * 2)); CK(hipMalloc(&R16, mn * 2)); CK(hipMalloc(&C16, mn * 2));
            CK(hipMemcpy(A16, hA16.data(), mk * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(R16, hR16.data(), mn * 2, hipMemcpyHostToDevice));
            std::vector<_Float16> got(mn);
            ragb::GemmWhParams gh{A16, W16f, b, res ? R16 : nullptr, C16, M, N, K, K, N, N, act, nullptr};
            auto check16 = [&](const char* tag, float t) {
                CK(hipMemcpy(got.data(), C16, mn * 2, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < mn; ++i) bad += memcmp(&got[i], &want[i], 2) != 0;
                printf("    %-48s %7.3f ms %6.1f TF  differ %zu\n", tag, t, flop / t / 1e9, bad);
                CK(hipMemset(C16, 0, mn * 2));
            };
            CK(hipMemset(C16, 0, mn * 2));
            if (N % 256 == 0) check16("row-major fp16 wh<1;NB=8,NS=3> 256x256", time_ms([&] { launch_wh<1, 8, 3>(gh); }));
            if (N % 192 == 0) check16("row-major fp16 wh<1;NB=6,NS=3> 256x192", time_ms([&] { launch_wh<1, 6, 3>(gh); }));
            check16("row-major fp16 wh<1;NB=4,NS=3> 256x128", time_ms([&] { launch_wh<1, 4, 3>(gh); }));
            check16("row-major fp16 wh<1;NB=4,NS=4> 256x128", time_ms([&] { launch_wh<1, 4, 4>(gh); }));
            (void)hipFree(A16); (void)hipFree(R16); (void)hipFree(C16);
        }
        // tiled (T16)
        using T = ragb::Tiled<_Float16>;
        std::vector<_Float16> tA((size_t)Mp * K, (_Float16)0.f), tR((size_t)Mp * N, (_Float16)0.f), tC((size_t)Mp * N);
        for (int m = 0; m < M; ++m) {
            for (int k = 0; k < K; ++k) tA[T::idx(m, k, K)] = hA16[(size_t)m * K + k];
            for (int n = 0; n < N; ++n) tR[T::idx(m, n, N)] = hR16[(size_t)m * N + n];
        }
        _Float16 *dA, *dR, *dC;
        CK(hipMalloc(&dA, tA.size() * 2)); CK(hipMalloc(&dR, tR.size() * 2)); CK(hipMalloc(&dC, tC.size() * 2));
        CK(hipMemcpy(dA, tA.data(), tA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, tR.data(), tR.size() * 2, hipMemcpyHostToDevice));
        ragb::GemmWtParams gt{dA, W16f, b, res ? dR : nullptr, dC, M, N, K, K, N, N, act, nullptr};
        auto checkt = [&](const char* tag, float t) {
            CK(hipMemcpy(tC.data(), dC, tC.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) bad += memcmp(&tC[T::idx(m, n, N)], &want[(size_t)m * N + n], 2) != 0;
            printf("    %-48s %7.3f ms %6.1f TF  differ %zu\n", tag, t, flop / t / 1e9, bad);
            CK(hipMemset(dC, 0, tC.size() * 2));
        };
        CK(hipMemset(dC, 0, tC.size() * 2));
        if (g_mode & 1) {
            checkt("T16 wt<NW=8,NB=4,KS=2,NS=3> 256x128 x2", time_ms([&] { launch_wt<1, 8, 4, 2, 3, 2>(gt); }));
            checkt("T16 wt<NW=8,NB=4,KS=1,NS=4> 256x128 x3", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3>(gt); }));
            checkt("T16 wt<NW=8,NB=4,KS=1,NS=5> 256x128 x3", time_ms([&] { launch_wt<1, 8, 4, 1, 5, 3>(gt); }));
            checkt("T16 wt<NW=4,NB=4,KS=2,NS=3> 128x128 x3", time_ms([&] { launch_wt<1, 4, 4, 2, 3, 3>(gt); }));
            if (N % 256 == 0) checkt("T16 wt<NW=4,NB=8,KS=2,NS=3> 128x256 x2", time_ms([&] { launch_wt<1, 4, 8, 2, 3, 2>(gt); }));
        }
        if ((g_mode & 512) && N % 256 == 0) {
            checkt("T16 wt<8,4,1,4,3> 256x128 x3 (product)", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3>(gt); }));
            checkt("T16 wt16<KS=2,NS=3> 256x256, 16 waves", time_ms([&] { launch_wt16<2, 3>(gt); }));
            checkt("T16 wt16<KS=2,NS=4> 256x256, 16 waves", time_ms([&] { launch_wt16<2, 4>(gt); }));
        }
        if (g_mode & 32) checkt("T16 wt<NW=8,NB=4,KS=1,NS=4> 256x128 x3", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3>(gt); }));
        if (g_mode & 8) {
            checkt("  ablation: A always from slab 0", time_ms([&] { launch_wt<1, 8, 4, 2, 3, 2, 1>(gt); }));
            checkt("  ablation: no output stores", time_ms([&] { launch_wt<1, 8, 4, 2, 3, 2, 2>(gt); }));
            checkt("  ablation: both", time_ms([&] { launch_wt<1, 8, 4, 2, 3, 2, 3>(gt); }));
            checkt("  KS=1,NS=4 x3 (product)", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3>(gt); }));
            checkt("  KS=1,NS=4 x3, one W LDS read per step", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3, 4>(gt); }));
            checkt("  KS=1,NS=4 x3, one W read + A resident + no stores", time_ms([&] { launch_wt<1, 8, 4, 1, 4, 3, 7>(gt); }));
        }
        (void)hipFree(dA); (void)hipFree(dR); (void)hipFree(dC);
    }

    if (g_mode & 256) {
        ragb::GemmWlParams w2{A, W2, b, res ? R : nullptr, C0, M, N, K, K, N, N, act, nullptr};
        const float t2 = time_ms([&] { launch_wl<2, 1, 3>(w2); });
        printf("    row-major two-plane wl<2;1,3> 128x128 x2       %7.3f ms %6.1f TF\n", t2, flop / t2 / 1e9);
    }
    if (g_mode & 2) {
        // ---- two fp16 planes, fp32 in memory: must equal gemm_nt_wl_kernel<2> bit for bit
        ragb::GemmWlParams w2{A, W2, b, res ? R : nullptr, C0, M, N, K, K, N, N, act, nullptr};
        const float t2 = time_ms([&] { launch_wl<2, 1, 3>(w2); });
        CK(hipMemcpy(c0.data(), C0, mn * 4, hipMemcpyDeviceToHost));
        printf("    row-major two-plane wl<2;1,3> 128x128 x2       %7.3f ms %6.1f TF\n", t2, flop / t2 / 1e9);
        using T = ragb::Tiled<float>;
        std::vector<float> tA((size_t)Mp * K, 0.f), tR((size_t)Mp * N, 0.f), tC((size_t)Mp * N);
        for (int m = 0; m < M; ++m) {
            for (int k = 0; k < K; ++k) tA[T::idx(m, k, K)] = hA[(size_t)m * K + k];
            for (int n = 0; n < N; ++n) tR[T::idx(m, n, N)] = hR[(size_t)m * N + n];
        }
        float *dA, *dR, *dC;
        CK(hipMalloc(&dA, tA.size() * 4)); CK(hipMalloc(&dR, tR.size() * 4)); CK(hipMalloc(&dC, tC.size() * 4));
        CK(hipMemcpy(dA, tA.data(), tA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, tR.data(), tR.size() * 4, hipMemcpyHostToDevice));
        ragb::GemmWtParams gt{dA, W2, b, res ? dR : nullptr, dC, M, N, K, K, N, N, act, nullptr};
        auto checkt = [&](const char* tag, float t) {
            CK(hipMemcpy(tC.data(), dC, tC.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) bad += memcmp(&tC[T::idx(m, n, N)], &c0[(size_t)m * N + n], 4) != 0;
            printf("    %-48s %7.3f ms %6.1f TF  differ %zu\n", tag, t, flop / t / 1e9, bad);
            CK(hipMemset(dC, 0, tC.size() * 4));
        };
        CK(hipMemset(dC, 0, tC.size() * 4));
        checkt("T32 wt<NW=4,NB=4,KS=1,NS=3> 128x128 x3", time_ms([&] { launch_wt<0, 4, 4, 1, 3, 3>(gt); }));
        checkt("T32 wt<NW=4,NB=4,KS=1,NS=4> 128x128 x2", time_ms([&] { launch_wt<0, 4, 4, 1, 4, 2>(gt); }));
        checkt("T32 wt<NW=8,NB=4,KS=1,NS=3> 256x128 x1", time_ms([&] { launch_wt<0, 8, 4, 1, 3, 1>(gt); }));
        if (N % 192 == 0) checkt("T32 wt<NW=4,NB=6,KS=1,NS=3> 128x192 x2", time_ms([&] { launch_wt<0, 4, 6, 1, 3, 2>(gt); }));
        checkt("T32 wt<NW=4,NB=2,KS=2,NS=3> 128x64 x4", time_ms([&] { launch_wt<0, 4, 2, 2, 3, 4>(gt); }));
        (void)hipFree(dA); (void)hipFree(dR); (void)hipFree(dC);
    }
    fflush(stdout);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C0); (void)hipFree(R); (void)hipFree(b);
    (void)hipFree(W16f); (void)hipFree(W2);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 178405;
    g_mode = argc > 2 ? atoi(argv[2]) : 3;
    if (g_mode == 512) {  // the 16-wave 256 x 256 experiment on the shapes whose N is a multiple of 256
        run(4096 + 37, 512, 384, ragb::ACT_NONE, true, "small check");
        run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
        run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
        run(M, 768, 768, ragb::ACT_NONE, true, "attn out (base)");
        run(M, 3072, 768, ragb::ACT_GELU_ERF, false, "ffn1 (base) + gelu");
        run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
        return 0;
    }
    if (g_mode == 128) {  // counter runs: the default mode's two-plane kernel on one shape, nothing else
        g_mode = 256;
        run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM), two-plane counters");
        return 0;
    }
    if (g_mode == 16) {   // counter runs: the product's T16 kernel on one shape, nothing else
        g_mode = 32;
        run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM), counters");
        return 0;
    }
    run(4096 + 37, 384, 384, ragb::ACT_NONE, true, "small check");
    run(2048 + 5, 256, 64, ragb::ACT_GELU_ERF, false, "short K, ragged M");
    run(M, 1152, 384, ragb::ACT_NONE, false, "qkv (MiniLM)");
    run(M, 384, 384, ragb::ACT_NONE, true, "attn out + residual");
    run(M, 1536, 384, ragb::ACT_GELU_ERF, false, "ffn1 + gelu");
    run(M, 384, 1536, ragb::ACT_NONE, true, "ffn2 + residual");
    run(M, 2304, 768, ragb::ACT_NONE, false, "qkv (base)");
    run(M, 3072, 768, ragb::ACT_GELU_ERF, false, "ffn1 (base) + gelu");
    run(M, 768, 3072, ragb::ACT_NONE, true, "ffn2 (base)");
    return 0;
}
