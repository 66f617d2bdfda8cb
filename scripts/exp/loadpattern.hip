// Experiment: read-bandwidth of row-block access patterns on a 10M x 768 fp32 corpus.
//  pattern 0: fragment-shaped (lane -> row l&31, 16 B at col 8s + 4(l>>5))  [scan kernel v1]
//  pattern 1: line-shaped     (lane -> row 8j + (l>>3), 16 B piece l&7 of a 128-B line)
//  pattern 2: fully linear float4 stream (reference ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int PAT, int D>
__global__ __launch_bounds__(512) void rd(const float* X, long long n_rows, int d, int n_tiles, int n_iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = d / 8;            // steps of 8 cols (pattern 0) ; chunks of 32 cols = 4 steps
    int tile = blockIdx.x * 8 + wave;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < n_iters; ++it, tile += gridDim.x * 8) {
        int t = tile < n_tiles ? tile : n_tiles - 1;
        const float* base = X + (long long)t * 32 * d;
        if (PAT == 0) {
            const float* p = base + (long long)(lane & 31) * d + 4 * (lane >> 5);
            for (int s0 = 0; s0 < S; s0 += D) {
                f32x4 v[D];
#pragma unroll
                for (int i = 0; i < D; ++i) v[i] = *(const f32x4*)(p + 8 * (s0 + i));
#pragma unroll
                for (int i = 0; i < D; ++i) acc += v[i];
            }
        } else if (PAT == 1) {
            const float* p = base + (long long)(lane >> 3) * d + 4 * (lane & 7);
            for (int c0 = 0; c0 < S / 4; c0 += D / 4) {
                f32x4 v[D];
#pragma unroll
                for (int i = 0; i < D; ++i) v[i] = *(const f32x4*)(p + (long long)(8 * (i & 3)) * d + 32 * (c0 + (i >> 2)));
#pragma unroll
                for (int i = 0; i < D; ++i) acc += v[i];
            }
        } else {
            const float* p = base + 4 * lane;
            for (int s0 = 0; s0 < 32 * d / 256; s0 += D) {
                f32x4 v[D];
#pragma unroll
                for (int i = 0; i < D; ++i) v[i] = *(const f32x4*)(p + 256 * (s0 + i));
#pragma unroll
                for (int i = 0; i < D; ++i) acc += v[i];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int PAT, int D>
void run(const float* X, long long n, int d, float* sink, const char* name) {
    int n_tiles = (int)(n / 32), grid = 256;
    int n_iters = (n_tiles + grid * 8 - 1) / (grid * 8);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) rd<PAT, D><<<grid, 512>>>(X, n, d, n_tiles, n_iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        if (rep) printf("%-28s D=%2d: %.3f ms  %.0f GB/s\n", name, D, ms, 4.0 * n * d / (ms * 1e-3) / 1e9);
    }
}

int main() {
    const long long n = 10000000; const int d = 768;
    float *X, *sink; CK(hipMalloc(&X, n * d * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(X, 0x3c, n * d * 4));
    run<0, 8>(X, n, d, sink, "fragment 32x(32B)");
    run<0, 16>(X, n, d, sink, "fragment 32x(32B)");
    run<1, 8>(X, n, d, sink, "line 8x(128B)");
    run<1, 16>(X, n, d, sink, "line 8x(128B)");
    run<2, 8>(X, n, d, sink, "linear 1KB");
    run<2, 16>(X, n, d, sink, "linear 1KB");
    return 0;
}
