// Experiment: aggregate bandwidth of 16-byte-per-lane loads from a buffer that fits the L2s (every
// workgroup sweeps the same `bytes` repeatedly), next to the same sweep over a buffer far larger than L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sweep(const f32x4* buf, size_t n_vec, size_t loads_per_thread, float* sink) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) % n_vec;
    const size_t step = stride % n_vec;
    for (size_t it = 0; it < loads_per_thread; it += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[u] = buf[i]; i += step; if (i >= n_vec) i -= n_vec; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = acc[0];
}

int main() {
    float* sink; CK(hipMalloc(&sink, 4));
    for (size_t mb : {1, 2, 3, 4, 8, 16, 32, 64, 256, 2048}) {
        const size_t bytes = mb << 20, n_vec = bytes / 16;
        f32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
        const int grid = 1024;
        const size_t loads = (8ull << 30) / 16 / ((size_t)grid * 256);   // ~8 GB delivered per launch
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        sweep<<<grid, 256>>>(buf, n_vec, 64, sink);
        CK(hipEventRecord(e0));
        sweep<<<grid, 256>>>(buf, n_vec, loads, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("buffer %5zu MB, 8 GB of 16-byte loads by 1024 workgroups: %8.3f ms  %7.2f TB/s delivered to the CUs\n", mb, ms,
               (double)loads * grid * 256 * 16 / ms / 1e9);
        CK(hipFree(buf));
    }
    return 0;
}
