// Experiment: plain streaming write / copy bandwidth (float4 per lane, fully coalesced), to tell what a
// GEMM with a large fp32 output can hope for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill(f32x4* out, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = f32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void copy(const f32x4* in, f32x4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
    for (size_t mb : {256, 822, 4096}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        f32x4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int grid : {1024, 4096}) {
            fill<<<grid, 256>>>(b, n, 1.f);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) fill<<<grid, 256>>>(b, n, (float)i);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("write %5zu MB, grid %4d: %7.3f ms  %6.2f TB/s", mb, grid, ms, bytes / ms / 1e9);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) copy<<<grid, 256>>>(a, b, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("   | copy: %7.3f ms  %6.2f TB/s read + as much written\n", ms, bytes / ms / 1e9);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
