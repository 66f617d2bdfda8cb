// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Records (XCC_ID, SE_ID, CU_ID) of every workgroup.
//   hipcc -O2 --offload-arch=gfx950 -o cumask_probe scripts/exp/cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void where(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);        // stay resident so that the grid spreads
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15) << 16) | (hw & 0xffff);
}

static void run(const char* tag, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int n = 4096;
    unsigned* d; CK(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int per_xcc[8] = {0};
    for (unsigned v : h) { const unsigned xcc = v >> 16, se = (v >> 13) & 7, cu = (v >> 8) & 15; cus.insert((xcc << 8) | (se << 4) | cu); }
    for (unsigned c : cus) per_xcc[c >> 8]++;
    int bits = 0; for (uint32_t m : mask) bits += __builtin_popcount(m);
    printf("%-34s mask bits %3d -> distinct (xcc, se, cu) %3zu   per XCC:", tag, bits, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    CK(hipFree(d)); CK(hipStreamDestroy(st));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("multiProcessorCount %d\n", p.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256", all);
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffff; run("bits 0..15", m); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffff; run("bits 0..31", m); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 8; ++i) m[i] = 0x3; run("bits 32 i + {0,1}", m); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 16) m[i / 32] |= 1u << (i % 32); run("every 16th bit", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = 0xffff0000u; run("all but bits 0..15", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 8; ++i) m[i] = 0xfffffffcu; run("all but 32 i + {0,1}", m); }
    return 0;
}
