// ivf_kernels.hip.h — list scan of the IVFFlat `nprobe` mode (rag_ivf_* in include/rag_amd.h).
//
// The reference's own data generator writes a faiss IndexIVFFlat (scripts/create_test_docs.py:83-104: nlist 4096,
// nprobe 64) and FAISSStore.load sets `index.nprobe` (faiss_store.py:84-92): on such a file the reference does NOT
// search exhaustively — a query only sees the rows of the `nprobe` inverted lists whose centroids are nearest to it.
// This kernel is step 2 of that search (step 1, the coarse quantizer, is the flat scan over the nlist centroids):
// one workgroup per (query, probed list) scores every row of the list with the canonical dot product of the flat
// search (flat_kernels.hip.h; oracle/flat_oracle.c:rago_dot) and emits the list's best k ranking keys; the tournament
// merge of the flat search then ranks the nprobe lists of a query.  Ranking is (score, ascending stored id) — faiss
// leaves ties to its heap's visiting order; the candidate SET (the union of the probed lists) is what the mode is about.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragk {

constexpr int kIvfMaxK = 256;   // keys a workgroup keeps per (query, list)

struct IvfScanParams {
    const float* X;            // [n][row_stride] rows in LIST order, columns zero-padded to d8
    long long row_stride;
    const float* xnorm;        // [n] canonical squared norms (L2), list order
    const uint32_t* ids;       // [n] stored id of each list-order row
    const long long* list_off; // [nlist + 1]
    const float* Q;            // [nq][d]
    const float* qnorm;        // [nq] (L2)
    const long long* probe;    // [nq][nprobe] list numbers from the coarse search, -1 = none
    u64* partial;              // [nq][nprobe][k] ranking keys, best first, 0 = empty
    int d, d8, nprobe, k, l2;
};

// 512 keys in LDS, descending, by 256 threads (bitonic network: 45 compare-exchange steps)
__device__ __forceinline__ void ivf_sort512_desc(u64* keys, int t) {
    for (int size = 2; size <= 512; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            const int lo = 2 * t - (t & (stride - 1));   // the pair (lo, lo + stride)
            const bool desc = (lo & size) == 0;
            const u64 a = keys[lo], b = keys[lo + stride];
            if ((a < b) == desc) {
                keys[lo] = b;
                keys[lo + stride] = a;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void ivf_list_scan_kernel(const IvfScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char ivf_smem[];
    u64* keys = reinterpret_cast<u64*>(ivf_smem);                 // [512]: best 256 | this chunk's 256
    float* qv = reinterpret_cast<float*>(ivf_smem + 512 * 8);     // [d8]
    const int t = threadIdx.x, slot = blockIdx.x, q = blockIdx.y;
    u64* out = p.partial + ((size_t)q * p.nprobe + slot) * p.k;
    const long long list = p.probe[(size_t)q * p.nprobe + slot];
    const bool dead_query = p.l2 && !(p.qnorm[q] <= 3.402823466e+38f);   // every distance is inf or NaN: no results
    if (list < 0 || dead_query) {   // workgroup-uniform
        for (int i = t; i < p.k; i += 256) out[i] = 0ull;
        return;
    }
    for (int c = t; c < p.d8; c += 256) qv[c] = c < p.d ? p.Q[(size_t)q * p.d + c] : 0.f;
    keys[t] = 0ull;
    const long long r0 = p.list_off[list], n = p.list_off[list + 1] - r0;
    __syncthreads();
    const int d4 = p.d8 >> 2;
    const f32x4* q4 = reinterpret_cast<const f32x4*>(qv);
    for (long long base = 0; base < n; base += 256) {
        u64 key = 0ull;
        const long long row = r0 + base + t;
        if (base + t < n) {
            // the canonical chain: groups of 8 visited 0,4,1,5,2,6,3,7, one fmaf per term, start +0
            const f32x4* x4 = reinterpret_cast<const f32x4*>(p.X + row * p.row_stride);
            float acc = 0.f;
            f32x4 xa = x4[0], xb = x4[1];
            for (int s = 2; s <= d4; s += 2) {   // the next group's row values are requested under this group's chain
                const int sn = s < d4 ? s : 0;
                const f32x4 xa_n = x4[sn], xb_n = x4[sn + 1];
                const f32x4 qa = q4[s - 2], qb = q4[s - 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(xa[e], qa[e], acc);
                    acc = __builtin_fmaf(xb[e], qb[e], acc);
                }
                xa = xa_n;
                xb = xb_n;
            }
            float score = acc + 0.0f;
            if (p.l2) score = __builtin_fmaf(2.0f, acc, -p.xnorm[row]) + 0.0f;
            if (score >= RAGK_SCORE_FLOOR) key = make_key(score, p.ids[row]);   // NaN, -inf, -FLT_MAX: never a result
        }
        keys[256 + t] = key;
        ivf_sort512_desc(keys, t);   // the best 256 of (best so far, this chunk) are in front again
    }
    for (int i = t; i < p.k; i += 256) out[i] = keys[i];
}

}  // namespace ragk
