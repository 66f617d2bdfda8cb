// bert_tiled.hip.h — the non-GEMM kernels of the big-batch path over activations in MFMA-fragment order (gemm_wt.hip.h:
// Tiled<E>): embeddings + LayerNorm, LayerNorm, attention, first-token gather, pooling, and the copy out to row-major
// fp32.  Same arithmetic as their row-major counterparts in bert_kernels.hip.h (fp32 inside; E is only how a value
// is stored), reference: src/pipeline/components/reranker.py:248-252 and embedding.py:127-133 (the model's forward).
//
// A token's row is scattered over the fragments of its 32-token row block, so the row-wise kernels work per ROW BLOCK:
// a workgroup moves the block between memory and a row-major LDS image with whole-fragment (1 KiB, coalesced)
// transfers and does the row arithmetic on the LDS image, one wave per eight tokens, exactly as the row-major kernels
// do it in registers.  LDS rows are padded by four floats: fragment-shaped accesses (32 lanes, 32 different rows, the
// same columns) then fall into distinct banks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

// four / eight consecutive features of one token (f % 4 == 0 / f % 8 == 0): contiguous in both layouts
template <typename E>
__device__ __forceinline__ f32x4 tiled_load4(const E* x, long long m, int f, int F) {
    if constexpr (sizeof(E) == 2) {
        const f16x4 v = *reinterpret_cast<const f16x4*>(x + Tiled<E>::idx(m, f, F));
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(x + Tiled<E>::idx(m, f, F));
    }
}
template <typename E>
__device__ __forceinline__ void tiled_store4(E* x, long long m, int f, int F, const f32x4 v) {
    if constexpr (sizeof(E) == 2) {
        const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4*>(x + Tiled<E>::idx(m, f, F)) = hv;
    } else {
        *reinterpret_cast<f32x4*>(x + Tiled<E>::idx(m, f, F)) = v;
    }
}

constexpr int kRowPad = 4;   // floats

// LDS image [32][H + kRowPad] fp32  <->  row block `rb` of a tiled [.][H] matrix; all 256 threads, whole fragments
template <typename E>
__device__ __forceinline__ void block_to_lds(const E* x, int rb, int H, float* rows, int tid) {
    const int ld = H + kRowPad, nks = H >> 4;
    for (int i = tid; i < nks * 64; i += 256) {
        const int ks = i >> 6, l = i & 63, r = l & 31, h = l >> 5;
        float* dst = rows + r * ld + 16 * ks + 8 * h;
        if constexpr (sizeof(E) == 2) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)rb * nks + ks) * 512 + l * 8);
            *reinterpret_cast<f32x4*>(dst) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        } else {
            const float* src = x + ((size_t)rb * nks + ks) * 512 + l * 4;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
            *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src + 256);
        }
    }
}
template <typename E>
__device__ __forceinline__ void lds_to_block(const float* rows, E* x, int rb, int H, int tid) {
    const int ld = H + kRowPad, nks = H >> 4;
    for (int i = tid; i < nks * 64; i += 256) {
        const int ks = i >> 6, l = i & 63, r = l & 31, h = l >> 5;
        const float* src = rows + r * ld + 16 * ks + 8 * h;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        if constexpr (sizeof(E) == 2) {
            const f16x8 v = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3],
                             (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
            *reinterpret_cast<f16x8*>(x + ((size_t)rb * nks + ks) * 512 + l * 8) = v;
        } else {
            float* dst = x + ((size_t)rb * nks + ks) * 512 + l * 4;
            *reinterpret_cast<f32x4*>(dst) = a;
            *reinterpret_cast<f32x4*>(dst + 256) = b;
        }
    }
}

// Embeddings + LayerNorm, one workgroup (4 waves) per row block; the per-token arithmetic is embed_ln_kernel's.
// Rows past T are written as zeros.  Dynamic LDS: 32 (H + 4) floats.
template <typename E>
__global__ __launch_bounds__(256) void embed_ln_tiled_kernel(const EmbedParams p, E* out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* rows = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = blockIdx.x, ld = p.H + kRowPad;
    const int nch = p.H >> 2;
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i, t = rb * 32 + row;
        float* o = rows + row * ld;
        if (t >= p.T) {
            for (int ch = lane; ch < nch; ch += 64) *reinterpret_cast<f32x4*>(o + 4 * ch) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int s = find_seq(p.cu, p.nseq, t);
        int pos = t - p.cu[s] + p.pos_offset;
        pos = pos < p.max_pos ? pos : p.max_pos - 1;
        int id = p.ids[t];
        id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
        int ty = p.type_ids ? p.type_ids[t] : 0;
        ty = ty < 0 ? 0 : (ty >= p.type_vocab ? p.type_vocab - 1 : ty);
        const float* w = p.word_emb + (size_t)id * p.H;
        const float* pe = p.pos_emb + (size_t)pos * p.H;
        const float* te = p.type_emb ? p.type_emb + (size_t)ty * p.H : nullptr;
        f32x4 v[kMaxChunks];
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int ch = lane + 64 * j;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (ch < nch) {
                x = *reinterpret_cast<const f32x4*>(w + 4 * ch) + *reinterpret_cast<const f32x4*>(pe + 4 * ch);
                if (te) x += *reinterpret_cast<const f32x4*>(te + 4 * ch);
            }
            v[j] = x;
        }
        ln_store(v, lane, p.H, p.eps, p.ln_g, p.ln_b, o);
    }
    __syncthreads();
    lds_to_block(rows, out, rb, p.H, tid);
}

// y = LayerNorm(x) per token; x and y tiled (may be the same buffer: a row block is read whole before it is written).
template <typename E>
__global__ __launch_bounds__(256) void ln_tiled_kernel(const E* x, const float* g, const float* b, E* y, int T, int H, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* rows = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = blockIdx.x, ld = H + kRowPad;
    const int nch = H >> 2;
    block_to_lds(x, rb, H, rows, tid);
    __syncthreads();
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i, t = rb * 32 + row;
        if (t >= T) continue;     // (rows past T keep whatever the block held: nothing valid depends on them)
        float* o = rows + row * ld;
        f32x4 v[kMaxChunks];
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int ch = lane + 64 * j;
            v[j] = ch < nch ? *reinterpret_cast<const f32x4*>(o + 4 * ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ln_store(v, lane, H, eps, g, b, o);
    }
    __syncthreads();
    lds_to_block(rows, y, rb, H, tid);
}

// attention_mfma_kernel (bert_kernels.hip.h) over a tiled qkv [T][3H]: same products, same softmax, same order of
// operations; only the addresses differ.  ctx is tiled [T][H]; with first_only it is the row-major fp32 [nseq][H]
// array the last layer's first-token path continues from.
template <int DH, typename E>
__global__ __launch_bounds__(64) void attention_tiled_kernel(const E* qkv, const int* cu, E* ctx, float* ctx_first, int H,
                                                             int heads, float scale, int first_only) {
    constexpr int KLD = DH + 4;
    __shared__ __attribute__((aligned(16))) float Ks[32 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[32 * DH];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int s = blockIdx.z, head = blockIdx.y, qb = blockIdx.x;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    if (qb * 32 >= L) return;
    const int qidx = qb * 32 + r;
    const bool qvalid = first_only ? qidx == 0 : qidx < L;
    const int F = 3 * H;

    f32x4 qf[DH / 8];
    {
        const long long qm = t0 + (qvalid ? qidx : L - 1);
#pragma unroll
        for (int st = 0; st < DH / 8; ++st) {
            f32x4 v = tiled_load4(qkv, qm, head * DH + 8 * st + 4 * h, F);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = qvalid ? v[e] * scale : 0.f;
            qf[st] = v;
        }
    }
    f32x16 oT[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oT[dt][i] = 0.f;
    float mx = -__builtin_inff(), den = 0.f;

    const int n_kt = (L + 31) / 32;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * 32, nk = min(32, L - k0);
        __syncthreads();
        // stage the K / V tile: lanes run over keys first (neighbouring tokens are neighbours in a fragment)
        for (int idx = lane; idx < 32 * (DH / 4); idx += 64) {
            const int j = idx & 31, c = (idx >> 5) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (j < nk) {
                kv = tiled_load4(qkv, (long long)t0 + k0 + j, H + head * DH + c, F);
                vv = tiled_load4(qkv, (long long)t0 + k0 + j, 2 * H + head * DH + c, F);
            }
            *reinterpret_cast<f32x4*>(&Ks[j * KLD + c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[j * DH + c]) = vv;
        }
        __syncthreads();

        f32x16 sT;
#pragma unroll
        for (int i = 0; i < 16; ++i) sT[i] = 0.f;
#pragma unroll
        for (int st = 0; st < DH / 8; ++st) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[r * KLD + 8 * st + 4 * h]);
#pragma unroll
            for (int t = 0; t < 4; ++t) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[st][t], sT, 0, 0, 0);
        }
        float tmax = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
            if (key >= nk) sT[i] = -__builtin_inff();
            tmax = fmaxf(tmax, sT[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mx, tmax);
        const float alpha = expf(mx - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sT[i] = expf(sT[i] - mnew);
            psum += sT[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * alpha + psum;
        mx = mnew;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oT[dt][i] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const float vf = Vs[key * DH + dt * 32 + r];
                oT[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sT[i], oT[dt], 0, 0, 0);
            }
        }
    }
    if (qvalid) {
        const float inv = 1.0f / den;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {oT[dt][4 * g] * inv, oT[dt][4 * g + 1] * inv, oT[dt][4 * g + 2] * inv, oT[dt][4 * g + 3] * inv};
                const int f = head * DH + dt * 32 + 8 * g + 4 * h;
                if (first_only)
                    *reinterpret_cast<f32x4*>(ctx_first + (size_t)s * H + f) = v;
                else
                    tiled_store4(ctx, (long long)t0 + qidx, f, H, v);
            }
    }
}

// out[s][c] = x(cu[s], c): the sequences' first tokens, row-major fp32
template <typename E>
__global__ void gather_rows_tiled_kernel(const E* x, const int* cu, float* out, int nseq, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * H) return;
    const int s = i / H, c = i - s * H;
    out[i] = (float)x[Tiled<E>::idx(cu[s], c, H)];
}

// out[t][c] = x(t, c): tiled -> row-major fp32 (RAG_BERT_OUT_HIDDEN); one workgroup per row block
template <typename E>
__global__ __launch_bounds__(256) void untile_kernel(const E* x, float* out, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* rows = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, rb = blockIdx.x, ld = H + kRowPad;
    block_to_lds(x, rb, H, rows, tid);
    __syncthreads();
    const int nch = H >> 2;
    for (int i = tid; i < 32 * nch; i += 256) {
        const int row = i / nch, ch = i - row * nch, t = rb * 32 + row;
        if (t < T) *reinterpret_cast<f32x4*>(out + (size_t)t * H + 4 * ch) = *reinterpret_cast<const f32x4*>(rows + row * ld + 4 * ch);
    }
}

// pool_kernel's modes 0 (mean over the sequence's tokens) and 1 (first token) over a tiled x
template <typename E>
__global__ __launch_bounds__(256) void pool_tiled_kernel(const E* x, const int* cu, float* out, int H, int mode, int normalize) {
    __shared__ float red[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        float a = 0.f;
        if (c < H) {
            if (mode == 0) {
                for (int t = 0; t < L; ++t) a += (float)x[Tiled<E>::idx(t0 + t, c, H)];
                a /= (float)L;
            } else {
                a = (float)x[Tiled<E>::idx(t0, c, H)];
            }
        }
        v[j] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(tot), 1e-12f) : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        if (c < H) out[(size_t)s * H + c] = v[j] * inv;
    }
}

}  // namespace ragb
