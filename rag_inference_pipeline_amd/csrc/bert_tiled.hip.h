// bert_tiled.hip.h — the non-GEMM kernels of the big-batch path over activations in MFMA-fragment order (gemm_wt.hip.h:
// Tiled<E>): embeddings + LayerNorm, LayerNorm, attention, first-token gather, pooling, and the copy out to row-major
// fp32.  Same arithmetic as their row-major counterparts in bert_kernels.hip.h (fp32 inside; E is only how a value
// is stored), reference: src/pipeline/components/reranker.py:248-252 and embedding.py:127-133 (the model's forward).
//
// A token's row is scattered over the fragments of its 32-token row block, so the row-wise kernels (embedding +
// LayerNorm, LayerNorm) work per ROW BLOCK with whole-fragment (1 KiB, coalesced) loads and stores: wave w of the block's
// workgroup takes the 16-feature steps w, w + 4, ..., a lane keeps its token's values of those steps in registers, and a
// token's sums are its two lanes' (one shuffle) over the four waves' (128 floats of LDS).  Only the copy out to row-major
// fp32 (untile_kernel) goes through an LDS row image (rows padded by four floats: fragment-shaped accesses — 32 lanes, 32
// different rows, the same columns — then fall into distinct banks).
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

// LDS image [32][H + kRowPad] fp32  <-  row block `rb` of a tiled [.][H] matrix; all 256 threads, whole fragments
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
// a lane's eight values of fragment (rb, ks): token 32 rb + (lane & 31), features 16 ks + 8 (lane >> 5) + 0..7
template <typename E>
__device__ __forceinline__ void frag_load8(const E* x, size_t frag, int lane, float (&v)[8]) {
    if constexpr (sizeof(E) == 2) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(x + frag * 512 + lane * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + frag * 512 + lane * 4);
        const f32x4 c = *reinterpret_cast<const f32x4*>(x + frag * 512 + 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = c[e]; }
    }
}
template <typename E>
__device__ __forceinline__ void frag_store8(E* x, size_t frag, int lane, const float (&v)[8]) {
    if constexpr (sizeof(E) == 2) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (_Float16)v[e];
        *reinterpret_cast<f16x8*>(x + frag * 512 + lane * 8) = t;
    } else {
        *reinterpret_cast<f32x4*>(x + frag * 512 + lane * 4) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(x + frag * 512 + 256 + lane * 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// Embeddings + LayerNorm, one workgroup (4 waves) per row block, built like ln_tiled_kernel below: wave w takes the
// 16-feature steps w, w + 4, ..., lane (r, h) gathers token r's eight features of a step from the word / position / type
// tables (32 contiguous bytes per table), keeps them in registers, and a token's sums are its two lanes' over the four
// waves'.  Rows past T are written as zeros.  NK = steps per wave = ceil(H / 64).  (The first form did the row
// arithmetic on an LDS row image with one wave per eight tokens: 1.2 TB/s; this one 3x that.)
template <typename E, int NK>
__global__ __launch_bounds__(256) void embed_ln_tiled_kernel(const EmbedParams p, E* out) {
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x, nks = p.H >> 4;
    const size_t f0 = (size_t)rb * nks;
    const int t = rb * 32 + r;
    const bool live = t < p.T;
    const float *w = nullptr, *pe = nullptr, *te = nullptr;
    if (live) {
        const int s = find_seq(p.cu, p.nseq, t);
        int pos = t - p.cu[s] + p.pos_offset;
        pos = pos < p.max_pos ? pos : p.max_pos - 1;
        int id = p.ids[t];
        id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
        int ty = p.type_ids ? p.type_ids[t] : 0;
        ty = ty < 0 ? 0 : (ty >= p.type_vocab ? p.type_vocab - 1 : ty);
        w = p.word_emb + (size_t)id * p.H;
        pe = p.pos_emb + (size_t)pos * p.H;
        te = p.type_emb ? p.type_emb + (size_t)ty * p.H : nullptr;
    }
    float v[NK][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int ks = wave + 4 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (ks < nks && live) {
            const int f = 16 * ks + 8 * h;
            f32x4 a0 = *reinterpret_cast<const f32x4*>(w + f) + *reinterpret_cast<const f32x4*>(pe + f);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(w + f + 4) + *reinterpret_cast<const f32x4*>(pe + f + 4);
            if (te) {
                a0 += *reinterpret_cast<const f32x4*>(te + f);
                a1 += *reinterpret_cast<const f32x4*>(te + f + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = a0[e]; v[i][4 + e] = a1[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[i][e];
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    if (h == 0) red[0][wave][r] = sum;
    __syncthreads();
    const float mean = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) / (float)p.H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NK; ++i)
        if (wave + 4 * i < nks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q = __builtin_fmaf(d, d, q);
            }
        }
    q += __shfl_xor(q, 32, 64);
    if (h == 0) red[1][wave][r] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) / (float)p.H + p.eps);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int ks = wave + 4 * i;
        if (ks < nks) {
            const int f = 16 * ks + 8 * h;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + f), g1 = *reinterpret_cast<const f32x4*>(p.ln_g + f + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + f), b1 = *reinterpret_cast<const f32x4*>(p.ln_b + f + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = live ? (v[i][e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]) : 0.f;
            frag_store8(out, f0 + ks, lane, o);
        }
    }
}

// y = LayerNorm(x) per token; x and y tiled (may be the same buffer).  One workgroup per row block, no row image:
// wave w takes the 16-feature steps w, w + 4, ... — whole fragments, 1 KiB per load and store instruction — and keeps
// its lanes' values in registers (H / 8 floats per lane); a token's sum is its two lanes' (shuffle) over the four
// waves' (128 floats of LDS), taken twice: mean, then the centred second moment, as the row-major kernel does.
// NK = steps per wave = ceil(H / 64).
template <typename E, int NK>
__global__ __launch_bounds__(256) void ln_tiled_kernel(const E* x, const float* g, const float* b, E* y, int T, int H, float eps) {
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x, nks = H >> 4;
    const size_t f0 = (size_t)rb * nks;
    float v[NK][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int ks = wave + 4 * i;
        if (ks < nks) {
            frag_load8(x, f0 + ks, lane, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    s += __shfl_xor(s, 32, 64);
    if (h == 0) red[0][wave][r] = s;
    __syncthreads();
    const float mean = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NK; ++i)
        if (wave + 4 * i < nks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q = __builtin_fmaf(d, d, q);
            }
        }
    q += __shfl_xor(q, 32, 64);
    if (h == 0) red[1][wave][r] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int ks = wave + 4 * i;
        if (ks < nks) {
            const int f = 16 * ks + 8 * h;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + f), g1 = *reinterpret_cast<const f32x4*>(g + f + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + f), b1 = *reinterpret_cast<const f32x4*>(b + f + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (v[i][e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
            frag_store8(y, f0 + ks, lane, o);
        }
    }
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

// ---- attention on the fp16 matrix cores (T16 only) --------------------------------------------------------------------
// One wave per (sequence, head, 32 query rows), the lane is the query, as above; but both products are
// v_mfma_f32_32x32x16_f16 — 4 (DH = 32) or 8 matrix instructions per 32-key tile where the fp32 form issues 32 or 64 —
// and the fragment-ordered layout feeds them directly:
//   Sᵀ[key][query] = K·Qᵀ     A = K fragment, B = Q fragment: both are 16 contiguous bytes per lane in memory
//                              (token = lane & 31, features 16 st + 8 (lane >> 5) ..), loaded straight into registers
//   softmax                    fp32, over the lane's 16 registers, its partner lane and the key tiles; 1/sqrt(dh) is
//                              applied to the fp32 scores
//   Oᵀ[dh][query] += Vᵀ·Pᵀ     B = the probabilities where they sit: registers 8 s .. 8 s + 7 of the score tile, rounded to
//                              fp16, are the k-step-s fragment with element j of lane half h = key 16 s + 8 (j >> 2) + 4 h +
//                              (j & 3) (cdna_hip_programming.md §3, "an accumulator tile as the next MFMA's operand");
//                              A = Vᵀ in that same key order, read from a row-major [key][dh] LDS tile with the hardware
//                              transpose ds_read_b64_tr_b16 (T10): lane 4 q + p of a 16-lane group addresses key row q,
//                              columns 4 p .. 4 p + 3, and receives its own column of the four rows.
// V rows are 64 bytes apart for DH = 32 and 192 for DH = 64: a 32-lane half then reads 4 rows x 64 bytes from 64 distinct
// banks.  EXEC is all ones at every transposed read (the only early exit is wave-uniform).
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));

// QB query blocks per wave (blockIdx.x covers 32 QB queries): the K fragments and the V tile of a key tile are loaded once
// and used by every one of them.  The product runs QB = 1: FETCH_SIZE shows each sequence's K and V re-read per 32 queries
// (0.84 GB per layer where qkv is 0.41) at 4.9 TB/s, yet QB = 2 — half those reads — measured 5 % SLOWER on both models
// (8.2 vs 7.8 ms, 44.2 vs 42.7 per 3200 pairs): the re-reads come from the Infinity Cache, and half as many waves with
// twice the serial work hide less latency.
template <int DH, int QB>
__global__ __launch_bounds__(64) void attention_t16_kernel(const _Float16* qkv, const int* cu, _Float16* ctx, float* ctx_first,
                                                           int H, int heads, float scale, int first_only) {
    using T = Tiled<_Float16>;
    constexpr int VLD = DH == 32 ? 32 : 96;   // halves per V row in LDS
    __shared__ __attribute__((aligned(16))) _Float16 Vs[32 * VLD];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int s = blockIdx.z, head = blockIdx.y, qb0 = blockIdx.x * QB;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    if (qb0 * 32 >= L) return;
    const int F = 3 * H;

    f16x8 qf[QB][DH / 16];
    f32x16 oT[QB][DH / 32];
    float mx[QB], den[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qidx = (qb0 + b) * 32 + r;
        const long long qm = t0 + (qidx < L ? qidx : L - 1);
#pragma unroll
        for (int st = 0; st < DH / 16; ++st)
            qf[b][st] = *reinterpret_cast<const f16x8*>(qkv + T::idx(qm, head * DH + 16 * st + 8 * h, F));
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oT[b][dt][i] = 0.f;
        mx[b] = -__builtin_inff();
        den[b] = 0.f;
    }
    // transposed read: this lane addresses row (lane & 15) >> 2 of a 4-key block, columns 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const int tr_off = ((lane & 15) >> 2) * VLD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    const int n_kt = (L + 31) / 32;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * 32, nk = min(32, L - k0);
        __syncthreads();   // the previous tile's reads of Vs are done
        for (int idx = lane; idx < 32 * (DH / 8); idx += 64) {
            const int j = idx & 31, c8 = (idx >> 5) * 8;
            f16x8 vv;
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = (_Float16)0.f;
            if (j < nk) vv = *reinterpret_cast<const f16x8*>(qkv + T::idx((long long)t0 + k0 + j, 2 * H + head * DH + c8, F));
            *reinterpret_cast<f16x8*>(&Vs[j * VLD + c8]) = vv;
        }
        f16x8 kf[DH / 16];
        {
            const long long km = (long long)t0 + k0 + (r < nk ? r : nk - 1);   // keys past the sequence: masked below
#pragma unroll
            for (int st = 0; st < DH / 16; ++st)
                kf[st] = *reinterpret_cast<const f16x8*>(qkv + T::idx(km, H + head * DH + 16 * st + 8 * h, F));
        }
        __syncthreads();

#pragma unroll
        for (int b = 0; b < QB; ++b) {
            if (b > 0 && (qb0 + b) * 32 >= L) break;   // (wave-uniform: EXEC stays all ones at the transposed reads)
            f32x16 sT;
#pragma unroll
            for (int i = 0; i < 16; ++i) sT[i] = 0.f;
#pragma unroll
            for (int st = 0; st < DH / 16; ++st) sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[st], qf[b][st], sT, 0, 0, 0);
            float tmax = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
                sT[i] = key < nk ? sT[i] * scale : -__builtin_inff();
                tmax = fmaxf(tmax, sT[i]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mx[b], tmax);
            const float alpha = expf(mx[b] - mnew);
            float psum = 0.f;
            f16x8 pfr[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const _Float16 ph = (_Float16)expf(sT[i] - mnew);   // the denominator sums what the product will use
                pfr[i >> 3][i & 7] = ph;
                psum += (float)ph;
            }
            psum += __shfl_xor(psum, 32, 64);
            den[b] = den[b] * alpha + psum;
            mx[b] = mnew;
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oT[b][dt][i] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8 pf = pfr[s2];
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt) {
                    const _Float16* base = Vs + (16 * s2 + 4 * h) * VLD + dt * 32 + tr_off;
                    const fp16x4_raw lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(base));
                    const fp16x4_raw hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(base + 8 * VLD));
                    const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
                    const f16x8 vf = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                    oT[b][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oT[b][dt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qidx = (qb0 + b) * 32 + r;
        const bool qvalid = first_only ? qidx == 0 : qidx < L;
        if (!qvalid) continue;
        const float inv = 1.0f / den[b];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {oT[b][dt][4 * g] * inv, oT[b][dt][4 * g + 1] * inv, oT[b][dt][4 * g + 2] * inv, oT[b][dt][4 * g + 3] * inv};
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
