// bert_lnf.hip.h — the small kernels of "LayerNorm folded into its consumers" (gemm_wl.hip.h explains the scheme; the
// GEMM side lives in gemm_nt_ws_kernel's epilogue).  Query-encoder path only (<= 1024 packed tokens; reference
// src/pipeline/components/embedding.py:127-133, model.encode of a batch of queries).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ragb {

// Embeddings WITHOUT their LayerNorm: y0 = word + position + type, and per row the block statistics the consumers take
// (every 32-column block gets (row mean, row M2 / n_blocks): combined, that is the row's mean and M2 exactly — the
// wave has the whole row in registers, so it computes those two directly).  One wave per token, as embed_ln_kernel.
__global__ __launch_bounds__(256) void embed_pre_kernel(const EmbedParams p, float2* ts) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= p.T) return;
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
    const int nch = p.H >> 2;
    f32x4 v[kMaxChunks];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int ch = lane + 64 * j;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (ch < nch) {
            x = *reinterpret_cast<const f32x4*>(w + 4 * ch) + *reinterpret_cast<const f32x4*>(pe + 4 * ch);
            if (te) x += *reinterpret_cast<const f32x4*>(te + 4 * ch);
            *reinterpret_cast<f32x4*>(p.out + (size_t)t * p.H + 4 * ch) = x;
            sum += (x[0] + x[1]) + (x[2] + x[3]);
        }
        v[j] = x;
    }
    const float mean = wave_sum(sum) / (float)p.H;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j)
        if (lane + 64 * j < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dl = v[j][e] - mean;
                m2 = __builtin_fmaf(dl, dl, m2);
            }
        }
    m2 = wave_sum(m2);
    const int nblk = p.H >> 5;
    if (lane < nblk) ts[(size_t)t * nblk + lane] = make_float2(mean, m2 / (float)nblk);
}

// x[t] = LayerNorm(y[t]) from the block statistics: the one place a folded LayerNorm is materialised (the end of the
// encoder, in front of pooling / the hidden-state output).  One wave per token.
__global__ __launch_bounds__(256) void ln_from_tiles_kernel(const float* y, const float2* ts, const float* g, const float* b,
                                                           float* x, int T, int H, float eps, int blkw) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float mean, rstd;
    lnf_row_stats(ts + (size_t)t * (H / blkw), H / blkw, eps, mean, rstd, (float)blkw);
    const int nch = H >> 2;
    for (int ch = lane; ch < nch; ch += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)t * H + 4 * ch);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * ch);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * ch);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + bb[e];
        *reinterpret_cast<f32x4*>(x + (size_t)t * H + 4 * ch) = o;
    }
}

// out[s] = LayerNorm(y[first token of sequence s]) — the residual rows of a last layer that runs on first tokens only
__global__ __launch_bounds__(256) void gather_rows_ln_kernel(const float* y, const float2* ts, const float* g, const float* b,
                                                            const int* cu, float* out, int nseq, int H, float eps, int blkw) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseq) return;
    const int t = cu[s];
    float mean, rstd;
    lnf_row_stats(ts + (size_t)t * (H / blkw), H / blkw, eps, mean, rstd, (float)blkw);
    const int nch = H >> 2;
    for (int ch = lane; ch < nch; ch += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)t * H + 4 * ch);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * ch);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * ch);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + bb[e];
        *reinterpret_cast<f32x4*>(out + (size_t)s * H + 4 * ch) = o;
    }
}

// rs[t] = (mean, rstd) of row t from its tile statistics: one thread per row, one pass — so that the big-batch consumers
// read ONE value per row (by DMA) instead of combining the tiles themselves in front of their epilogue.
__global__ void lnf_finalize_kernel(const float2* ts, int nblk, float blkw, float eps, float2* rs, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float mean, rstd;
    lnf_row_stats(ts + (size_t)t * nblk, nblk, eps, mean, rstd, blkw);
    rs[t] = make_float2(mean, rstd);
}

// For a GEMM C = LN(y) W^T + bias with the LayerNorm folded (W' = W gamma in its image):
//   s[n] = sum_k W[n][k] gamma[k],   bias2[n] = bias[n] + sum_k W[n][k] beta[k].   One wave per output feature n.
__global__ __launch_bounds__(256) void fold_prep_kernel(const float* W, const float* bias, const float* gamma, const float* beta,
                                                       int N, int K, float* s_out, float* bias_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float a = 0.f, c = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        a = __builtin_fmaf(w, gamma[k], a);
        c = __builtin_fmaf(w, beta[k], c);
    }
    a = wave_sum(a);
    c = wave_sum(c);
    if (lane == 0) {
        s_out[n] = a;
        bias_out[n] = bias[n] + c;
    }
}

}  // namespace ragb
