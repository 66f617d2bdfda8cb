/*
 * flat_oracle.c — CPU restatement of the reference's flat-index search.  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (rag_inference_pipeline_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic on this path lives in faiss-cpu 1.13.1
 * (uv.lock:959-960), which is not vendored under /root/reference and is not installed here,
 * and none of the reference's tests assert a search result (tests/test_components.py:138-157
 * use a MagicMock; tests/test_retrieval_service.py:245-284 check size/exceptions only).  This
 * file therefore restates the *published* IndexFlat semantics that the reference's call site
 * relies on (src/pipeline/components/faiss_store.py:113-158 -> index.search(embeddings, k)):
 *
 *   - exhaustive fp32 inner product (IndexFlatIP) or squared L2 distance (IndexFlatL2)
 *     of every query against every stored row;
 *   - per query the k best, sorted best-first (IP descending, L2 ascending);
 *   - labels are insertion row numbers (int64); when k > ntotal the tail is label -1 with
 *     distance -FLT_MAX (IP) / +FLT_MAX (L2) — the heap's neutral element.
 *
 * Where FAISS leaves behaviour unspecified this oracle fixes it, and the HIP path follows:
 *   - summation order of a dot product: "canonical order" below (FAISS's own order depends on
 *     BLAS and on nq, so it is not reproducible across batch sizes either);
 *   - ties: equal score -> smaller id first;
 *   - L2: dist = ||q||^2 - (2*ip - ||x||^2), clamped at 0 (FAISS's BLAS path clamps too).
 * Non-finite values follow the heap FAISS selects with: a row whose ranking score is NaN, -inf or
 * -FLT_MAX is never returned (rago_search below); +inf scores rank first like any other value.
 *
 * Build: make -C oracle   (gcc -O3 -mavx2 -mfma -fopenmp, -ffp-contract=off)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define RAGO_SIMD 1
#else
#define RAGO_SIMD 0
#endif

#define RAGO_METRIC_IP 0
#define RAGO_METRIC_L2 1
#define RAGO_QT 32 /* queries per register tile, as the HIP kernel */

/* ---- canonical arithmetic --------------------------------------------------------------- */

/* Canonical dot product: one fp32 fmaf chain, start +0, over the dimension padded with zeros to
 * a multiple of 8, visiting each group of 8 as 0,4,1,5,2,6,3,7.  (That is the order in which
 * v_mfma_f32_32x32x2_f32 consumes a 16-byte-per-lane fragment; every step is one correctly
 * rounded fma, so the result is a pure function of the two vectors.) */
float rago_dot(const float* x, const float* q, int32_t d) {
    float acc = 0.0f;
    int32_t d8 = (d + 7) & ~7;
    for (int32_t s = 0; s < d8; s += 8) {
        for (int32_t t = 0; t < 4; ++t) {
            int32_t k0 = s + t, k1 = s + 4 + t;
            float x0 = k0 < d ? x[k0] : 0.0f, q0 = k0 < d ? q[k0] : 0.0f;
            float x1 = k1 < d ? x[k1] : 0.0f, q1 = k1 < d ? q[k1] : 0.0f;
            acc = fmaf(x0, q0, acc);
            acc = fmaf(x1, q1, acc);
        }
    }
    return acc;
}

/* Squared norm in the same canonical order (used for the L2 metric). */
float rago_sqnorm(const float* x, int32_t d) { return rago_dot(x, x, d); }

/* fp64 truth for near-tie classification in the tests. */
double rago_dot_f64(const float* x, const float* q, int32_t d) {
    double acc = 0.0;
    for (int32_t k = 0; k < d; ++k) acc += (double)x[k] * (double)q[k];
    return acc;
}

/* Ranking score (larger is better) from the canonical inner product.
 * IP: ip + 0 (folds -0 into +0).  L2: 2*ip - ||x||^2, one rounding. */
static inline float rank_score(float ip, float xn, int32_t metric) {
    return metric == RAGO_METRIC_IP ? ip + 0.0f : fmaf(2.0f, ip, -xn) + 0.0f;
}

/* Monotone map float -> uint32 (larger float, larger integer). */
static inline uint32_t ord32(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unord32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
/* 64-bit ranking key: larger key = better (higher score, then smaller row). 0 = empty. */
static inline uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)ord32(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

/* ---- top-k list -------------------------------------------------------------------------- */

typedef struct {
    uint64_t* keys; /* sorted descending, n valid */
    int32_t n, k;
} topk_t;

static inline void topk_push(topk_t* t, uint64_t key) {
    if (t->n == t->k) {
        if (key <= t->keys[t->k - 1]) return;
    } else {
        t->n++;
    }
    int32_t i = t->n - 1;
    while (i > 0 && t->keys[i - 1] < key) {
        t->keys[i] = t->keys[i - 1];
        --i;
    }
    t->keys[i] = key;
}

/* ---- scan ------------------------------------------------------------------------------- */

/* Scores of one row against up to 32 queries, canonical order; Qt is [d8][32] (transposed,
 * zero padded).  Bit-identical to rago_dot per query. */
static void row_scores(const float* x, int32_t d, int32_t d8, const float* Qt, float* out32) {
#if RAGO_SIMD
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    for (int32_t s = 0; s < d8; s += 8) {
        for (int32_t t = 0; t < 4; ++t) {
            for (int32_t h = 0; h < 2; ++h) {
                int32_t k = s + 4 * h + t;
                __m256 xv = _mm256_set1_ps(k < d ? x[k] : 0.0f);
                const float* qr = Qt + (size_t)k * RAGO_QT;
                a0 = _mm256_fmadd_ps(xv, _mm256_loadu_ps(qr), a0);
                a1 = _mm256_fmadd_ps(xv, _mm256_loadu_ps(qr + 8), a1);
                a2 = _mm256_fmadd_ps(xv, _mm256_loadu_ps(qr + 16), a2);
                a3 = _mm256_fmadd_ps(xv, _mm256_loadu_ps(qr + 24), a3);
            }
        }
    }
    _mm256_storeu_ps(out32, a0);
    _mm256_storeu_ps(out32 + 8, a1);
    _mm256_storeu_ps(out32 + 16, a2);
    _mm256_storeu_ps(out32 + 24, a3);
#else
    float acc[RAGO_QT];
    for (int b = 0; b < RAGO_QT; ++b) acc[b] = 0.0f;
    for (int32_t s = 0; s < d8; s += 8)
        for (int32_t t = 0; t < 4; ++t)
            for (int32_t h = 0; h < 2; ++h) {
                int32_t k = s + 4 * h + t;
                float xv = k < d ? x[k] : 0.0f;
                const float* qr = Qt + (size_t)k * RAGO_QT;
                for (int b = 0; b < RAGO_QT; ++b) acc[b] = fmaf(xv, qr[b], acc[b]);
            }
    memcpy(out32, acc, sizeof acc);
#endif
}

/* Full score matrix (nq x N ranking scores), small cases only. */
int rago_scores(const float* X, int64_t N, int32_t d, const float* Q, int32_t nq, int32_t metric,
                float* out /* nq x N */) {
    for (int32_t b = 0; b < nq; ++b)
        for (int64_t r = 0; r < N; ++r) {
            const float* x = X + (size_t)r * d;
            float xn = metric == RAGO_METRIC_L2 ? rago_sqnorm(x, d) : 0.0f;
            out[(size_t)b * N + r] = rank_score(rago_dot(x, Q + (size_t)b * d, d), xn, metric);
        }
    return 0;
}

/* Exact search, the oracle proper.  Follows faiss_store.py:152 index.search(embeddings, k).
 * nthreads <= 0 -> OpenMP default.  Returns 0, or -1 on bad arguments / allocation failure. */
int rago_search(const float* X, int64_t N, int32_t d, int32_t metric, const float* Q, int32_t nq,
                int32_t k, int64_t id_offset, int32_t nthreads, float* out_scores,
                int64_t* out_ids) {
    if (!X && N > 0) return -1;
    if (!Q || nq < 0 || k <= 0 || d <= 0 || N < 0 || N > 0xFFFFFFFEll) return -1;
    int32_t d8 = (d + 7) & ~7;
#ifdef _OPENMP
    int T = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    int T = 1;
    (void)nthreads;
#endif
    for (int32_t q0 = 0; q0 < nq; q0 += RAGO_QT) {
        int32_t nb = nq - q0 < RAGO_QT ? nq - q0 : RAGO_QT;
        float* Qt = (float*)calloc((size_t)d8 * RAGO_QT, sizeof(float));
        uint64_t* lists = (uint64_t*)calloc((size_t)T * RAGO_QT * k, sizeof(uint64_t));
        int32_t* counts = (int32_t*)calloc((size_t)T * RAGO_QT, sizeof(int32_t));
        float* qn = (float*)calloc(RAGO_QT, sizeof(float));
        if (!Qt || !lists || !counts || !qn) {
            free(Qt); free(lists); free(counts); free(qn);
            return -1;
        }
        for (int32_t b = 0; b < nb; ++b) {
            const float* q = Q + (size_t)(q0 + b) * d;
            for (int32_t j = 0; j < d; ++j) Qt[(size_t)j * RAGO_QT + b] = q[j];
            qn[b] = rago_sqnorm(q, d);
        }
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
        {
#ifdef _OPENMP
            int tid = omp_get_thread_num();
            int nt = omp_get_num_threads();
#else
            int tid = 0, nt = 1;
#endif
            topk_t tk[RAGO_QT];
            for (int b = 0; b < RAGO_QT; ++b) {
                tk[b].keys = lists + ((size_t)tid * RAGO_QT + b) * k;
                tk[b].n = 0;
                tk[b].k = k;
            }
            int64_t lo = N * tid / nt, hi = N * (tid + 1) / nt;
            float sc[RAGO_QT];
            for (int64_t r = lo; r < hi; ++r) {
                const float* x = X + (size_t)r * d;
                row_scores(x, d, d8, Qt, sc);
                float xn = metric == RAGO_METRIC_L2 ? rago_sqnorm(x, d) : 0.0f;
                for (int32_t b = 0; b < nb; ++b) {
                    /* L2: faiss admits dist < FLT_MAX only; with ||q||^2 non-finite every distance
                     * is inf or NaN, so such a query has no results at all. */
                    if (metric == RAGO_METRIC_L2 && !(qn[b] <= FLT_MAX)) continue;
                    float rs = rank_score(sc[b], xn, metric);
                    /* The heap behind IndexFlat.search starts at -FLT_MAX (IP; +FLT_MAX for L2) and
                     * admits a candidate only if it compares strictly better (faiss heap / result
                     * handler semantics): a score that is NaN, -inf or -FLT_MAX never becomes a
                     * result and its slot stays (-1, -FLT_MAX).  Same rule here and in the HIP path. */
                    if (!(rs > -FLT_MAX)) continue;
                    topk_push(&tk[b], make_key(rs, (uint32_t)r));
                }
            }
            for (int b = 0; b < RAGO_QT; ++b) counts[(size_t)tid * RAGO_QT + b] = tk[b].n;
        }
        /* merge the per-thread lists (order-independent: keys are totally ordered) */
        uint64_t* fin = (uint64_t*)calloc((size_t)k, sizeof(uint64_t));
        if (!fin) { free(Qt); free(lists); free(counts); free(qn); return -1; }
        for (int32_t b = 0; b < nb; ++b) {
            topk_t m = {fin, 0, k};
            for (int t = 0; t < T; ++t) {
                const uint64_t* l = lists + ((size_t)t * RAGO_QT + b) * k;
                for (int32_t i = 0; i < counts[(size_t)t * RAGO_QT + b]; ++i) topk_push(&m, l[i]);
            }
            for (int32_t i = 0; i < k; ++i) {
                size_t o = (size_t)(q0 + b) * k + i;
                if (i < m.n) {
                    float s = unord32((uint32_t)(fin[i] >> 32));
                    uint32_t row = 0xFFFFFFFFu - (uint32_t)(fin[i] & 0xFFFFFFFFu);
                    if (metric == RAGO_METRIC_L2) {
                        float dist = qn[b] - s;
                        out_scores[o] = dist < 0.0f ? 0.0f : dist;
                    } else {
                        out_scores[o] = s;
                    }
                    out_ids[o] = (int64_t)row + id_offset;
                } else {
                    out_scores[o] = metric == RAGO_METRIC_L2 ? FLT_MAX : -FLT_MAX;
                    out_ids[o] = -1;
                }
            }
        }
        free(fin); free(Qt); free(lists); free(counts); free(qn);
    }
    return 0;
}

/* Merge n_shards sorted per-shard lists (the step after the all-gather). */
int rago_merge(int32_t metric, int32_t n_shards, int32_t nq, int32_t k, const float* scores,
               const int64_t* ids, float* out_scores, int64_t* out_ids) {
    typedef struct { float s; int64_t id; } ent;
    ent* buf = (ent*)malloc(sizeof(ent) * (size_t)n_shards * k);
    if (!buf) return -1;
    for (int32_t b = 0; b < nq; ++b) {
        int32_t n = 0;
        for (int32_t g = 0; g < n_shards; ++g)
            for (int32_t i = 0; i < k; ++i) {
                size_t o = ((size_t)g * nq + b) * k + i;
                if (ids[o] < 0) continue;
                ent e = {scores[o], ids[o]};
                /* insertion sort: better first; ties by ascending id */
                int32_t j = n++;
                while (j > 0) {
                    ent p = buf[j - 1];
                    int better = metric == RAGO_METRIC_L2
                                     ? (e.s < p.s || (e.s == p.s && e.id < p.id))
                                     : (e.s > p.s || (e.s == p.s && e.id < p.id));
                    if (!better) break;
                    buf[j] = p;
                    --j;
                }
                buf[j] = e;
            }
        for (int32_t i = 0; i < k; ++i) {
            size_t o = (size_t)b * k + i;
            if (i < n) {
                out_scores[o] = buf[i].s;
                out_ids[o] = buf[i].id;
            } else {
                out_scores[o] = metric == RAGO_METRIC_L2 ? FLT_MAX : -FLT_MAX;
                out_ids[o] = -1;
            }
        }
    }
    free(buf);
    return 0;
}

/* ---- synthetic corpus (bench / full-size parity), restated bit-exactly by the HIP generator -- */

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Un-normalised element: sum of four 16-bit uniforms, centred, scaled by 2^-16 (exact in fp32;
 * approximately N(0, 1/3)). */
static inline float synth_elem(uint64_t seed, uint64_t row, uint32_t col, uint32_t d) {
    uint64_t r = mix64(seed ^ mix64(row * (uint64_t)d + col));
    int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) +
                (int32_t)((r >> 32) & 0xFFFF) + (int32_t)((r >> 48) & 0xFFFF);
    return (float)(s - 131070) * (1.0f / 65536.0f);
}

/* 1/sqrt(v) from multiplies and fmas only (bit-trick seed + 4 Newton steps), so that the CPU and
 * the GPU produce the same bits: libm / hardware sqrt and divide are not bit-compatible. */
float rago_rsqrt(float v) {
    if (!(v > 0.0f)) return 0.0f;
    uint32_t u;
    memcpy(&u, &v, 4);
    u = 0x5f3759dfu - (u >> 1);
    float y;
    memcpy(&y, &u, 4);
    const float h = 0.5f * v;
    for (int it = 0; it < 4; ++it) {
        float t = y * y;
        float w = fmaf(-h, t, 1.5f);
        y = y * w;
    }
    return y;
}

/* Rows [row0, row0+n) of the synthetic corpus: elements as above, then each row scaled by
 * rago_rsqrt(sum of squares), the sum taken as a sequential fmaf chain in index order. */
int rago_synth_rows(uint64_t seed, int64_t row0, int64_t n, int32_t d, float* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {
        float* o = out + (size_t)i * d;
        float ss = 0.0f;
        for (int32_t j = 0; j < d; ++j) {
            float v = synth_elem(seed, (uint64_t)(row0 + i), (uint32_t)j, (uint32_t)d);
            o[j] = v;
            ss = fmaf(v, v, ss);
        }
        float inv = rago_rsqrt(ss);
        for (int32_t j = 0; j < d; ++j) o[j] = o[j] * inv;
    }
    return 0;
}

int rago_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
