/*
 * pq_oracle.c -- CPU ORACLE for the PQ/OPQ encode-reconstruct hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product path (reductive_amd/ + libpqhip.so) never links,
 * imports or calls anything in oracle/.
 *
 * It restates, in plain C, the algorithm of finalfusion/reductive v0.9.0
 * (all citations relative to the reference tree):
 *   src/pq/pq.rs:268-283       Pq::quantize_batch_into   (optional x.dot(P), then primitives)
 *   src/pq/pq.rs:309-327       Pq::reconstruct_batch_into (gather, optional .dot(P^T))
 *   src/pq/primitives.rs:64-104   quantize_batch_into     (loop over subquantizers)
 *   src/pq/primitives.rs:110-173  reconstruct_into / reconstruct_batch_into
 *   src/kmeans.rs:133-159      cluster_assignments        (distance matrix, first-minimum argmin)
 *   src/kmeans.rs:111-126      cluster_assignment         (single-vector twin)
 *   src/linalg.rs:150-180      SquaredEuclideanDistance Ix2 x Ix2
 *   src/linalg.rs:118-148      SquaredEuclideanDistance Ix1 x Ix2
 *   src/kmeans.rs:166-198, 308-360  update_centroids, kmeans_iteration, mean_squared_error
 *   src/pq/opq.rs:156-195      Opq::train_iteration without its LAPACK calls
 *                              ("next" row: the k-means step of PQ/OPQ training)
 *   pqo_adc_tables / pqo_adc_scan: "next" row rank 4 (asymmetric distance scan over codes), declared
 *                              from linalg.rs:118-148 -- see the comment at the functions
 *
 * PARITY STATUS.  The reference is Rust and cannot be compiled in this image
 * (no rustc/cargo), so this restatement is pinned by the reference's own
 * known-answer tests (tests/golden/reference_kats.json, transcribed DATA from
 * pq.rs:378-407, kmeans.rs:382-395, linalg.rs:291-313).  Those KATs are exact
 * under any summation order.  The floating-point ROUNDING ORDER and the
 * tie-break on real-valued data are fixed by third-party crates that are not
 * in /root/reference (ndarray 0.15 `dot` -> matrixmultiply 0.3 sgemm;
 * ordered-float 2; core::iter::Iterator::min_by_key) -- for those aspects
 * parity is UNPINNED by the reference's tests and this file DECLARES the
 * canonical arithmetic "CANON-F32" below, restating those crates' published
 * algorithms.
 *
 * CANON-F32 (what libpqhip.so must match bit-for-bit for codes):
 *  (1) squared norms  x.x and c.c : ndarray numeric_util::unrolled_dot --
 *      eight partial sums p[l] += x[8t+l]*y[8t+l] (separately rounded multiply
 *      and add, no FMA), then s = 0; s += (p0+p4); s += (p1+p5); s += (p2+p6);
 *      s += (p3+p7); then the <8 tail elements s += x*y in order.
 *  (2) dot products of the distance GEMM and of the rotation GEMM:
 *      matrixmultiply's FMA micro-kernel -- every output element is ONE
 *      sequential fmaf chain over k starting from +0, restarted every
 *      KC = 256 values of k; blocks after the first are added to the running
 *      output with one rounded add:  C = fl(C + chain_b).
 *  (3) distance  D[i][j] = fl( fl(xx_i + cc_j) - fl(dp + dp) )   (linalg.rs:173-174)
 *  (4) assignment = FIRST index of the minimum under ordered-float's total
 *      order (NaN greater than everything, NaN == NaN, -0 == +0)
 *      (kmeans.rs:149-156; Iterator::min_by_key keeps the first of equal minima).
 *  (5) codes: `usize as u8` (primitives.rs:98-100); K <= 256 so no wrap here.
 *  (6) reconstruction: exact copy of codebook rows (primitives.rs:141-147);
 *      OPQ un-rotation r.dot(P^T) by rule (2).
 *  (7) k-means step: centroid sums are sequential f32 adds in row order, counts are f32,
 *      means by IEEE division; the loss is one sequential f32 fold over all elements.
 *      (KATs: kmeans.rs:401-434, 506-520.)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PQO_KC 256 /* matrixmultiply sgemm k-block */

/* ---- (1) ndarray unrolled_dot ------------------------------------------------ */
float pqo_dot_unrolled(const float *x, const float *y, int64_t n)
{
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    while (n - i >= 8) {
        for (int l = 0; l < 8; ++l) {
            float prod = x[i + l] * y[i + l];
            p[l] = p[l] + prod;
        }
        i += 8;
    }
    float sum = 0.0f;
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; i < n; ++i) {
        float prod = x[i] * y[i];
        sum = sum + prod;
    }
    return sum;
}

/* ---- (4) ordered-float comparison ---------------------------------------------- */
static inline int of_less(float a, float b)
{
    if (isnan(a)) return 0;
    if (isnan(b)) return 1;
    return a < b;
}

int64_t pqo_first_min(const float *d, int64_t n)
{
    int64_t best = 0;
    for (int64_t j = 1; j < n; ++j)
        if (of_less(d[j], d[best])) best = j;
    return best;
}

/* ---- linalg.rs:150-180, literal (small inputs; used by the KAT tests) ---------- */
/* x: [n, dd] row stride x_rs (unit column stride); c: [k, dd] contiguous; out [n,k] */
void pqo_sqdist_mm(const float *x, int64_t n, int64_t x_rs, const float *c, int64_t k,
                   int64_t dd, float *out)
{
    float *cc = (float *)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    for (int64_t j = 0; j < k; ++j) cc[j] = pqo_dot_unrolled(c + j * dd, c + j * dd, dd);
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = x + i * x_rs;
        float xx = pqo_dot_unrolled(xi, xi, dd);
        for (int64_t j = 0; j < k; ++j) {
            float total = 0.0f;
            for (int64_t kb = 0; kb < dd; kb += PQO_KC) {
                int64_t ke = kb + PQO_KC < dd ? kb + PQO_KC : dd;
                float ab = 0.0f;
                for (int64_t q = kb; q < ke; ++q) ab = __builtin_fmaf(xi[q], c[j * dd + q], ab);
                total = (kb == 0) ? ab : total + ab;
            }
            float t = xx + cc[j];
            float u = total + total;
            out[i * k + j] = t - u;
        }
    }
    free(cc);
}

/* kmeans.rs:133-159 (Axis(0) instances) */
void pqo_cluster_assignments(const float *centroids, int64_t k, int64_t dd, const float *x,
                             int64_t n, int64_t x_rs, int64_t *out)
{
    float *d = (float *)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    for (int64_t i = 0; i < n; ++i) {
        pqo_sqdist_mm(x + i * x_rs, 1, x_rs, centroids, k, dd, d);
        out[i] = pqo_first_min(d, k);
    }
    free(d);
}

/* ---- rotation  rx = x.dot(P)   (pq.rs:276) -------------------------------------- */
static void rotate_row(const float *x, const float *P, int64_t d, float *out, float *ab);

/* x [n,d] strided (elements), out [n,d] contiguous */
void pqo_rotate(const float *x, int64_t n, int64_t d, int64_t x_rs, int64_t x_cs, const float *P,
                float *out)
{
    float *row = (float *)malloc(sizeof(float) * (size_t)d * 2);
    float *ab = row + d;
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = 0; k < d; ++k) row[k] = x[i * x_rs + k * x_cs];
        rotate_row(row, P, d, out + i * d, ab);
    }
    free(row);
}

/* ---- encode ---------------------------------------------------------------------- */
typedef struct {
    int64_t M, K, dsub;
    float *ct; /* [M][dsub][K]  transposed centroids  */
    float *cc; /* [M][K]        squared norms (linalg.rs:168) */
} enc_tables;

static void build_tables(enc_tables *t, const float *cb, int64_t M, int64_t K, int64_t dsub)
{
    t->M = M; t->K = K; t->dsub = dsub;
    t->ct = (float *)malloc(sizeof(float) * (size_t)(M * dsub * K));
    t->cc = (float *)malloc(sizeof(float) * (size_t)(M * K));
    for (int64_t m = 0; m < M; ++m)
        for (int64_t j = 0; j < K; ++j) {
            const float *c = cb + (m * K + j) * dsub;
            t->cc[m * K + j] = pqo_dot_unrolled(c, c, dsub);
            for (int64_t k = 0; k < dsub; ++k) t->ct[(m * dsub + k) * K + j] = c[k];
        }
}

static void free_tables(enc_tables *t) { free(t->ct); free(t->cc); }

#define HOT_CAT2(a, b) a##b
#define HOT_CAT(a, b) HOT_CAT2(a, b)
#define HOT_NAME(f) HOT_CAT(f, _base)
#define HOT_ATTR
#include "pq_oracle_hot.h"
#undef HOT_NAME
#undef HOT_ATTR
#if defined(__x86_64__)
#define HOT_NAME(f) HOT_CAT(f, _avx2fma)
#define HOT_ATTR __attribute__((target("avx2,fma")))
#include "pq_oracle_hot.h"
#undef HOT_NAME
#undef HOT_ATTR
static int have_avx2fma(void)
{
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}
#else
static int have_avx2fma(void) { return 0; }
#define rotate_row_avx2fma rotate_row_base
#define encode_row_avx2fma encode_row_base
#endif

static void rotate_row(const float *x, const float *P, int64_t d, float *out, float *ab)
{
    if (have_avx2fma()) rotate_row_avx2fma(x, P, d, out, ab);
    else rotate_row_base(x, P, d, out, ab);
}

static void encode_row(const enc_tables *t, const float *row, float *acc, int64_t *codes)
{
    if (have_avx2fma()) encode_row_avx2fma(t, row, acc, codes);
    else encode_row_base(t, row, acc, codes);
}

/* 1 when the AVX2+FMA build of the hot loops is in use (reported by the bench) */
int pqo_uses_fma_simd(void) { return have_avx2fma(); }

typedef struct {
    const enc_tables *t;
    const float *P; /* or NULL */
    const float *x;
    int64_t x_rs, x_cs;
    int64_t row_begin, row_end;
    void *out;
    int out_width; /* bytes: 1, 2, 4, 8 */
    int64_t o_rs, o_cs;
} enc_job;

static void *encode_range(void *arg)
{
    enc_job *j = (enc_job *)arg;
    const enc_tables *t = j->t;
    const int64_t d = t->M * t->dsub;
    float *buf = (float *)malloc(sizeof(float) * (size_t)(3 * d + 2 * t->K));
    float *row = buf, *rrow = buf + d, *ab = buf + 2 * d, *acc = buf + 3 * d;
    int64_t *codes = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->M);
    for (int64_t i = j->row_begin; i < j->row_end; ++i) {
        for (int64_t k = 0; k < d; ++k) row[k] = j->x[i * j->x_rs + k * j->x_cs];
        const float *src = row;
        if (j->P) { rotate_row(row, j->P, d, rrow, ab); src = rrow; }
        encode_row(t, src, acc, codes);
        for (int64_t m = 0; m < t->M; ++m) {
            int64_t off = i * j->o_rs + m * j->o_cs;
            switch (j->out_width) {
            case 1: ((uint8_t *)j->out)[off] = (uint8_t)codes[m]; break;
            case 2: ((uint16_t *)j->out)[off] = (uint16_t)codes[m]; break;
            case 4: ((uint32_t *)j->out)[off] = (uint32_t)codes[m]; break;
            default: ((uint64_t *)j->out)[off] = (uint64_t)codes[m]; break;
            }
        }
    }
    free(codes);
    free(buf);
    return NULL;
}

/* Pq::quantize_batch_into.  cb: [M][K][dsub] C-order; P: [d][d] row-major or NULL;
 * x: [n,d] with element strides; out: [n,M] of out_width-byte unsigned ints with element strides.
 * n_threads <= 1 is the reference's own threading (its encode loop is single-threaded,
 * primitives.rs:89-103); n_threads > 1 shards rows over pthreads (same arithmetic per row). */
int pqo_quantize_batch(const float *cb, int64_t M, int64_t K, int64_t dsub, const float *P,
                       const float *x, int64_t n, int64_t x_rs, int64_t x_cs, void *out,
                       int out_width, int64_t o_rs, int64_t o_cs, int n_threads)
{
    if (M <= 0 || K <= 0 || dsub <= 0 || n < 0) return 1;
    if (out_width != 1 && out_width != 2 && out_width != 4 && out_width != 8) return 1;
    enc_tables t;
    build_tables(&t, cb, M, K, dsub);
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n) n_threads = n > 0 ? (int)n : 1;
    enc_job *jobs = (enc_job *)malloc(sizeof(enc_job) * (size_t)n_threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    int64_t per = (n + n_threads - 1) / n_threads;
    for (int i = 0; i < n_threads; ++i) {
        int64_t b = i * per, e = b + per;
        if (b > n) b = n;
        if (e > n) e = n;
        jobs[i] = (enc_job){&t, P, x, x_rs, x_cs, b, e, out, out_width, o_rs, o_cs};
    }
    if (n_threads == 1) {
        encode_range(&jobs[0]);
    } else {
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, encode_range, &jobs[i]);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    }
    free(th);
    free(jobs);
    free_tables(&t);
    return 0;
}

/* ---- reconstruct ------------------------------------------------------------------ */
/* Pq::reconstruct_batch_into.  codes [n,M] (code_width-byte unsigned, element strides);
 * out [n,d] f32 with element strides.  Returns 2 if a code >= K (reference: index_axis panic,
 * primitives.rs:146). */
int pqo_reconstruct_batch(const float *cb, int64_t M, int64_t K, int64_t dsub, const float *P,
                          const void *codes, int code_width, int64_t n, int64_t c_rs,
                          int64_t c_cs, float *out, int64_t o_rs, int64_t o_cs)
{
    const int64_t d = M * dsub;
    float *row = (float *)malloc(sizeof(float) * (size_t)d * 3);
    float *rrow = row + d, *ab = row + 2 * d;
    float *PT = NULL;
    if (P) { /* r.dot(P^T): out[k] = sum_c r[c] * P[k][c]  == rotate_row with P^T */
        PT = (float *)malloc(sizeof(float) * (size_t)(d * d));
        for (int64_t k = 0; k < d; ++k)
            for (int64_t c = 0; c < d; ++c) PT[c * d + k] = P[k * d + c];
    }
    int rc = 0;
    for (int64_t i = 0; i < n && !rc; ++i) {
        for (int64_t m = 0; m < M; ++m) {
            int64_t off = i * c_rs + m * c_cs;
            uint64_t code;
            switch (code_width) {
            case 1: code = ((const uint8_t *)codes)[off]; break;
            case 2: code = ((const uint16_t *)codes)[off]; break;
            case 4: code = ((const uint32_t *)codes)[off]; break;
            default: code = ((const uint64_t *)codes)[off]; break;
            }
            if (code >= (uint64_t)K) { rc = 2; break; }
            memcpy(row + m * dsub, cb + (m * K + (int64_t)code) * dsub, sizeof(float) * (size_t)dsub);
        }
        if (rc) break;
        const float *src = row;
        if (PT) { rotate_row(row, PT, d, rrow, ab); src = rrow; }
        for (int64_t k = 0; k < d; ++k) out[i * o_rs + k * o_cs] = src[k];
    }
    free(PT);
    free(row);
    return rc;
}

/* ---- single-vector twins (kept on the CPU by the product too) ---------------------- */
/* linalg.rs:118-148: dp_j = centroid_j.dot(x) is a 1-D x 1-D ndarray dot = unrolled_dot. */
int pqo_quantize_vector(const float *cb, int64_t M, int64_t K, int64_t dsub, const float *P,
                        const float *x, int64_t *codes)
{
    const int64_t d = M * dsub;
    float *rx = (float *)malloc(sizeof(float) * (size_t)(d + K));
    float *dist = rx + d;
    if (P) {
        /* x.dot(P) for a 1-D x: ndarray's non-BLAS gemv is, per output c, a dot of x with the
         * strided column P[:,c] -> the plain sequential loop  s = s + a*b. */
        for (int64_t c = 0; c < d; ++c) {
            float s = 0.0f;
            for (int64_t k = 0; k < d; ++k) { float prod = x[k] * P[k * d + c]; s = s + prod; }
            rx[c] = s;
        }
    } else {
        memcpy(rx, x, sizeof(float) * (size_t)d);
    }
    for (int64_t m = 0; m < M; ++m) {
        const float *xs = rx + m * dsub;
        float xx = pqo_dot_unrolled(xs, xs, dsub);
        for (int64_t j = 0; j < K; ++j) {
            const float *c = cb + (m * K + j) * dsub;
            float cc = pqo_dot_unrolled(c, c, dsub);
            float dp = pqo_dot_unrolled(c, xs, dsub);
            float t = xx + cc;
            float u = dp + dp;
            dist[j] = t - u;
        }
        codes[m] = pqo_first_min(dist, K);
    }
    free(rx);
    return 0;
}

/* ---- "next" row: the k-means step of training -------------- */
/* kmeans.rs:166-198 update_centroids.  centroids [K][dim] are zero-filled, every instance is
 * added to its centroid IN ROW ORDER (`centroid += &instance`, one rounded f32 add per element),
 * counts are f32 incremented by 1.0 (so they stick at 2^24), and non-empty centroids are divided
 * by their count (`centroid /= &count`, IEEE division).  Empty clusters stay at zero. */
void pqo_update_centroids(float *centroids, int64_t K, int64_t dim, const float *x, int64_t n,
                          int64_t x_rs, int64_t x_cs, const int64_t *assign)
{
    float *counts = (float *)calloc((size_t)K, sizeof(float));
    for (int64_t i = 0; i < K * dim; ++i) centroids[i] = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        float *c = centroids + assign[i] * dim;
        for (int64_t e = 0; e < dim; ++e) c[e] = c[e] + x[i * x_rs + e * x_cs];
        counts[assign[i]] = counts[assign[i]] + 1.0f;
    }
    for (int64_t k = 0; k < K; ++k)
        if (counts[k] > 0.0f)
            for (int64_t e = 0; e < dim; ++e) centroids[k * dim + e] = centroids[k * dim + e] / counts[k];
    free(counts);
}

/* kmeans.rs:329-360 mean_squared_error: errors = centroids.select(assignments) - instances;
 * sse = errors.into_iter().map(|v| v * v).sum()  -- ONE sequential f32 fold over all n*dim
 * elements in row-major order; result sse / (n*dim as f32). */
float pqo_mean_squared_error(const float *centroids, int64_t dim, const float *x, int64_t n,
                             int64_t x_rs, int64_t x_cs, const int64_t *assign)
{
    float sse = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        const float *c = centroids + assign[i] * dim;
        for (int64_t e = 0; e < dim; ++e) {
            float err = c[e] - x[i * x_rs + e * x_cs];
            float sq = err * err;
            sse = sse + sq;
        }
    }
    return sse / (float)(uint64_t)(n * dim);
}

/* kmeans.rs:308-327 kmeans_iteration (assignments by rule (1)-(4), update_centroids,
 * mean_squared_error) run `n_iterations` times on every subquantizer's column block, i.e.
 * kmeans_with_centroids(NIterationsCondition(n)) of pq.rs:176 for all m at once; n_iterations = 1
 * is opq.rs:227-245 update_subquantizers.  cb [M][K][dsub] is updated in place; loss [M] (may be
 * NULL) receives the last iteration's mean squared error of every subquantizer. */
int pqo_kmeans_iterations(float *cb, int64_t M, int64_t K, int64_t dsub, const float *x, int64_t n,
                          int64_t x_rs, int64_t x_cs, int n_iterations, float *loss, int n_threads)
{
    int64_t *codes = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n * M));
    int64_t *col = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int it = 0; it < n_iterations; ++it) {
        int rc = pqo_quantize_batch(cb, M, K, dsub, NULL, x, n, x_rs, x_cs, codes, 8, M, 1, n_threads);
        if (rc) { free(codes); free(col); return rc; }
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t i = 0; i < n; ++i) col[i] = codes[i * M + m];
            const float *xm = x + m * dsub * x_cs;
            pqo_update_centroids(cb + m * K * dsub, K, dsub, xm, n, x_rs, x_cs, col);
            if (loss && it == n_iterations - 1)
                loss[m] = pqo_mean_squared_error(cb + m * K * dsub, dsub, xm, n, x_rs, x_cs, col);
        }
    }
    free(codes);
    free(col);
    return 0;
}

/* ---- OPQ training iteration (opq.rs:156-195), everything except LAPACK ------------------------ */
/* C = A^T . B for A [n][da], B [n][db] (`instances.t().dot(&reconstructed)`, opq.rs:191): an
 * ndarray 2-D dot = matrixmultiply sgemm, rule (2) with k running over the n ROWS: every output
 * element is a sequential fmaf chain over the rows of a 256-row block, and the block results are
 * added to C in block order with one rounded add each (the first block initialises C). */
typedef struct { const float *a, *b; float *c; int64_t n, da, db, a_rs, b_rs, i0, i1; } xtb_job;

static void *xtb_range(void *arg)
{
    xtb_job *j = (xtb_job *)arg;
    float *ab = (float *)malloc(sizeof(float) * (size_t)j->db);
    for (int64_t i = j->i0; i < j->i1; ++i) {
        float *ci = j->c + i * j->db;
        for (int64_t rb = 0; rb < j->n; rb += PQO_KC) {
            const int64_t re = rb + PQO_KC < j->n ? rb + PQO_KC : j->n;
            for (int64_t c = 0; c < j->db; ++c) ab[c] = 0.0f;
            for (int64_t r = rb; r < re; ++r) {
                const float av = j->a[r * j->a_rs + i];
                const float *br = j->b + r * j->b_rs;
                for (int64_t c = 0; c < j->db; ++c) ab[c] = fmaf(av, br[c], ab[c]);
            }
            if (rb == 0) for (int64_t c = 0; c < j->db; ++c) ci[c] = ab[c];
            else for (int64_t c = 0; c < j->db; ++c) ci[c] = ci[c] + ab[c];
        }
        if (j->n == 0) for (int64_t c = 0; c < j->db; ++c) ci[c] = 0.0f;
    }
    free(ab);
    return NULL;
}

void pqo_at_dot_b(const float *a, int64_t n, int64_t da, int64_t a_rs, const float *b, int64_t db,
                  int64_t b_rs, float *c, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > da) n_threads = (int)(da > 0 ? da : 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    xtb_job *jobs = (xtb_job *)malloc(sizeof(xtb_job) * (size_t)n_threads);
    const int64_t per = (da + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        xtb_job jb = {a, b, c, n, da, db, a_rs, b_rs, t * per, (t + 1) * per < da ? (t + 1) * per : da};
        if (jb.i0 > da) jb.i0 = da;
        jobs[t] = jb;
        pthread_create(&th[t], NULL, xtb_range, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* The device part of Opq::train_iteration (opq.rs:156-195): rx = instances.dot(projection) (:167),
 * update_subquantizers (:168 -> :227-245, one kmeans_iteration per subquantizer), the
 * quantize -> reconstruct round trip on rx with the NEW centroids (:176-182, primitives without
 * projection), and cross = instances.t().dot(&reconstructed) (:191).  The SVD of `cross` and
 * projection = u.dot(vt) (:191-192) are LAPACK's and stay with the caller.
 * cb [M][K][dsub] is updated in place; cross [d][d] is written. */
int pqo_opq_train_step(float *cb, int64_t M, int64_t K, int64_t dsub, const float *P, const float *x,
                       int64_t n, int64_t x_rs, int64_t x_cs, float *cross, int n_threads)
{
    const int64_t d = M * dsub;
    float *rx = (float *)malloc(sizeof(float) * (size_t)(n * d + 1));
    float *xc = (float *)malloc(sizeof(float) * (size_t)(n * d + 1));
    int64_t *codes = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n * M + 1));
    pqo_rotate(x, n, d, x_rs, x_cs, P, rx);
    int rc = pqo_kmeans_iterations(cb, M, K, dsub, rx, n, d, 1, 1, NULL, n_threads);
    if (!rc) rc = pqo_quantize_batch(cb, M, K, dsub, NULL, rx, n, d, 1, codes, 8, M, 1, n_threads);
    if (!rc) rc = pqo_reconstruct_batch(cb, M, K, dsub, NULL, codes, 8, n, M, 1, rx, d, 1);
    if (!rc) {
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = 0; k < d; ++k) xc[i * d + k] = x[i * x_rs + k * x_cs];
        pqo_at_dot_b(xc, n, d, d, rx, d, d, cross, n_threads);
    }
    free(rx); free(xc); free(codes);
    return rc;
}

/* ---- "next" row: asymmetric distance computation over codes --------
 * Not a function of reductive itself (it is the step that follows encode in any PQ pipeline); it is
 * DECLARED here from the reference's own pieces so that it has an exact definition:
 *   tables[m][j] = y_m.squared_euclidean_distance(quantizers[m])[j]       (linalg.rs:118-148, the very
 *                  distances kmeans::cluster_assignment minimises at kmeans.rs:111-126), with
 *                  y = query.dot(P) first when the quantizer has a projection (pq.rs:293);
 *                  => argmin_j tables[m][j] == quantize_vector(query)[m]   (tested);
 *   dist[i]      = sum over m = 0..M-1, in order, from +0, of tables[m][codes[i][m]]  (f32 adds,
 *                  Rust's `Iterator::sum::<f32>()` order).
 * Parity: unpinned by the reference by construction (no reference lines); the oracle is the definition. */
int pqo_adc_tables(const float *cb, int64_t M, int64_t K, int64_t dsub, const float *P, const float *y,
                   float *tables)
{
    const int64_t d = M * dsub;
    float *rx = (float *)malloc(sizeof(float) * (size_t)d);
    if (P) {
        for (int64_t c = 0; c < d; ++c) {       /* 1-D x 2-D ndarray dot, as in pqo_quantize_vector */
            float s = 0.0f;
            for (int64_t k = 0; k < d; ++k) { float prod = y[k] * P[k * d + c]; s = s + prod; }
            rx[c] = s;
        }
    } else {
        memcpy(rx, y, sizeof(float) * (size_t)d);
    }
    for (int64_t m = 0; m < M; ++m) {
        const float *xs = rx + m * dsub;
        float xx = pqo_dot_unrolled(xs, xs, dsub);
        for (int64_t j = 0; j < K; ++j) {
            const float *c = cb + (m * K + j) * dsub;
            float cc = pqo_dot_unrolled(c, c, dsub);
            float dp = pqo_dot_unrolled(c, xs, dsub);
            float t = xx + cc;
            float u = dp + dp;
            tables[m * K + j] = t - u;
        }
    }
    free(rx);
    return 0;
}

/* returns 2 when a code >= K is met (the lookup would index out of bounds) */
int pqo_adc_scan(const float *tables, int64_t M, int64_t K, const void *codes, int code_bytes, int64_t n,
                 int64_t c_rs, int64_t c_cs, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.0f;
        for (int64_t m = 0; m < M; ++m) {
            const int64_t off = i * c_rs + m * c_cs;
            uint64_t code;
            switch (code_bytes) {
            case 1: code = ((const uint8_t *)codes)[off]; break;
            case 2: code = ((const uint16_t *)codes)[off]; break;
            case 4: code = ((const uint32_t *)codes)[off]; break;
            default: code = ((const uint64_t *)codes)[off]; break;
            }
            if (code >= (uint64_t)K) return 2;
            s = s + tables[m * K + (int64_t)code];
        }
        out[i] = s;
    }
    return 0;
}

int pqo_abi_version(void) { return 4; }
