// kernels_adc.hip.h -- "next" row: asymmetric distance computation over a
// resident code matrix.  Lookup tables from the reference's own vector-to-matrix distance
// (linalg.rs:118-148), then a table-sum scan over the u8 codes -- HBM-bound: M bytes in, 4 bytes out
// per code row.  (Non-template kernels: include from exactly one translation unit, pqhip_adc.hip.)
#pragma once
#include "common.hip.h"

namespace pqhip {

// ndarray numeric_util::unrolled_dot(a, b) (rule 1) for two global vectors.
__device__ inline float dot_unrolled_global(const float* __restrict__ a, const float* __restrict__ b, int n)
{
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; n - i >= 8; i += 8) {
#pragma unroll
        for (int l = 0; l < 8; ++l) p[l] = fadd(p[l], fmul(a[i + l], b[i + l]));
    }
    float s = 0.f;
    s = fadd(s, fadd(p[0], p[4]));
    s = fadd(s, fadd(p[1], p[5]));
    s = fadd(s, fadd(p[2], p[6]));
    s = fadd(s, fadd(p[3], p[7]));
    for (; i < n; ++i) s = fadd(s, fmul(a[i], b[i]));
    return s;
}

// y[q][c] = sum_k x[q][k] * P[k][c], the 1-D x 2-D ndarray dot of pq.rs:293 (`x.dot(projection)` for a
// single vector): per output column one sequential  s = s + x[k] * P[k][c]  (separately rounded).
__global__ void k_adc_rotate_queries(const float* __restrict__ x, int64_t x_rs, int nq, const float* __restrict__ P,
                                     int d, float* __restrict__ y)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nq * d) return;
    const int q = (int)(idx / d), c = (int)(idx - (int64_t)q * d);
    const float* xr = x + q * x_rs;
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = fadd(s, fmul(xr[k], P[(int64_t)k * d + c]));
    y[idx] = s;
}

// tables[q][m][j] = fl( fl(yy + cc_j) - fl(dp + dp) ),  yy = y_m . y_m,  dp = c_j . y_m  (unrolled dots):
// `instance.squared_euclidean_distance(centroids)` of linalg.rs:118-148 for sub-vector m of query q.
__global__ void k_adc_tables(const float* __restrict__ y, int64_t y_rs, int nq, const float* __restrict__ cb,
                             const float* __restrict__ cc, int M, int K, int dsub, int k_pad,
                             float* __restrict__ tables)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_q = (int64_t)M * K;
    if (idx >= nq * per_q) return;
    const int q = (int)(idx / per_q);
    const int r = (int)(idx - q * per_q);
    const int m = r / K, j = r - m * K;
    const float* ym = y + q * y_rs + (int64_t)m * dsub;
    const float yy = norm_unrolled_global(ym, dsub);
    const float dp = dot_unrolled_global(cb + ((int64_t)m * K + j) * dsub, ym, dsub);
    tables[idx] = fsub(fadd(yy, cc[(int64_t)m * k_pad + j]), fadd(dp, dp));
}

// ---------------------------------------------------------------------------------------------
// Table-sum scan: out[i] = sum_{m = 0..M-1, in order, from +0} lut[m][codes[i][m]]   (u8 codes)
//
// HBM-bound by design: M bytes in + 4 bytes out per row.  The M x K table (15 KB at M=15, K=256)
// lives in LDS; every lane owns whole rows, so the sum over m is a lane-local sequential chain --
// exactly the declared order -- and there is no cross-lane step.  Code bytes come straight from
// global memory: a lane fetches the aligned dwords that cover its row (rows of consecutive lanes are
// consecutive in memory, so a wave's fetch is one contiguous span), realigns them with v_alignbyte
// and extracts the bytes; no LDS staging of the codes -- the LDS pipe is kept for the gathers, which
// are the binding on-chip resource (random banks: ~3.5 cycles per 32 lookups).
// Rows whose window would leave the code matrix (first / last rows) take byte loads.
// A code >= K raises *err (the lookup would leave its table) and reads entry 0.
// ---------------------------------------------------------------------------------------------
constexpr int kAdcMaxValueWords = 25;   // M <= 100 on the fast path

typedef unsigned u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2_u __attribute__((ext_vector_type(2), aligned(4)));

// NV = ceil(M / 4) dwords of code bytes per row; the window fetched is NV + 1 aligned dwords.
template <int NV>
__global__ __launch_bounds__(256) void k_adc_scan_u8(const uint8_t* __restrict__ codes, int64_t n, int64_t c_rs,
                                                     const float* __restrict__ lut, int M, int K,
                                                     float* __restrict__ out, int64_t rows_per_wg,
                                                     int* __restrict__ err)
{
    constexpr int NW = NV + 1;
    extern __shared__ __attribute__((aligned(16))) float lut_s[];   // [M][K]
    for (int i = threadIdx.x; i < M * K; i += 256) lut_s[i] = lut[i];
    __syncthreads();
    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_wg;
    int64_t row_end = row_begin + rows_per_wg;
    if (row_end > n) row_end = n;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(codes);
    const uintptr_t hi = lo + (uintptr_t)((n - 1) * c_rs + M);      // one past the last code byte
    bool bad = false;
    for (int64_t row = row_begin + threadIdx.x; row < row_end; row += 256) {
        const uintptr_t a = lo + (uintptr_t)(row * c_rs);
        const uintptr_t a0 = a & ~(uintptr_t)3;
        unsigned w[NW];
        if (a0 >= lo && a0 + 4 * NW <= hi) {
            const unsigned* p = reinterpret_cast<const unsigned*>(a0);
            constexpr int N4 = (NW / 4) * 4, N2 = N4 + ((NW - N4) / 2) * 2;
#pragma unroll
            for (int k = 0; k < N4; k += 4) {
                const u32x4_u v = *reinterpret_cast<const u32x4_u*>(p + k);
                w[k] = v[0]; w[k + 1] = v[1]; w[k + 2] = v[2]; w[k + 3] = v[3];
            }
            if (N2 > N4) {
                const u32x2_u v = *reinterpret_cast<const u32x2_u*>(p + N4);
                w[N4] = v[0]; w[N4 + 1] = v[1];
            }
            if (NW > N2) w[N2] = p[N2];
        } else {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                unsigned v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uintptr_t b = a0 + 4 * k + e;
                    if (b >= a && b < a + (uintptr_t)M) v |= (unsigned)*reinterpret_cast<const uint8_t*>(b) << (8 * e);
                }
                w[k] = v;
            }
        }
        const unsigned sh = (unsigned)(a & 3);
        float s = 0.f;
        const float* lm = lut_s;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const unsigned v = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);   // bytes 4k .. 4k+3 of the row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (4 * k + e < M) {
                    unsigned c = (v >> (8 * e)) & 0xffu;
                    if (c >= (unsigned)K) { bad = true; c = 0; }
                    s = fadd(s, lm[c]);
                    lm += K;
                }
            }
        }
        out[row] = s;
    }
    if (bad) atomicOr(err, 1);
}

// ---------------------------------------------------------------------------------------------
// Multi-query scan: one pass over the code matrix serves NQ queries (NQ = 4 or 8):
//     out[q][i] = sum_{m in order, from +0} lut[q][m][codes[i][m]]          q = 0 .. NQ-1
// The single-query kernel above re-reads the whole code matrix per query; here a row's M bytes are fetched once
// and cost M + 4 NQ bytes of HBM traffic for NQ distances.  The NQ tables sit in LDS interleaved by query,
// lut_s[half][m][code][4]: one ds_read_b128 brings a code's entries for four queries (entries 16 bytes apart: 16
// distinct bank groups, like the 4-byte gathers of the single-query form), and the four sums advance with two packed
// adds -- each query's sum is still its own sequential f32 chain over m, the declared order.  One 1,024-thread
// workgroup per CU (the 8-query image is 120 KB of LDS at M = 15, K = 256), contiguous row range per workgroup.
// ---------------------------------------------------------------------------------------------
template <int NV, int NQ>
__global__ __launch_bounds__(1024) void k_adc_scan_u8_mq(const uint8_t* __restrict__ codes, int64_t n, int64_t c_rs,
                                                         const float* __restrict__ lut /* [NQ][M][K] */, int M, int K,
                                                         float* __restrict__ out, int64_t o_rs, int64_t rows_per_wg,
                                                         int* __restrict__ err)
{
    static_assert(NQ == 4 || NQ == 8, "queries per pass");
    constexpr int NW = NV + 1, NH = NQ / 4;
    extern __shared__ __attribute__((aligned(16))) float lutq_s[];   // [NH][M][K][4]
    const int MK = M * K;
    for (int i = threadIdx.x; i < NQ * MK; i += 1024) {
        const int q = i / MK, r = i - q * MK;
        lutq_s[((q >> 2) * MK + r) * 4 + (q & 3)] = lut[i];
    }
    __syncthreads();
    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_wg;
    int64_t row_end = row_begin + rows_per_wg;
    if (row_end > n) row_end = n;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(codes);
    const uintptr_t hi = lo + (uintptr_t)((n - 1) * c_rs + M);      // one past the last code byte
    bool bad = false;
    for (int64_t row = row_begin + threadIdx.x; row < row_end; row += 1024) {
        const uintptr_t a = lo + (uintptr_t)(row * c_rs);
        const uintptr_t a0 = a & ~(uintptr_t)3;
        unsigned w[NW];
        if (a0 >= lo && a0 + 4 * NW <= hi) {
            const unsigned* p = reinterpret_cast<const unsigned*>(a0);
            constexpr int N4 = (NW / 4) * 4, N2 = N4 + ((NW - N4) / 2) * 2;
#pragma unroll
            for (int k = 0; k < N4; k += 4) {
                const u32x4_u v = *reinterpret_cast<const u32x4_u*>(p + k);
                w[k] = v[0]; w[k + 1] = v[1]; w[k + 2] = v[2]; w[k + 3] = v[3];
            }
            if (N2 > N4) {
                const u32x2_u v = *reinterpret_cast<const u32x2_u*>(p + N4);
                w[N4] = v[0]; w[N4 + 1] = v[1];
            }
            if (NW > N2) w[N2] = p[N2];
        } else {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                unsigned v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uintptr_t b = a0 + 4 * k + e;
                    if (b >= a && b < a + (uintptr_t)M) v |= (unsigned)*reinterpret_cast<const uint8_t*>(b) << (8 * e);
                }
                w[k] = v;
            }
        }
        const unsigned sh = (unsigned)(a & 3);
        f32x2 s[NH][2];
#pragma unroll
        for (int hq = 0; hq < NH; ++hq) { s[hq][0] = (f32x2){0.f, 0.f}; s[hq][1] = (f32x2){0.f, 0.f}; }
        const float* lm = lutq_s;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const unsigned v = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);   // bytes 4k .. 4k+3 of the row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (4 * k + e < M) {
                    unsigned c = (v >> (8 * e)) & 0xffu;
                    if (c >= (unsigned)K) { bad = true; c = 0; }
#pragma unroll
                    for (int hq = 0; hq < NH; ++hq) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(lm + (size_t)hq * MK * 4 + 4 * c);
                        s[hq][0] = pk_add(s[hq][0], (f32x2){t[0], t[1]});
                        s[hq][1] = pk_add(s[hq][1], (f32x2){t[2], t[3]});
                    }
                    lm += 4 * K;
                }
            }
        }
#pragma unroll
        for (int hq = 0; hq < NH; ++hq) {
            out[(int64_t)(4 * hq + 0) * o_rs + row] = s[hq][0][0];
            out[(int64_t)(4 * hq + 1) * o_rs + row] = s[hq][0][1];
            out[(int64_t)(4 * hq + 2) * o_rs + row] = s[hq][1][0];
            out[(int64_t)(4 * hq + 3) * o_rs + row] = s[hq][1][1];
        }
    }
    if (bad) atomicOr(err, 1);
}

// 32-bit codes (K > 256) whose M x K table still fits LDS (M K <= 40,960 entries: K = 1,024 at M = 15 takes 60 KB, K = 2,048
// 120 KB): the table in LDS as in the u8 kernel, one 1,024-thread workgroup per CU over a contiguous row range, a lane owns
// whole rows and fetches its 4 M code bytes with 16-byte loads (rows are 4-byte aligned; consecutive lanes read consecutive
// rows, so a wave's loads cover one contiguous span).  Same sum order, same range flag.
__global__ __launch_bounds__(1024) void k_adc_scan_wide(const uint32_t* __restrict__ codes, int64_t n, int64_t c_rs,
                                                        const float* __restrict__ lut, int M, int K,
                                                        float* __restrict__ out, int64_t rows_per_wg, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) float lut_w[];   // [M][K]
    for (int i = threadIdx.x; i < M * K; i += 1024) lut_w[i] = lut[i];
    __syncthreads();
    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_wg;
    int64_t row_end = row_begin + rows_per_wg;
    if (row_end > n) row_end = n;
    bool bad = false;
    for (int64_t row = row_begin + threadIdx.x; row < row_end; row += 1024) {
        const uint32_t* cr = codes + row * c_rs;
        float s = 0.f;
        const float* lm = lut_w;
        int m = 0;
        for (; m + 4 <= M; m += 4) {
            const u32x4_u v = *reinterpret_cast<const u32x4_u*>(cr + m);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned c = v[e];
                if (c >= (unsigned)K) { bad = true; c = 0; }
                s = fadd(s, lm[c]);
                lm += K;
            }
        }
        for (; m < M; ++m) {
            unsigned c = cr[m];
            if (c >= (unsigned)K) { bad = true; c = 0; }
            s = fadd(s, lm[c]);
            lm += K;
        }
        out[row] = s;
    }
    if (bad) atomicOr(err, 1);
}

// any index width / any table size: tables read through L2, one thread per row.  No throughput claim.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_adc_scan_any(const IdxT* __restrict__ codes, int64_t n, int64_t c_rs,
                                                      const float* __restrict__ lut, int M, int K,
                                                      float* __restrict__ out, int* __restrict__ err)
{
    bool bad = false;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
        const IdxT* cr = codes + row * c_rs;
        float s = 0.f;
        for (int m = 0; m < M; ++m) {
            uint64_t c = (uint64_t)cr[m];
            if (c >= (uint64_t)K) { bad = true; c = 0; }
            s = fadd(s, lut[(int64_t)m * K + (int64_t)c]);
        }
        out[row] = s;
    }
    if (bad) atomicOr(err, 1);
}

}  // namespace pqhip
