// kernels_smallk.hip.h -- PQ encode for small codebooks (K <= 64): HBM-bound, so no matrix core.
//
// With few centroids the distance GEMM is tiny (2 K d flop per vector: 4,096 at the reference's own
// bench shape d = 128, M = 16, K = 16, benches/pq.rs:9-10) while the MFMA kernels pay a fixed
// per-distance epilogue on 32-centroid tiles; measured round 1: 2.1e9 vectors/s at that shape = 15 %
// of HBM and 5 % of the MFMA peak -- on neither roofline.  This kernel streams x once:
//   * one LANE owns one row (64 rows per wave); all M sub-vectors of the row are encoded by that lane,
//     so x is read exactly once, two sub-vectors ahead of its use;
//   * the centroids never touch a vector register or LDS: the transposed image cbt[m][k][KP]
//     (k_build_cbt, kernels_basic.hip.h) is read through the SCALAR cache and a pair of centroids is the scalar operand of
//     one v_pk_fma_f32:  (dp_j, dp_j+1) = fma(x_k, (c_j[k], c_j+1[k]), (dp_j, dp_j+1)),  k ascending
//     from +0 -- rule 2's chain, two centroids per instruction;
//   * the distance is the literal three operations of linalg.rs:173-174 (packed), the argmin a
//     lane-local strict `<` scan in ascending j (first minimum, kmeans.rs:149-156).
// Rows with NaN / Inf / huge norms take encode_rows_slow_v (total order of ordered-float); codebooks
// with non-finite norms never reach this kernel (host dispatch).
#pragma once
#include "kernels_mfma.hip.h"

namespace pqhip {

// The codebook images are read-only for the whole launch: loads through the constant address space stay on
// the scalar path (s_load_dwordx16) even after the kernel's own code stores (plain generic pointers lose that
// once a store may alias them, and the centroids would come through the vector memory pipe instead).
typedef const f32x16 __attribute__((address_space(4)))* sk_c16ptr;

struct SmallKArgs {
    const float* x;      // [n][x_rs]
    int64_t n;
    int64_t x_rs;
    uint8_t* out;        // [n][o_rs]
    int64_t o_rs;
    const float* cbt;    // [M][dsub][KP]  transposed, zero padded
    const float* cc;     // [M][k_pad]     +inf padded
    const float* cb;     // [M][K][dsub]   (exact path)
    int M, K, k_pad;
};

template <int KP, int DSUB>
__global__ __launch_bounds__(256) void k_encode_smallk(SmallKArgs a)
{
    static_assert(KP == 16 || KP == 32 || KP == 64, "padded centroid count");
    constexpr int JB = KP < 32 ? KP : 32;            // centroids per pass (JB / 2 accumulator pairs)
    // x is fetched one sub-vector (DSUB floats per lane) at a time, two sub-vectors ahead of its use (three
    // register sets).  Few registers per wave is the point: the scalar-cache round trips of the centroid
    // blocks (~200 cycles per 8 packed fmas) are hidden by occupancy -- 6 to 8 waves per SIMD -- rather than
    // by a deeper software pipeline, which would need ~100 SGPRs.
    constexpr int G = 1;
    constexpr int CF = G * DSUB;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;   // the wave's 64 rows
    if (row0 >= a.n) return;
    const int64_t row = row0 + lane;
    const bool valid = row < a.n;
    const float* xr = a.x + (valid ? row : a.n - 1) * a.x_rs;   // clamped: loads stay in bounds, result not stored
    uint8_t* orow = a.out + row * a.o_rs;

    auto load_chunk = [&](float (&v)[CF], int m0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float w[DSUB];
            if (m0 + g < a.M) load_row_floats<DSUB, DSUB>(xr + (int64_t)(m0 + g) * DSUB, w);
            else {
#pragma unroll
                for (int e = 0; e < DSUB; ++e) w[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < DSUB; ++e) v[g * DSUB + e] = w[e];
        }
    };
    auto encode_chunk = [&](const float (&v)[CF], int m0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int m = m0 + g;
            if (m < a.M) {                                   // wave-uniform
                float w[DSUB];
#pragma unroll
                for (int e = 0; e < DSUB; ++e) w[e] = v[g * DSUB + e];
                const float xx = norm_unrolled_static<DSUB>(w);
                const f32x2 xx2 = {xx, xx};
                const float* cm = a.cbt + (int64_t)m * DSUB * KP;    // uniform: scalar loads
                const float* ccm = a.cc + (int64_t)m * a.k_pad;
                float best = __builtin_inff();
                int bidx = 0;
#pragma unroll
                for (int jb = 0; jb < KP; jb += JB) {
                    f32x2 acc[JB / 2];
#pragma unroll
                    for (int i = 0; i < JB / 2; ++i) acc[i] = (f32x2){0.f, 0.f};
                    // The centroid block of step k + 1 is requested (64-byte scalar loads, s_load_dwordx16: the scalar
                    // cache is shared by every wave of the CU, so one request per 16 centroids, not one per pair)
                    // before the packed fmas of step k are issued: a scalar-cache round trip hides behind them.
                    f32x16 cur[JB / 16], nxt[JB / 16];
#pragma unroll
                    for (int i16 = 0; i16 < JB / 16; ++i16) cur[i16] = *(sk_c16ptr)(cm + jb + 16 * i16);
#pragma unroll
                    for (int k = 0; k < DSUB; ++k) {
                        if (k + 1 < DSUB) {
#pragma unroll
                            for (int i16 = 0; i16 < JB / 16; ++i16) nxt[i16] = *(sk_c16ptr)(cm + (k + 1) * KP + jb + 16 * i16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const f32x2 xb = {w[k], w[k]};
#pragma unroll
                        for (int i16 = 0; i16 < JB / 16; ++i16) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const f32x2 c2 = {cur[i16][2 * i], cur[i16][2 * i + 1]};
                                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[8 * i16 + i]) : "v"(xb), "s"(c2));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (k + 1 < DSUB) {
#pragma unroll
                            for (int i16 = 0; i16 < JB / 16; ++i16) cur[i16] = nxt[i16];
                        }
                    }
#pragma unroll
                    for (int i16 = 0; i16 < JB / 16; ++i16) {
                        const f32x16 n16 = *(sk_c16ptr)(ccm + jb + 16 * i16);
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const int i = 8 * i16 + i8;
                            const f32x2 c2 = {n16[2 * i8], n16[2 * i8 + 1]};
                            f32x2 t, u;
                            asm("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(xx2), "s"(c2));            // fl(xx + cc)
                            asm("v_pk_add_f32 %0, %1, %1" : "=v"(u) : "v"(acc[i]));                  // fl(dp + dp)
                            const float d0 = fsub(t[0], u[0]), d1 = fsub(t[1], u[1]);                // fl(t - u)
                            const bool l0 = d0 < best;
                            best = l0 ? d0 : best;
                            bidx = l0 ? jb + 2 * i : bidx;
                            const bool l1 = d1 < best;
                            best = l1 ? d1 : best;
                            bidx = l1 ? jb + 2 * i + 1 : bidx;
                        }
                    }
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && !(xx < kBigNorm));
                if (valid && !((bal >> lane) & 1ull)) orow[m] = (uint8_t)bidx;
                if (bal) {                                   // NaN / Inf / huge rows: exact path, 32 rows at a time
                    const unsigned lo = (unsigned)bal, hi = (unsigned)(bal >> 32);
                    if (lo) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, m, row0, lo);
                    if (hi) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, m, row0 + 32, hi);
                }
            }
        }
    };

    float xa[CF], xb_[CF], xc[CF];
    load_chunk(xa, 0);
    if (1 < a.M) load_chunk(xb_, 1);
    for (int m0 = 0; m0 < a.M; m0 += 3) {
        if (m0 + 2 < a.M) load_chunk(xc, m0 + 2);
        encode_chunk(xa, m0);
        if (m0 + 3 < a.M) load_chunk(xa, m0 + 3);
        if (m0 + 1 < a.M) encode_chunk(xb_, m0 + 1);
        if (m0 + 4 < a.M) load_chunk(xb_, m0 + 4);
        if (m0 + 2 < a.M) encode_chunk(xc, m0 + 2);
    }
}

}  // namespace pqhip
