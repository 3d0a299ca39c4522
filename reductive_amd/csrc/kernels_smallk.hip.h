// kernels_smallk.hip.h -- PQ encode for small codebooks (K <= 64): HBM-bound, so no matrix core.
//
// With few centroids the distance GEMM is tiny (2 K d flop per vector: 4,096 at the reference's own
// bench shape d = 128, M = 16, K = 16, benches/pq.rs:9-10) while the MFMA kernels pay a fixed
// per-distance epilogue on 32-centroid tiles; measured round 1: 2.1e9 vectors/s at that shape = 15 %
// of HBM and 5 % of the MFMA peak -- on neither roofline.  This kernel streams x once:
//   * one LANE owns one row (64 rows per wave); all M sub-vectors of the row are encoded by that lane,
//     so x is read exactly once -- in whole row segments staged through a wave-private LDS slab;
//   * the centroids never touch a vector register or LDS: the transposed image cbt[m][k][KP]
//     (k_build_cbt, kernels_prep.hip.h) is read through the SCALAR cache and a pair of centroids is the scalar operand of
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
    // x reaches the lanes through a wave-private LDS slab: the wave fetches [64 rows][CF floats] chunks with
    // 16-byte loads in which consecutive lanes cover consecutive bytes of a row (full 64..160-byte row
    // segments; one row per lane straight from global memory -- 64 different cache lines per load
    // instruction -- tops out near 3 TB/s whatever the kernel), stores them with a padded row stride and
    // every lane reads back its own row.  A chunk is G whole sub-vectors, about 128 bytes per row; the next
    // chunk travels from HBM into registers while the current one is encoded.
    constexpr int G = (DSUB % 4 == 0) ? ((32 / DSUB) > 0 ? 32 / DSUB : 1) : ((DSUB % 4 == 2) ? ((32 / DSUB) / 2 * 2 > 0 ? (32 / DSUB) / 2 * 2 : 2) : 4);
    constexpr int CF = G * DSUB;                     // floats per row and chunk: a multiple of 4
    static_assert(CF % 4 == 0, "whole 16-byte pieces");
    constexpr int PPR = CF / 4;                      // 16-byte pieces per row
    constexpr int XS = CF + 4;                       // slab row stride in floats (16-byte aligned, spreads the banks)
    __shared__ __attribute__((aligned(16))) float slab_s[4][64 * XS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;   // the wave's 64 rows
    if (row0 >= a.n) return;
    const int64_t row = row0 + lane;
    const bool valid = row < a.n;
    uint8_t* orow = a.out + row * a.o_rs;
    float* slab = slab_s[wave];
    const int nrows = (int)((a.n - row0 < 64) ? a.n - row0 : 64);

    // piece p = lane + 64 i of a chunk: row p / PPR, 16-byte piece p % PPR (compile-time divisor)
    f32x4 st[PPR];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < PPR; ++i) {
            const int p = lane + 64 * i;
            const int r = p / PPR, c = p - r * PPR;
            const int rr = (r < nrows) ? r : nrows - 1;          // clamped: stays in bounds, result not stored
            const int mlast = m0 + (4 * c + 3) / DSUB;           // last sub-vector this piece touches
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (mlast < a.M) v = *reinterpret_cast<const f32x4_u*>(a.x + (row0 + rr) * a.x_rs + (int64_t)m0 * DSUB + 4 * c);
            else if (m0 + (4 * c) / DSUB < a.M) {                // piece straddles the end of the row (CF not dividing d)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m0 + (4 * c + e) / DSUB < a.M) v[e] = a.x[(row0 + rr) * a.x_rs + (int64_t)m0 * DSUB + 4 * c + e];
            }
            st[i] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < PPR; ++i) {
            const int p = lane + 64 * i;
            const int r = p / PPR, c = p - r * PPR;
            *reinterpret_cast<f32x4*>(slab + r * XS + 4 * c) = st[i];
        }
    };
    auto read_row = [&](float (&v)[CF]) {
#pragma unroll
        for (int c = 0; c < PPR; ++c) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(slab + lane * XS + 4 * c);
            v[4 * c] = q[0]; v[4 * c + 1] = q[1]; v[4 * c + 2] = q[2]; v[4 * c + 3] = q[3];
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
                    // before the packed fmas of step k are issued.
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

    float xv[CF];
    fetch(0);
    stash();
    read_row(xv);
    for (int m0 = 0; m0 < a.M; m0 += G) {
        const bool more = m0 + G < a.M;
        if (more) fetch(m0 + G);                 // next chunk: HBM -> registers while this one is encoded
        encode_chunk(xv, m0);
        if (more) {                              // (wave-private slab, LDS is in order per wave: no barrier)
            stash();
            read_row(xv);
        }
    }
}

}  // namespace pqhip
