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
//   * the distance is fma(dp, -2, fl(xx + cc)) written straight into the high word of a 64-bit key {bits(d), j} (equal to the
//     literal three operations of linalg.rs:173-174 while dp + dp cannot overflow: rows with huge norms take the exact path), and
//     the argmin is the LDS unit's: one no-return ds_min_i64 per candidate into the lane's slot (round 4; for d >= 0 the signed
//     64-bit order is (distance, index) lexicographic = first minimum, kmeans.rs:149-156; a negative minimum -- a row that
//     coincides with a centroid up to rounding -- sends the row to the exact path).  Rounds 2-3 scanned in the lane: compare and
//     two selects per candidate, 5 vector instructions per candidate with the distance against 1.5 + one LDS atomic now; the
//     slot of sub-vector m is read back while sub-vector m + 1 is being encoded.
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
    int tiles_per_wave;  // k_encode_small16 only
    int word_stores;     // k_encode_small16 only: codes 4-byte aligned (base and row stride)
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
    constexpr int NPASS = KP / JB;                   // 1, or 2 passes of 32 centroids at KP = 64
    __shared__ __attribute__((aligned(16))) long long slot_s[4][2][NPASS][64];   // [wave][sub-vector parity][pass][lane]
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
    // centroid index inside a pass: the low words of the 64-bit keys (one register each, never rewritten)
    int lo[JB];
#pragma unroll
    for (int r = 0; r < JB; ++r) {
        lo[r] = r;
        asm volatile("" : "+v"(lo[r]));
    }
    int pend_m = -1;                                 // sub-vector whose slot has not been read back yet (wave-uniform)
    float pend_xx = 0.f;
    // code byte of sub-vector pend_m from its slot(s); rows with a huge / non-finite norm or a negative minimum: exact path
    auto finalize = [&](const long long (&k)[NPASS]) {
        long long kb = k[0];
#pragma unroll
        for (int ps = 1; ps < NPASS; ++ps) {
            const long long kk = k[ps] + (long long)(JB * ps);       // index of pass ps: + 32 ps (no carry out of the low word)
            kb = kk < kb ? kk : kb;
        }
        const float best = __int_as_float((int)(kb >> 32));
        const int bidx = (int)(unsigned)kb;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && (!(pend_xx < kBigNorm) || best < 0.f));
        if (valid && !((bal >> lane) & 1ull)) orow[pend_m] = (uint8_t)bidx;
        if (bal) {                                   // NaN / Inf / huge rows, rows on top of a centroid: exact path, 32 rows at a time
            const unsigned l32 = (unsigned)bal, h32 = (unsigned)(bal >> 32);
            if (l32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, pend_m, row0, l32);
            if (h32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, pend_m, row0 + 32, h32);
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
                const int par = m & 1;
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
                        // x_k for both halves straight out of the pair it was read into (op_sel: no broadcast copy)
                        const f32x2 xb = {w[k & ~1], w[(k & ~1) + 1 < DSUB ? (k & ~1) + 1 : k]};
#pragma unroll
                        for (int i16 = 0; i16 < JB / 16; ++i16) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const f32x2 c2 = {cur[i16][2 * i], cur[i16][2 * i + 1]};
                                if (k & 1)
                                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[8 * i16 + i]) : "v"(xb), "s"(c2));
                                else
                                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[8 * i16 + i]) : "v"(xb), "s"(c2));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (k + 1 < DSUB) {
#pragma unroll
                            for (int i16 = 0; i16 < JB / 16; ++i16) cur[i16] = nxt[i16];
                        }
                    }
                    // the previous sub-vector's slot: requested here, consumed behind this pass's atomics (its own atomics were
                    // issued a whole dot-product loop ago)
                    long long kprev[NPASS];
                    const bool fin = jb == 0 && pend_m >= 0;     // wave-uniform
                    if (fin) {
#pragma unroll
                        for (int ps = 0; ps < NPASS; ++ps)
                            kprev[ps] = __hip_atomic_load(&slot_s[wave][pend_m & 1][ps][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    long long* slot = &slot_s[wave][par][jb / JB][lane];
#pragma unroll
                    for (int i16 = 0; i16 < JB / 16; ++i16) {
                        const f32x16 n16 = *(sk_c16ptr)(ccm + jb + 16 * i16);
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const int i = 8 * i16 + i8;
                            const f32x2 c2 = {n16[2 * i8], n16[2 * i8 + 1]};
                            f32x2 t;
                            asm("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(xx2), "s"(c2));            // fl(xx + cc)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                float dd;                                                             // = fl(t - fl(dp + dp)) (no overflow here)
                                asm("v_fma_f32 %0, %1, -2.0, %2" : "=v"(dd) : "v"(acc[i][e]), "v"(t[e]));  // (three-address: lands in the key's high word)
                                const long long key = ((long long)__float_as_int(dd) << 32) | (long long)(unsigned)lo[2 * i + e];
                                if (i == 0 && e == 0) __hip_atomic_store(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                else (void)__hip_atomic_fetch_min(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            }
                        }
                    }
                    if (fin) finalize(kprev);
                }
                pend_m = m;
                pend_xx = xx;
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
    if (pend_m >= 0) {                           // the last sub-vector's slot
        long long kl[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            kl[ps] = __hip_atomic_load(&slot_s[wave][pend_m & 1][ps][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        finalize(kl);
    }
}

}  // namespace pqhip
