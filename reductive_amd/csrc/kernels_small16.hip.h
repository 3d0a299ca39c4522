// kernels_small16.hip.h -- PQ encode for small codebooks (K <= 32) and sub-vectors of 4 / 8 / 12 / 16 / 20 / 24 / 32 floats on
// v_mfma_f32_16x16x4_f32 (round 4).
//
// Why.  k_encode_smallk (kernels_smallk.hip.h) keeps the centroids on the scalar path: every k-step of a sub-vector waits for one
// s_load_dwordx16 (scalar loads return out of order, so the wait is lgkmcnt(0) and the prefetch distance one step), and with
// 220 VGPRs two waves share a SIMD.  Measured on the reference's bench shape (d = 128, M = 16, K = 16, benches/pq.rs:9-10;
// tools/sk_time.py, 10 M rows): 1.49 ms whether the argmin costs 5 or 1.5 vector instructions per candidate, 1.16 ms with the x
// stream removed, 1.19 ms with 7/8 of the dot products removed, while a bare stream of the same rows takes 0.80 ms in every
// access pattern tried (tools/mb/mb_stream.hip: 6.0-6.4 TB/s from one row per lane to 1 KiB per instruction, at one to eight
// waves per SIMD).  What is left is instruction issue: the SIMD serialises FP32 matrix and vector instructions
// (SQ_VALU_MFMA_BUSY_CYCLES + SQ_ACTIVE_INST_VALU add up to the kernel's duration, tools/sq_counters.sh), so the kernel is
// built to issue as few of either as the arithmetic allows:
//   * a lane fetches 32 bytes of a row per row block and stage -- lane (i16, q): floats [32 st + 8 q, + 8) of row 16 rb + i16,
//     so the four lane groups read one 128-byte line and an instruction touches 16 lines; the next stage's loads (64 rows x 32
//     floats) are in flight while the current one is encoded (two register sets, the loop body is instantiated for both).
//     (That is the geometry up to 8-float sub-vectors; 12- to 20-float ones take 32-row tiles and one whole sub-vector per lane,
//     row block and stage, 24- and 32-float ones 16-row tiles: always whole sub-vectors per lane, 24 to 40 registers per stage);
//   * ||x_m||^2 (rule 1) is lane-local -- the lane holds whole sub-vectors: 9 instructions for 64 (row, sub-vector) pairs --
//     and one gather over the lane groups (v_permlane16_swap + 2 v_permlane32_swap) hands a lane the norms of the stage's
//     sub-vectors for its row;
//   * B operands: the 4 x 4 transpose over the lane groups of k_rotate_pblock9, in place: register r of lane group q of
//     piece j becomes x[row][32 st + 8 r + 4 j + q] -- the B operand of k-group j of the sub-vector that lane group r fetched;
//   * A = 16 centroids: lane (i16 = lane & 15, q = lane >> 4) supplies c[m][16 t + i16][4 s + q] -- the transposed image
//     cbt[m][k][KP] (k_build_cbt) copied verbatim to LDS is exactly that layout, one ds_read_b32 per (t, s), reused by the tile's
//     four row blocks and read one sub-vector ahead;
//   * DSUB / 4 chained instructions are rule 2's fmaf chain (pqhip_selftest_mfma_chain); the lane receives
//     dp[centroid 16 t + 4 q + v][row 16 rb + i16], v = 0..3;
//   * d = fma(dp, -2, fl(xx + cc)) lands in the high word of a key {bits(d), 16 t + 4 q + v}; the lane's 4 T keys go to its
//     own LDS slot (one ds_write_b64, 4 T - 1 ds_min_i64) and one sub-vector later lane L = row L reads the four slots of its
//     row and takes their minimum;
//   * the code bytes of a row are collected in a register and leave as one 32-bit store per four subquantizers;
//   * a row whose minimum is negative, infinite or NaN (it coincides with a centroid up to rounding, or holds NaN / Inf /
//     huge values) gets the code 0xFF -- never a real code for K <= 32 -- and the tile is noted in a scalar mask; after the
//     loop, where nothing is live, the wave re-reads the codes of the noted tiles and sends the marked (row, m) to
//     encode_rows_slow_v.  (The other kernels test the norm against kBigNorm instead: that keeps NaNs out of the keys.  Here a
//     NaN or an overflow anywhere in a row's candidates makes its minimum key negative, NaN or +Inf -- a negative NaN is a
//     negative integer, a positive one the largest -- so testing the minimum alone is enough.)
// Three waves per SIMD (four for 16-float sub-vectors, two for K > 16); a wave runs `tiles_per_wave` consecutive tiles.
// Requires 16-byte aligned rows (host dispatch);
// codes are bit-identical to k_encode_smallk and the oracle (tests/test_gpu_smallk.py).
#pragma once
#include "small16_launch.h"
#include "kernels_mfma16.hip.h"   // gather_groups

namespace pqhip {

// value of lane groups 0..3 (same i16) in every lane, as the pairs (g0, g2) and (g1, g3): operands of v_pk_add_f32 with op_sel
__device__ __forceinline__ void gather_pairs(float v, f32x2& e02, f32x2& o13)
{
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // [0]: even group of the pair, [1]: odd group
    const auto e = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false); // [0]: lower half, [1]: upper half
    const auto d = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);
    e02 = (f32x2){__uint_as_float(e[0]), __uint_as_float(e[1])};
    o13 = (f32x2){__uint_as_float(d[0]), __uint_as_float(d[1])};
}

// FULL: M is a multiple of the sub-vectors per stage (no stage runs past the end of a row).
// Waves per SIMD: three (156 registers at 8 floats; a fourth spills 36 of them: 1.8 instead of 1.2 ms), four for 12-, 16- and 24-float
// sub-vectors (<= 128 registers: d = 768 / M = 48 6.40 -> 6.08 ms per 10 M rows, d = 768 / M = 32 6.70 -> 5.96), three again for 20- and
// 32-float ones (32 floats with four: 7.38 -> 8.24 ms), two for K > 16.
template <int T, int DSUB, bool FULL>
__global__ __launch_bounds__(256, T == 1 ? ((DSUB == 16 || DSUB == 12 || DSUB == 24) ? 4 : 3) : 2) void k_encode_small16(SmallKArgs a)
{
    static_assert((T == 1 || T == 2) && small16_has(16 * T, DSUB), "no such instantiation");
    constexpr int KP = 16 * T;
    constexpr int S = DSUB / 4;               // 16-byte pieces = matrix instructions per chain
    // A lane holds PPL pieces of a row per row block and stage -- whole sub-vectors, so that the norms are lane-local -- and a
    // tile is RBN row blocks: 64 rows x 32 floats per stage up to 8-float sub-vectors, 32 rows x 64 floats for 16-float ones,
    // 16 rows x 128 floats for 32-float ones (the same 32 registers per stage in every case).
    constexpr int PPL = small16_pieces_per_lane(DSUB);
    constexpr int RBN = small16_tile_rows(DSUB) / 16;
    constexpr int TR = 16 * RBN;              // rows per tile
    constexpr int SVL = PPL / S;              // sub-vectors a lane holds per row block and stage
    constexpr int NSV = 4 * SVL;              // sub-vectors per stage (16 PPL floats of a row)
    extern __shared__ __attribute__((aligned(16))) float small16_dyn_s[];
    float* const afrag_s = small16_dyn_s;                                  // [M][DSUB][KP]
    float* const cc_s = small16_dyn_s + (size_t)a.M * DSUB * KP;           // [M][KP]
    // [wave][sub-vector parity][row block][q][i16], 16 idle slots after every row block: a lane's slot is slot `lane` of its row
    // block (an atomic instruction touches 512 contiguous bytes) and the four row blocks' slots of one lane group sit 32 banks
    // apart for the read-back (first layout, [row block][i16][q]: 2-way bank conflicts in every atomic and read -- the LDS pipe was
    // busy 92 % of the kernel, 5/7 of it in conflict cycles, tools/sq_counters.sh)
    constexpr int RBS = 80, PS = RBN * RBS;
    __shared__ __attribute__((aligned(16))) long long slot_s[4][2][PS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15;
    const int q = lane >> 4;

    for (int i = threadIdx.x; i < a.M * DSUB * KP; i += 256) afrag_s[i] = a.cbt[i];
    for (int i = threadIdx.x; i < a.M * KP; i += 256) cc_s[i] = a.cc[(int64_t)(i / KP) * a.k_pad + (i % KP)];
    __syncthreads();

    const int64_t n_tiles = (a.n + TR - 1) / TR;
    const int64_t tile_begin = ((int64_t)blockIdx.x * 4 + wave) * a.tiles_per_wave;
    if (tile_begin >= n_tiles) return;
    const int ntile = (int)((n_tiles - tile_begin < a.tiles_per_wave) ? n_tiles - tile_begin : a.tiles_per_wave);
    const int NP = (a.M * DSUB) / 4;          // 16-byte pieces per row
    const int NST = (NP + 4 * PPL - 1) / (4 * PPL);   // stages per tile

    // ---- x stage: raw[rb][j] = floats [16 PPL st + 4 PPL q + 4 j, + 4) of row 16 rb + i16 of the tile: the four lane groups read
    // whole 128-byte lines of a row with PPL instructions, an instruction touches 16 lines (one row per lane -- 64 lines per
    // instruction -- measured 15 % slower in this kernel).  Rows past the end are clamped to the last row, pieces past the end of
    // the row to its last piece: loaded, never used.
    unsigned voff[RBN];
    auto tile_offsets = [&](int64_t row0) {
        const int left = (int)((a.n - row0 < TR) ? a.n - row0 : TR);        // wave-uniform
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb) {
            const int r = (16 * rb + i16 < left) ? 16 * rb + i16 : left - 1;
            voff[rb] = (unsigned)((int64_t)r * a.x_rs * 4);
        }
    };
    auto issue = [&](f32x4 (&raw)[RBN][PPL], int64_t row0, int st) {
        const char* base = reinterpret_cast<const char*>(a.x + row0 * a.x_rs);
        unsigned pb[PPL];                     // byte offset of the lane's pieces inside the row
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int pc = 4 * PPL * st + PPL * q + j;
            pb[j] = 16u * (unsigned)((pc < NP) ? pc : NP - 1);
        }
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
            for (int j = 0; j < PPL; ++j) raw[rb][j] = *reinterpret_cast<const f32x4*>(base + (voff[rb] + pb[j]));
    };

    // index halves of the keys (one register each, never rewritten)
    int lo[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            lo[t][v] = 16 * t + 4 * q + v;
            asm volatile("" : "+v"(lo[t][v]));
        }
    long long* const my_slot = &slot_s[wave][0][0] + lane;                 // + PS par + RBS rb
    const long long* const row_slots = &slot_s[wave][0][0] + RBS * (q < RBN ? q : 0) + i16;   // lane L = row L = (rb = q, i16): + 16 q' for q' = 0..3

    unsigned long long flagged = 0;           // wave-uniform: tiles with rows for the exact path
    unsigned cw = 0;                          // code bytes of row `lane`, four subquantizers at a time
    // code byte of sub-vector m (already in the slots) for row `lane` of the tile at trow0
    const unsigned lane_o = (unsigned)(lane * a.o_rs);                       // byte offset of row `lane` inside a tile's codes
    auto finalize = [&](int m, int par, int64_t trow0, int ti) {
        const long long* p = row_slots + PS * par;
        const long long k0 = p[0], k1 = p[16], k2 = p[32], k3 = p[48];
        int hf, lf;
        if constexpr (T == 1) {
            // equal distances: the lower lane group holds the lower centroid index, so the high words decide alone
            const int h0 = (int)(k0 >> 32), h1 = (int)(k1 >> 32), h2 = (int)(k2 >> 32), h3 = (int)(k3 >> 32);
            const bool c1 = h1 < h0, c2 = h3 < h2;
            const int ha = c1 ? h1 : h0, la = c1 ? (int)k1 : (int)k0;
            const int hb = c2 ? h3 : h2, lb = c2 ? (int)k3 : (int)k2;
            const bool c3 = hb < ha;
            hf = c3 ? hb : ha;
            lf = c3 ? lb : la;
        } else {
            const long long ka = k1 < k0 ? k1 : k0;
            const long long kb = k3 < k2 ? k3 : k2;
            const long long kf = kb < ka ? kb : ka;
            hf = (int)(kf >> 32);
            lf = (int)kf;
        }
        const float best = __int_as_float(hf);
        const bool odd = !(best >= 0.f && best < __builtin_inff());
        const int code = odd ? 0xff : lf;
        const int left = (int)((a.n - trow0 < TR) ? a.n - trow0 : TR);      // wave-uniform
        const bool valid = lane < left;
        if (__builtin_amdgcn_ballot_w64(odd && valid)) flagged |= 1ull << ti;
        const int sh = 8 * (m & 3);           // wave-uniform
        cw = (sh == 0) ? (unsigned)code : (cw | ((unsigned)code << sh));
        const bool last = m == a.M - 1;
        if (sh == 24 || last) {
            uint8_t* o = a.out + trow0 * a.o_rs + (m & ~3) + lane_o;
            if (valid) {
                if (sh == 24 && a.word_stores) *reinterpret_cast<unsigned*>(o) = cw;
                else {                        // M not a multiple of 4 (the last one to three bytes), or codes that are not 4-byte aligned
                    for (int e = 0; e <= (m & 3); ++e) o[e] = (uint8_t)(cw >> (8 * e));
                }
            }
        }
    };
    auto read_a = [&](int m, float (&af)[T][S], f32x4 (&c4)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) af[t][s] = afrag_s[(m * DSUB + 4 * s + q) * KP + 16 * t + i16];
            c4[t] = *reinterpret_cast<const f32x4*>(&cc_s[m * KP + 16 * t + 4 * q]);
        }
    };

    // one stage: the sub-vectors [st NSV, + NSV) of the tile at row0, out of `cur` (consumed in place).
    // Lane group q holds the sub-vectors q SVL .. q SVL + SVL - 1 of the stage for the rows 16 rb + i16.
    float af[T][S];
    f32x4 c4[T];
    auto stage = [&](f32x4 (&cur)[RBN][PPL], int64_t row0, int st, int ti) {
        // rule 1, lane-local (the lane holds whole sub-vectors), and a gather over the lane groups:
        // xe[rb][l] = (||sv 0 SVL + l||^2, ||sv 2 SVL + l||^2), xo[rb][l] = (sv 1 SVL + l, sv 3 SVL + l) of row 16 rb + i16
        f32x2 xe[RBN][SVL], xo[RBN][SVL];
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
            for (int l = 0; l < SVL; ++l) {
                float xx;
                if constexpr (DSUB >= 8) {
                    // rule 1: p[i] = (((x_i^2 + x_(8+i)^2) + x_(16+i)^2) + ..) over the full chunks of eight, then
                    // s = (((p0 + p4) + (p1 + p5)) + (p2 + p6)) + (p3 + p7), then the (four) tail elements one by one;
                    // pieces 2 c and 2 c + 1 of the lane are chunk c, the tail is the last piece
                    constexpr int NCH = DSUB / 8;
                    f32x2 pp[4];              // (p0, p1), (p2, p3), (p4, p5), (p6, p7)
#pragma unroll
                    for (int pi = 0; pi < 4; ++pi) {
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            const f32x4 v = cur[rb][2 * ch + (pi >> 1)];
                            const f32x2 e = {v[2 * (pi & 1)], v[2 * (pi & 1) + 1]};
                            const f32x2 sq = pk_mul(e, e);
                            pp[pi] = ch == 0 ? sq : pk_add(pp[pi], sq);
                        }
                    }
                    const f32x2 u01 = pk_add(pp[0], pp[2]), u23 = pk_add(pp[1], pp[3]);
                    xx = fadd(fadd(fadd(u01[0], u01[1]), u23[0]), u23[1]);
                    if constexpr (DSUB % 8 != 0) {
                        const f32x4 v = cur[rb][2 * NCH];
                        const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
                        const f32x2 s01 = pk_mul(a01, a01), s23 = pk_mul(a23, a23);
                        xx = fadd(fadd(fadd(fadd(xx, s01[0]), s01[1]), s23[0]), s23[1]);
                    }
                } else {                      // four tail elements: ((x0^2 + x1^2) + x2^2) + x3^2
                    const f32x4 v = cur[rb][l];
                    const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
                    const f32x2 s01 = pk_mul(a01, a01), s23 = pk_mul(a23, a23);
                    xx = fadd(fadd(fadd(s01[0], s01[1]), s23[0]), s23[1]);
                }
                gather_pairs(xx, xe[rb][l], xo[rb][l]);
            }
        // B operands, in place: bo[rb][j][r] = x[row 16 rb + i16][32 st + 8 r + 4 j + q] (piece j of lane group r)
        float bo[RBN][PPL][4];
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const f32x4 v = cur[rb][j];
                const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
                const auto e = __builtin_amdgcn_permlane32_swap(p[0], r[0], false, false);
                const auto o = __builtin_amdgcn_permlane32_swap(p[1], r[1], false, false);
                bo[rb][j][0] = __uint_as_float(e[0]);
                bo[rb][j][1] = __uint_as_float(o[0]);
                bo[rb][j][2] = __uint_as_float(e[1]);
                bo[rb][j][3] = __uint_as_float(o[1]);
            }
#pragma unroll
        for (int c = 0; c < NSV; ++c) {
            const int m = st * NSV + c;
            if (FULL || m < a.M) {            // wave-uniform (false only past the end of the row in its last stage)
                const int parity = c & 1;     // NSV is even: the parity of m
                const int r = c / SVL, l = c % SVL;   // the lane group that fetched the sub-vector, and which of its sub-vectors
                f32x4 acc[RBN][T];
#pragma unroll
                for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[rb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][s], bo[rb][l * S + s][r], acc[rb][t], 0, 0, 0);
                f32x4 cm[T];
#pragma unroll
                for (int t = 0; t < T; ++t) cm[t] = c4[t];
                // the next sub-vector's A fragments and centroid norms (the row's last one re-reads the first)
                read_a((m + 1 < a.M) ? m + 1 : 0, af, c4);
                // the previous sub-vector's code while the chains run
                if (m > 0) finalize(m - 1, parity ^ 1, row0, ti);
#pragma unroll
                for (int rb = 0; rb < RBN; ++rb) {
                    long long* slot = my_slot + PS * parity + RBS * rb;
                    const f32x2 xp = (r & 1) ? xo[rb][l] : xe[rb][l];     // the norm is its low (r < 2) or high half
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const f32x2 c01 = {cm[t][0], cm[t][1]}, c23 = {cm[t][2], cm[t][3]};
                        f32x2 t01, t23;
                        if (r < 2) {
                            asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t01) : "v"(xp), "v"(c01));
                            asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t23) : "v"(xp), "v"(c23));
                        } else {
                            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t01) : "v"(xp), "v"(c01));
                            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t23) : "v"(xp), "v"(c23));
                        }
                        const float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            float dd;
                            asm("v_fma_f32 %0, %1, -2.0, %2" : "=v"(dd) : "v"(acc[rb][t][v]), "v"(tt[v]));
                            const long long key = ((long long)__float_as_int(dd) << 32) | (long long)(unsigned)lo[t][v];
                            if (t == 0 && v == 0) __hip_atomic_store(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            else (void)__hip_atomic_fetch_min(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                }
            }
        }
        if (st == NST - 1) finalize(a.M - 1, (a.M - 1) & 1, row0, ti);     // the tile's last sub-vector
    };

    // ---- the wave's stages, two register sets: while one is encoded the other is on its way from HBM.  After the last stage the
    // same stage is requested again (a load behind a branch makes the compiler wait for it at the join).
    f32x4 raw_a[RBN][PPL], raw_b[RBN][PPL];
    int64_t row0 = tile_begin * TR;
    int tile = 0, st = 0;
    const int total = ntile * NST;
    int64_t n_row0 = row0;
    int n_tile = 0, n_st = 0;
    auto advance = [&](int it) {              // (n_row0, n_tile, n_st) <- the stage after (row0, tile, st); voff follows
        n_row0 = row0; n_tile = tile; n_st = st;
        if (it + 1 < total) {
            if (++n_st == NST) {
                n_st = 0;
                ++n_tile;
                n_row0 += TR;
                tile_offsets(n_row0);
            }
        }
    };
    tile_offsets(row0);
    issue(raw_a, row0, 0);
    read_a(0, af, c4);
    for (int it = 0; it < total; it += 2) {
        advance(it);
        issue(raw_b, n_row0, n_st);
        stage(raw_a, row0, st, tile);
        row0 = n_row0; tile = n_tile; st = n_st;
        advance(it + 1);
        issue(raw_a, n_row0, n_st);
        if (it + 1 < total) stage(raw_b, row0, st, tile);
        row0 = n_row0; tile = n_tile; st = n_st;
    }

    // ---- rows for the exact path: marked 0xFF in the codes of the noted tiles (written by this wave: through L2)
    if (flagged) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
    }
    while (flagged) {                                                       // wave-uniform
        const int ti = __builtin_ctzll(flagged);
        flagged &= flagged - 1;
        const int64_t trow0 = (tile_begin + ti) * TR;
        const bool valid = trow0 + lane < a.n;
        const volatile uint8_t* o = a.out + (trow0 + lane) * a.o_rs;
        for (int m = 0; m < a.M; ++m) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && o[m] == 0xff);
            const unsigned l32 = (unsigned)bal, h32 = (unsigned)(bal >> 32);
            if (l32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, m, trow0, l32);
            if (h32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, m, trow0 + 32, h32);
        }
    }
}

}  // namespace pqhip
