// kernels_mfma16.hip.h -- PQ encode, fourth generation: the distance chains on v_mfma_f32_16x16x4_f32, four
// waves per SIMD.
//
// Why (round 3, tools/enc_power_ab.sh, profiles/r3_encode_experiments.md): k_encode_mfma_lds3 sits within 1 % of its
// issue model -- 80 matrix instructions + ~290 vector instructions + 128 LDS atomics per 32-row tile, which all share
// one issue port and the vector register ports -- and runs at 2.15-2.25 GHz under the power cap.  Two measured facts
// move that point: (1) the out-of-line exact path costs the loop 45 VGPRs through the call ABI (165 -> 120), i.e. one
// wave per SIMD; (2) with the SAME flop, registers and epilogue, the 16x16x4 form of the matrix instruction holds a
// ~4.5 % higher clock under the cap than the 32x32x2 form (half the accumulator traffic per MAC).
//
// Layout.  v_mfma_f32_16x16x4_f32: lane (i16 = lane & 15, q = lane >> 4) supplies A[i16][k = q] and B[k = q][i16]
// and receives D[4 q + v][i16], v = 0..3; the instruction is the k-ordered fmaf chain k = 0, 1, 2, 3 on top of C
// (pqhip_selftest_mfma_chain checks that on the device), so DP / 4 chained instructions are rule 2's chain.
// A = 16 centroids, B = 16 rows.  One step = 32 rows x 32 centroids = 2 row blocks x 2 centroid blocks x DP / 4
// instructions -- the same 16 accumulator registers, operand registers and LDS fragment bytes as the 32x32x2 kernel --
// but a lane now owns TWO rows (i16 and 16 + i16) with 8 candidates each per step, and a row's candidates sit in four
// lanes.  Consequences:
//   * B operands need no lane exchange: lane (i16, q) loads x[row][4 s + q] directly (10 dwords per tile);
//   * ||x||^2 (rule 1): lane group q holds the elements k = q (mod 4), i.e. ndarray's partial sums p[q] and p[q + 4];
//     u_q = p[q] + p[q + 4] is lane-local and (u0 + u1 + u2 + u3, then the tail elements) needs one all-gather over
//     the four lane groups: v_permlane16_swap + 2 v_permlane32_swap;
//   * keys {bits(d), 16 cb + 4 q + v} go to one LDS slot per (step parity, row block, lane); the slot of step t is
//     folded during step t + 1 -- read, OR 32 t into the index, ds_min into a slot per ROW (the four lane groups of a
//     row meet there, so there is no cross-lane merge at all) -- and the code byte of tile i is written during step 1
//     of tile i + 1;
//   * rows that need the exact path are only RECORDED in the loop (a mask per tile in LDS, a tile bit in a scalar
//     register) and re-evaluated after it, where nothing is live: the loop stays below 128 VGPRs.
// Instantiated for T in {2, 4, 8}, DP in {4, 8, .., 32} with dsub == DP, u8 / u32 codes; everything else stays on
// k_encode_mfma_lds3.  Results are bit-identical to it (tests/test_gpu_parity.py runs every shape through both).
#pragma once
#include "kernels_mfma.hip.h"

namespace pqhip {

// value of lane groups 0..3 (same i16) in every lane
__device__ __forceinline__ void gather_groups(float v, float (&o)[4])
{
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // [0]: even group of the pair, [1]: odd group
    const auto e = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false); // [0]: lower half, [1]: upper half
    const auto d = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);
    o[0] = __uint_as_float(e[0]); o[1] = __uint_as_float(d[0]); o[2] = __uint_as_float(e[1]); o[3] = __uint_as_float(d[1]);
}

template <int T, int DP, typename IdxT>
__global__ __launch_bounds__(256, 4) void k_encode_mfma16(EncodeArgs a)
{
    static_assert(T >= 2 && DP % 4 == 0 && DP >= 4 && DP <= 32, "no such instantiation");
    constexpr int S = DP / 4;                 // matrix instructions per chain
    constexpr int C8 = DP / 8;                // full 8-element chunks of rule 1
    constexpr bool TAIL = (DP % 8) != 0;      // four tail elements (k = 8 C8 .. 8 C8 + 3): the last matrix group
    __shared__ __attribute__((aligned(16))) float afrag_s[T][2][S][64];
    __shared__ __attribute__((aligned(16))) long long slot_s[4][2][2][64];   // [wave][step parity][row block][lane]
    __shared__ __attribute__((aligned(16))) long long fin_s[4][32];          // [wave][row of the tile]
    __shared__ __attribute__((aligned(16))) float cc_s[T * 32];
    __shared__ unsigned need_s[4][kMfma16MaxTiles];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15;
    const int q = lane >> 4;

    // ---- workgroup -> (row group, m); XCD-aware: the M workgroups of one row group share an XCD
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t qq = b >> 3;
    const int64_t g_local = qq / a.M;
    const int m = (int)(qq - g_local * a.M);
    const int64_t group = g_local * 8 + xcd;
    const bool wg_active = (g_local < a.chunks_per_xcd) && (group < a.n_chunks);

    constexpr long long kKeyInit = 0x7fffffffffffffffll;
    if (wg_active) {
        // the 32x32x2 image frags[m][t][s2][lane'] = c[32 t + (lane' & 31)][2 s2 + (lane' >> 5)] re-indexed for 16x16x4:
        // afrag_s[t][cb][s][(i16, q)] = c[32 t + 16 cb + i16][4 s + q]
        const float* fp = a.frags + (int64_t)m * T * (DP / 2) * 64;
        float* dst = &afrag_s[0][0][0][0];
        for (int i = threadIdx.x; i < T * 2 * S * 64; i += 256) {
            const int ln = i & 63;
            int r = i >> 6;
            const int s = r % S;
            r /= S;
            const int cbk = r & 1, t = r >> 1;
            const int li = ln & 15, lq = ln >> 4;
            dst[i] = fp[(t * (DP / 2) + 2 * s + (lq >> 1)) * 64 + 16 * cbk + li + 32 * (lq & 1)];
        }
        const float* ccm = a.cc + (int64_t)m * T * 32;
        for (int i = threadIdx.x; i < T * 32; i += 256) cc_s[i] = ccm[i];
        if (lane < 32) fin_s[wave][lane] = kKeyInit;
    }
    __syncthreads();
    const int64_t row_begin = (group * 4 + wave) * a.rows_per_item;
    if (!wg_active || row_begin >= a.n) return;
    int64_t row_end = row_begin + a.rows_per_item;
    if (row_end > a.n) row_end = a.n;
    const float* xcol = a.x + (int64_t)m * a.dsub + q;
    const bool bad_codebook = a.bad_flag != nullptr && *a.bad_flag != 0;  // wave-uniform

    // x tile: lane (i16, q) holds x[row0 + 16 rb + i16][4 s + q] -- the B operand of (row block rb, k-group s) as loaded.
    // Rows past the end are clamped to the last row (their result is never stored).
    const float* const plast = xcol + (a.n - 1) * a.x_rs;
    const float* prow = xcol + (row_begin + i16) * a.x_rs;
    const int64_t tile_step = 32 * a.x_rs, half_step = 16 * a.x_rs;
    auto load_tile = [&](float (&v)[2][S], int64_t tile_row0) {
        const int left = (int)((a.n - tile_row0 < 32) ? a.n - tile_row0 : 32);  // wave-uniform
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float* p = (16 * rb + i16 < left) ? prow + rb * half_step : plast;
#pragma unroll
            for (int s = 0; s < S; ++s) v[rb][s] = p[4 * s];
        }
    };
    // rule 1 from the distributed elements (see the header comment)
    auto norms = [&](const float (&v)[2][S], float (&xx)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float sq[S];
#pragma unroll
            for (int s = 0; s < S; ++s) sq[s] = fmul(v[rb][s], v[rb][s]);
            float sum = 0.f;
            if constexpr (C8 > 0) {
                float pl = sq[0], ph = sq[1];                     // 0 + x == x exactly for x >= +0 or NaN
#pragma unroll
                for (int c = 1; c < C8; ++c) { pl = fadd(pl, sq[2 * c]); ph = fadd(ph, sq[2 * c + 1]); }
                float u[4];
                gather_groups(fadd(pl, ph), u);
                sum = fadd(fadd(fadd(u[0], u[1]), u[2]), u[3]);
            }
            if constexpr (TAIL) {
                float tl[4];
                gather_groups(sq[2 * C8], tl);
                sum = (C8 > 0) ? fadd(sum, tl[0]) : tl[0];
                sum = fadd(fadd(fadd(sum, tl[1]), tl[2]), tl[3]);
            }
            xx[rb] = sum;
        }
    };

    // index halves of the keys: centroid offset inside a 32-centroid step, 16 cb + 4 q + v (the fold adds 32 t)
    // (one register per key: the low half of a key pair is never rewritten, the fma writes the high half in place)
    int lo[2][2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                lo[rb][cbk][v] = 16 * cbk + 4 * q + v;
                asm volatile("" : "+v"(lo[rb][cbk][v]));
            }

    const int64_t last_tile0 = row_begin + ((row_end - row_begin - 1) / 32) * 32;
    float vn[2][S], bop[2][S], xx[2];
    load_tile(vn, row_begin);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) bop[rb][s] = vn[rb][s];
    norms(bop, xx);
    if (row_begin + 32 <= last_tile0) prow += tile_step;
    load_tile(vn, (row_begin + 32 <= last_tile0) ? row_begin + 32 : last_tile0);

    f32x4 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) acc[rb][cbk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
                acc[rb][cbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag_s[0][cbk][s][lane], bop[rb][s], acc[rb][cbk], 0, 0, 0);
    f32x4 c4[2];
    auto read_cc = [&](int t, f32x4 (&c)[2]) {
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) c[cbk] = *reinterpret_cast<const f32x4*>(&cc_s[32 * t + 16 * cbk + 4 * q]);
    };
    read_cc(0, c4);

    long long* const fin_row = &fin_s[wave][lane & 31];                // lanes 0..31: the tile's rows
    unsigned long long flagged = 0;                                     // wave-uniform: tiles with rows for the exact path
    // code bytes of the tile that started at trow0 (lanes 0..31, one row each); `big`: its rows with a huge / NaN norm
    auto finish_tile = [&](long long kf, int64_t trow0, unsigned big, int tile_idx) {
        const float best = __int_as_float((int)(kf >> 32));
        const int bidx = (int)(unsigned)kf;
        const int64_t row = trow0 + lane;
        const bool valid = lane < 32 && row < a.n;
        const unsigned need = (unsigned)__builtin_amdgcn_ballot_w64(valid && (((big >> (lane & 31)) & 1u) || best < 0.f));
        if (valid && !((need >> (lane & 31)) & 1u)) reinterpret_cast<IdxT*>(a.out)[row * a.o_rs + m] = (IdxT)bidx;
        if (need) {                                                     // wave-uniform
            if (lane == 0) need_s[wave][tile_idx] = need;
            flagged |= 1ull << tile_idx;
        }
    };
    const bool upper16 = (lane & 16) != 0;
    auto big_rows = [&](float x0, float x1) -> unsigned {
        const float xs = upper16 ? x1 : x0;                             // lane L < 32: the norm of row L of the tile
        return (unsigned)__builtin_amdgcn_ballot_w64(lane < 32 && (bad_codebook || !(xs < kBigNorm)));
    };

    unsigned long long st_tiles = 0, st_steps = 0;
    const unsigned long long st_t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned big_cur = big_rows(xx[0], xx[1]), big_prev = 0;
    int tile_idx = 0;
    for (int64_t row0 = row_begin; row0 < row_end; row0 += 32, ++tile_idx) {
        const unsigned long long st_a = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
        const bool first = row0 == row_begin;                           // wave-uniform
        float bop_n[2][S], xx_n[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tp = (t + T - 1) % T;                             // the step whose slots are folded now
            // LDS queue is drained here for free: the previous chains took >= 640 cycles
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            const bool fold = !(first && t == 0);                       // wave-uniform
            long long kp[2] = {kKeyInit, kKeyInit};
            if (fold) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    kp[rb] = __hip_atomic_load(&slot_s[wave][tp & 1][rb][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            long long kf = kKeyInit;
            if (t == 1 && !first) {
                // every step of the previous tile has been folded (the last one during step 0): take the rows' keys
                // and re-arm the row slots before this tile's first fold reaches them (the LDS pipe keeps the order)
                kf = __hip_atomic_load(fin_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (lane < 32) __hip_atomic_store(fin_row, kKeyInit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            if (t == T - 1) {
                // operands of the next x tile are formed only now, when the current ones have been issued for the last
                // time but one; the tile after next starts its trip from HBM right away
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int s = 0; s < S; ++s) bop_n[rb][s] = vn[rb][s];
                norms(bop_n, xx_n);
                if (row0 + 64 <= last_tile0) prow += tile_step;
                load_tile(vn, (row0 + 64 <= last_tile0) ? row0 + 64 : last_tile0);
            }
            float an[2][S];  // A fragments of the NEXT chains: in flight while the VALU works below
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
                for (int s = 0; s < S; ++s) an[cbk][s] = afrag_s[(t + 1) % T][cbk][s][lane];
            __builtin_amdgcn_sched_barrier(0);
            // ---- VALU: 16 distances -> 16 keys ----
            long long key[2][2][4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f32x2 xx2 = {xx[rb], xx[rb]};
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) {
                    const f32x2 c01 = {c4[cbk][0], c4[cbk][1]}, c23 = {c4[cbk][2], c4[cbk][3]};
                    f32x2 t01, t23;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(t01) : "v"(xx2), "v"(c01));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(t23) : "v"(xx2), "v"(c23));
                    const float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float d = ffma(acc[rb][cbk][v], -2.0f, tt[v]);
                        key[rb][cbk][v] = ((long long)__float_as_int(d) << 32) | (long long)(unsigned)lo[rb][cbk][v];
                    }
                    asm volatile("" ::"v"(t01), "v"(t23));
                }
            }
            if (fold && tp > 0) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) kp[rb] |= (long long)(32 * tp);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- next chains + this step's atomics + the fold + next step's norms (queued behind the atomics) ----
            f32x4 nacc[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) nacc[rb][cbk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cbk = 0; cbk < 2; ++cbk)
                        nacc[rb][cbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[cbk][s], (t + 1 < T) ? bop[rb][s] : bop_n[rb][s],
                                                                            nacc[rb][cbk], 0, 0, 0);
#pragma unroll
                for (int r = (16 * s) / S; r < (16 * (s + 1)) / S; ++r) {
                    // a row block's first key of the step is stored (no read-modify-write, and the slot needs no
                    // re-arming), the other seven are min-ed into it
                    const int rb = r >> 3, cbk = (r >> 2) & 1, v = r & 3;
                    long long* slot = &slot_s[wave][t & 1][rb][lane];
                    if ((r & 7) == 0) __hip_atomic_store(slot, key[rb][cbk][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    else (void)__hip_atomic_fetch_min(slot, key[rb][cbk][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (fold) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    (void)__hip_atomic_fetch_min(&fin_s[wave][16 * rb + i16], kp[rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            read_cc((t + 1) % T, c4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) acc[rb][cbk] = nacc[rb][cbk];
            if (t == 1 && !first) finish_tile(kf, row0 - 32, big_prev, tile_idx - 1);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int s = 0; s < S; ++s) bop[rb][s] = bop_n[rb][s];
            xx[rb] = xx_n[rb];
        }
        big_prev = big_cur;
        big_cur = big_rows(xx[0], xx[1]);
        if (a.stamps) { st_tiles += 1; st_steps += __builtin_amdgcn_s_memtime() - st_a; }
    }
    // ---- drain: fold the last step, write the last tile's codes
    {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int tp = T - 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            long long kp = __hip_atomic_load(&slot_s[wave][tp & 1][rb][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            kp |= (long long)(32 * tp);
            (void)__hip_atomic_fetch_min(&fin_s[wave][16 * rb + i16], kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        const long long kf = __hip_atomic_load(fin_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        finish_tile(kf, last_tile0, big_prev, tile_idx - 1);
    }
    // ---- rows for the exact path (a negative or non-finite minimum, huge norms, a bad codebook): nothing is live here
    while (flagged) {                                                   // wave-uniform
        const int ti = __builtin_ctzll(flagged);
        flagged &= flagged - 1;
        const unsigned need = __builtin_amdgcn_readfirstlane(need_s[wave][ti]);
        encode_rows_slow_v<IdxT>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, a.dsub, a.k_pad, 0, m, row_begin + 32 * (int64_t)ti, need);
    }
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 5;
        o[0] = st_tiles; o[1] = st_steps; o[2] = 0;
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

}  // namespace pqhip
