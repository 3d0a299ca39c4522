// kernels_opq_fused2.hip.h -- OPQ encode in ONE kernel, second generation: rx = x.dot(P) (pq.rs:276) never leaves the
// register file and BOTH stationary operands live in LDS.  (Template kernels, instantiated from opq_fused2_launch.hip.)
//
// The first generation (round 2, removed from the library in round 4) staged x through LDS slabs, which left no room for the codebook
// fragments next to the P block: they came from L2 and cost 10 % (36.5 ms against 34.5 for the two-kernel path).  Round 3's
// rotation kernel (kernels_rotate8.hip.h) takes x straight from global memory into the MFMA operand registers, so LDS now
// holds the 64-slot P block (76.8 KB at d = 300), the fragments of the block's NM = 64 / dsub sub-codebooks (61.4 KB at
// dsub = 20, K = 256), their norms and one argmin slot per lane: 146 KB.  The kernel is k_rotate_pblock8's burst loop --
// P fragment as A operand, x as B operand, so the finished 32 x 64 tile has the ROWS on the lanes and, with P's columns
// staged in the slot order i = (r & 3) + 8 (r >> 2) + 4 h <-> local column 2 r + h, register r of lane (row j, half h) is
// rx[j][2 (16 t + r) + h]: exactly the B operand of k-step 16 t + r of the distance chains -- followed by the LDS-atomic
// encode epilogue of the first generation on the NM sub-vectors held in the accumulators.  Eight waves per workgroup (two per
// SIMD, <= 256 VGPRs): the rotation half keeps the matrix pipe busy with two waves (the GATHER form of k_rotate_pblock8 shows it),
// and both accumulator pairs, two burst buffers, the keys and two fragment sets are live around the seam between the halves.
// Arithmetic is CANON-F32 throughout (see kernels_opq_common.hip.h); rows that need the exact path are re-rotated by a scalar
// rule-2 chain (opq_rows_slow).  Instantiated for the burst structures of the shapes it is dispatched for
// (opq_fused2_launch.h); everything else keeps the two-kernel path.
#pragma once
#include "kernels_opq_common.hip.h"
#include "kernels_rotate8.hip.h"

namespace pqhip {

// The 32-slot form of rot8_burst (NS = 32: one column tile per block, for dimensions whose 64-column P block does not fit
// LDS): the LDS operands of group g + 1 are requested before the two matrix instructions of group g issue.
__device__ __forceinline__ void rot32_read(f32x2& o, const float* plane_g) { o = *reinterpret_cast<const f32x2*>(plane_g); }
template <bool FULL>
__device__ __forceinline__ void rot32_burst(const float* plane_b, const float* plane_next, int ng, const float (&xo)[16], f32x16& c0, f32x2& cur)
{
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (!FULL && g >= ng) break;
        f32x2 nxt = cur;
        const bool last = FULL ? (g == 7) : (g + 1 == ng);
        rot32_read(nxt, last ? plane_next : plane_b + (g + 1) * 128);
        __builtin_amdgcn_sched_barrier(0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0], xo[2 * g], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1], xo[2 * g + 1], c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// NS = slots (columns) of the P block a workgroup keeps in LDS: 64 (two column tiles; d <= ~420 next to three sub-codebooks
// of 20 floats) or 32 (one column tile: half the matrix instructions per x operand, but the block fits up to d ~ 960 --
// d = 768 / M = 48, the size of BASELINE configs[4], which until round 4 materialised rx through the chunk loop).
template <int DP, int T, bool SPLITK, bool ODD, bool TAIL, int NS = 64>
__global__ __launch_bounds__(512, 2) void k_opq_encode_fused2(OpqFusedArgs a)
{
    static_assert(DP % 2 == 0 && DP >= 2 && DP <= 32, "even sub-dimension up to 32");
    static_assert(T >= 2 && T <= 8, "2 .. 8 centroid tiles");
    static_assert(NS == 64 || (NS == 32 && 32 % DP == 0), "64 slots, or 32 slots of whole sub-vectors");
    constexpr int NWAVE = 8;
    constexpr int S = DP / 2;            // k-steps of one distance chain
    constexpr int NM = NS / DP;          // subquantizers per column block
    constexpr int GS = NS * 4;           // floats per 4-k group of the P image
    constexpr long long kKeyInit = 0x7fffffffffffffffll;
    extern __shared__ __attribute__((aligned(16))) float smem_f2[];
    const int d = a.d;
    const int ngroups = (d + 3) >> 2;                       // 4-k groups of the P image
    float* pl = smem_f2;                                    // [ngroups + 1][NS slots][4]  (+1: pre-reads past the last group)
    float* frag_s = pl + (size_t)(ngroups + 1) * GS;        // [NM][T][S][64]
    float* cc_s = frag_s + (size_t)NM * T * S * 64;         // [NM][256]
    long long* slot_s = reinterpret_cast<long long*>(cc_s + NM * 256);   // [8 waves][64 lanes]; exact path: 64 floats of scratch per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int cbk = (int)(q % a.ncb);
    const int64_t rg_local = q / a.ncb;
    const int64_t rg = rg_local * 8 + xcd;
    const int m0 = cbk * NM;                                // first subquantizer of this column block
    const int col0 = m0 * DP;

    // ---- stage the P block, columns permuted into MFMA-result slots (see header) ----
    {
        constexpr int NV4 = NM * DP / 4;                    // float4 per P row of this column block
        static_assert((NM * DP) % 4 == 0, "whole float4 columns");
        auto slot_of = [](int lc) {                         // local column -> MFMA-result slot
            const int t = lc >> 5, l = lc & 31, r = l >> 1, hh = l & 1;
            return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh;
        };
        const int total = d * NV4;
        for (int i0 = tid; i0 < total; i0 += 512 * 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + 512 * u;
                const int k = idx / NV4, c4 = idx - k * NV4;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < total && col0 + 4 * c4 < d) v[u] = *reinterpret_cast<const f32x4*>(a.P + (int64_t)k * d + col0 + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + 512 * u;
                if (idx < total) {
                    const int k = idx / NV4, c4 = idx - k * NV4;
                    const int inner = ((k & 1) << 1) | ((k >> 1) & 1);  // (k0, k1, k2, k3) -> (k0, k2, k1, k3)
                    float* base = pl + (k >> 2) * GS + inner;
#pragma unroll
                    for (int e = 0; e < 4; ++e) base[slot_of(4 * c4 + e) << 2] = v[u][e];
                }
            }
        }
        // zero the unused slots (local columns NM * DP .. 63) of every k
        constexpr int NPAD = NS - NM * DP;
        for (int idx = tid; idx < ngroups * 4 * NPAD; idx += 512) {
            const int k = idx / (NPAD > 0 ? NPAD : 1), lc = NM * DP + idx % (NPAD > 0 ? NPAD : 1);
            const int inner = ((k & 1) << 1) | ((k >> 1) & 1);
            pl[(k >> 2) * GS + (slot_of(lc) << 2) + inner] = 0.f;
        }
    }
    // fragments and norms of the block's sub-codebooks (clamped to the last real subquantizer: a ragged block re-reads it, unused)
    for (int idx = tid; idx < NM * T * S * 64; idx += 512) {
        const int ml = idx / (T * S * 64), rest = idx - ml * (T * S * 64);
        const int m = (m0 + ml < a.M) ? m0 + ml : a.M - 1;
        frag_s[idx] = a.frags[(int64_t)m * T * S * 64 + rest];
    }
    for (int idx = tid; idx < NM * 256; idx += 512) {
        const int ml = idx >> 8, jj = idx & 255;
        cc_s[idx] = (m0 + ml < a.M && jj < a.k_pad) ? a.cc[(int64_t)(m0 + ml) * a.k_pad + jj] : __builtin_inff();
    }
    slot_s[tid] = kKeyInit;
    __syncthreads();
    if (rg_local >= a.rg_per_xcd) return;
    const int64_t wg_row0 = rg * a.rows_per_wg;
    if (wg_row0 >= a.n) return;
    int64_t wg_row1 = wg_row0 + a.rows_per_wg;
    if (wg_row1 > a.n) wg_row1 = a.n;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long* slot = slot_s + wave * 64 + lane;
    const float* plane = pl + 4 * j + 2 * h;     // + 256 floats per 4-k group; + 128: second slot tile
    const int nfull = d >> 5;                    // bursts of 32 k in which every piece is real
    const int tail_groups = (d & 31) >> 2;       // 4-k groups of the partial last burst (TAIL: 1..7)
    const int nb = nfull + (TAIL ? 1 : 0);
    constexpr int KB = kKC / 32;                 // bursts per rule-2 block

    int lo[16];                                  // centroid offset of accumulator register r inside a 32-centroid tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        lo[r] = (r & 3) + 8 * (r >> 2);
        asm volatile("" : "+v"(lo[r]));
    }

    const int ntile = (int)((wg_row1 - wg_row0 + 31) >> 5);
    int64_t row0 = wg_row0 + 32 * wave;
    if (row0 >= wg_row1) return;
    int cur_tile = wave;
    auto row_ptr = [&](int64_t r0) {             // this lane's row of the tile at r0 (clamped to the last row), its half's 16 k
        const int64_t r = (r0 + j < a.n) ? r0 + j : a.n - 1;
        return a.x + r * a.x_rs + 16 * h;
    };
    auto load_burst = [&](f32x4 (&s)[4], const float* pb, int bi, bool full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (full || 32 * bi + 16 * h + 4 * e < d) s[e] = *reinterpret_cast<const f32x4*>(pb + 32 * bi + 4 * e);
            else s[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x4 sa[4], sb[4];
    const float* prow = row_ptr(row0);
    load_burst(sa, prow, 0, nfull > 0);
    Rot8Ops ops;
    f32x2 ops32 = {0.f, 0.f};
    if constexpr (NS == 64) rot8_read(ops, plane); else rot32_read(ops32, plane);
    unsigned long long st_tiles = 0, st_rot = 0, st_enc = 0;
    const unsigned long long st_t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (;;) {
        const unsigned long long st_a = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
        const int next_tile = cur_tile + NWAVE;
        const bool has_next = next_tile < ntile;
        const int64_t next_row0 = wg_row0 + 32 * (int64_t)next_tile;
        const float* pnext = has_next ? row_ptr(next_row0) : prow;
        // ================= rotation (k_rotate_pblock8's burst loop): t[ct][r] = rx[row j][col0 + 2 (16 ct + r) + h] =================
        f32x16 c0 = zero, c1 = zero, t0 = zero, t1 = zero;
#define F2_STEP(cu, nx, bi, IS_TAIL)                                                                   \
        {                                                                                              \
            float xo_[16];                                                                             \
            rot8_swap(cu, xo_);                                                                        \
            if ((bi) + 1 < nfull) load_burst(nx, prow, (bi) + 1, true);                                \
            else if (TAIL && (bi) + 1 == nfull) load_burst(nx, prow, nfull, false);                    \
            else if (has_next) load_burst(nx, pnext, 0, nfull > 0);                                    \
            const float* pn_ = ((bi) + 1 < nb) ? plane + ((bi) + 1) * 8 * GS : plane;                  \
            if constexpr (NS == 64) {                                                                  \
                if (!(IS_TAIL)) rot8_burst<true>(plane + (bi) * 8 * GS, pn_, 8, xo_, c0, c1, ops);     \
                else rot8_burst<false>(plane + (bi) * 8 * GS, pn_, tail_groups, xo_, c0, c1, ops);     \
            } else {                                                                                   \
                if (!(IS_TAIL)) rot32_burst<true>(plane + (bi) * 8 * GS, pn_, 8, xo_, c0, ops32);      \
                else rot32_burst<false>(plane + (bi) * 8 * GS, pn_, tail_groups, xo_, c0, ops32);      \
            }                                                                                          \
        }
#define F2_BLOCK(bi)                                                                                   \
        if (SPLITK && (bi) > 0 && ((bi) % KB) == 0) {                                                  \
            if ((bi) == KB) { t0 = c0; t1 = c1; }                                                      \
            else {                                                                                     \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) { t0[r] = fadd(t0[r], c0[r]); t1[r] = fadd(t1[r], c1[r]); } \
            }                                                                                          \
            c0 = zero; c1 = zero;                                                                      \
        }
        int bi = 0;
        for (; bi + 2 <= nfull; bi += 2) {
            F2_BLOCK(bi);
            F2_STEP(sa, sb, bi, false);
            F2_STEP(sb, sa, bi + 1, false);
        }
        if (ODD) {
            F2_BLOCK(bi);
            F2_STEP(sa, sb, bi, false);
            if (TAIL) {
                F2_STEP(sb, sa, bi + 1, true);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[e] = sb[e];
            }
        } else if (TAIL) {
            F2_BLOCK(bi);
            F2_STEP(sa, sb, bi, true);
#pragma unroll
            for (int e = 0; e < 4; ++e) sa[e] = sb[e];
        }
#undef F2_STEP
#undef F2_BLOCK
        if (SPLITK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t0[r] = fadd(t0[r], c0[r]); t1[r] = fadd(t1[r], c1[r]); }
        } else {
            t0 = c0; t1 = c1;
        }
        unsigned long long st_b = 0;
        if (a.stamps) { asm volatile("" ::"v"(t0), "v"(t1)); st_b = __builtin_amdgcn_s_memtime(); }

        // ================= encode the NM sub-vectors held in t0 / t1 =================
        // One flattened, fully unrolled pipeline over the steps g = ml * T + t (t = centroid tile): while the
        // matrix core runs the chain of step g + 1, the VALU turns the 16 distances of step g into keys and the
        // LDS unit folds them (no-return ds_min_i64 into the lane's slot, read back and re-armed by one
        // ds_wrxchg behind them).  Codebook fragments come from the workgroup's LDS image two steps ahead: step g's set
        // lives in fa (g even) / fb (g odd) and is reloaded with step g + 2 as soon as its chain has been issued.
        const int64_t row = row0 + j;
        const bool valid = row < a.n;
        const int nm_valid = (a.M - m0 < NM) ? a.M - m0 : NM;      // wave-uniform; >= 1
        const int G = nm_valid * T;                                 // real steps of this row tile
        const float* fpb = frag_s + lane;                                // step g: fpb + g * S * 64 (LDS image of this block's sub-codebooks)
        float fa[S], fb[S];
#pragma unroll
        for (int s = 0; s < S; ++s) fa[s] = fpb[s * 64];
#pragma unroll
        for (int s = 0; s < S; ++s) fb[s] = fpb[(S + s) * 64];       // T >= 2: step 1 exists

        // ---- ||rx_m||^2 of every sub-vector, rule 1 (unrolled_dot): lane half h holds the elements e = 2 s + h ----
        float xxm[NM];
#pragma unroll
        for (int ml = 0; ml < NM; ++ml) {
            constexpr int C = DP / 8;            // full chunks of 8
            constexpr int NT = (DP - 8 * C) / 2; // tail elements per half
            float sq[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int SS = ml * S + s;       // constant after unrolling
                const float v = (SS < 16) ? t0[SS & 15] : t1[SS & 15];
                sq[s] = fmul(v, v);
            }
            float sum = 0.f;
            if (C > 0) {
                // p[i] (l = 2 i + h) = sq of elements l, 8 + l, 16 + l, ..  = k-steps i, 4 + i, 8 + i, ..
                float p[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[i] = sq[i];                // 0 + x == x exactly for x >= +0 or NaN
#pragma unroll
                    for (int c = 1; c < C; ++c) p[i] = fadd(p[i], sq[4 * c + i]);
                }
                // half 0: (p0 + p4, p2 + p6); half 1: (p1 + p5, p3 + p7)
                const float u0 = fadd(p[0], p[2]), u1 = fadd(p[1], p[3]);
                float e0, o0, e1, o1;
                halves(u0, e0, o0);
                halves(u1, e1, o1);
                sum = fadd(fadd(fadd(e0, o0), e1), o1);   // 0 + (p0 + p4) is exact
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float e, o;
                halves(sq[4 * C + i], e, o);
                sum = fadd(fadd(sum, e), o);
            }
            xxm[ml] = sum;
        }

        f32x16 acc = zero;
#pragma unroll
        for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], t0[s], acc, 0, 0, 0);   // step 0: k-steps 0 .. S-1

#pragma unroll
        for (int ml = 0; ml < NM; ++ml) {
            if (ml < nm_valid) {                 // wave-uniform (ragged last column block)
                const int m = m0 + ml;
                float best = __builtin_inff();
                int bidx = 0;
                long long pending = kKeyInit;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int g = ml * T + t;                       // everything below is static after unrolling
                    const bool next = !(t == T - 1 && ml == NM - 1);
                    const int mln = (t == T - 1 && ml + 1 < NM) ? ml + 1 : ml;
                    float (&FN)[S] = ((g + 1) & 1) ? fb : fa;        // fragments of step g + 1
                    float (&FL)[S] = (g & 1) ? fb : fa;              // set of step g: its chain is issued, reload with g + 2
                    f32x4 c4[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        c4[g4] = *reinterpret_cast<const f32x4*>(&cc_s[ml * 256 + 32 * t + 8 * g4 + 4 * h]);
                    {
                        const int gl = (g + 2 < G) ? g + 2 : G - 1;  // clamped: the load is unconditional
                        const float* fl = fpb + gl * S * 64;
#pragma unroll
                        for (int s = 0; s < S; ++s) FL[s] = fl[s * 64];
                    }
                    if (t > 0) {
                        const float dprev = __int_as_float((int)(pending >> 32));
                        const bool lt = dprev < best;
                        best = lt ? dprev : best;
                        bidx = lt ? ((int)(unsigned)pending + 32 * (t - 1)) : bidx;
                    }
                    const f32x2 xx2 = {xxm[ml], xxm[ml]};
                    long long key[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x2 c01 = {c4[g4][0], c4[g4][1]}, c23 = {c4[g4][2], c4[g4][3]};
                        f32x2 t01, t23;
                        asm("v_pk_add_f32 %0, %1, %2" : "=v"(t01) : "v"(xx2), "v"(c01));
                        asm("v_pk_add_f32 %0, %1, %2" : "=v"(t23) : "v"(xx2), "v"(c23));
                        const float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int r = 4 * g4 + qq;
                            const float dd = ffma(acc[r], -2.0f, tt[qq]);
                            key[r] = ((long long)__float_as_int(dd) << 32) | (long long)(unsigned)lo[r];
                        }
                        asm volatile("" ::"v"(t01), "v"(t23));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 nacc = zero;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (next) {
                            const int SS = mln * S + s;
                            nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(FN[s], (SS < 16) ? t0[SS & 15] : t1[SS & 15], nacc, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = (16 * s) / S; r < (16 * (s + 1)) / S; ++r)
                            (void)__hip_atomic_fetch_min(slot, key[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    pending = __hip_atomic_exchange(slot, kKeyInit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = nacc;
                }
                {   // last step's slot
                    const float dprev = __int_as_float((int)(pending >> 32));
                    const bool lt = dprev < best;
                    best = lt ? dprev : best;
                    bidx = lt ? ((int)(unsigned)pending + 32 * (T - 1)) : bidx;
                }
                const bool neg = best < 0.f;
                bidx += 4 * h;
                const float od = __shfl_xor(best, 32);
                const int oi = __shfl_xor(bidx, 32);
                if (od < best || (od == best && oi < bidx)) bidx = oi;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && (!(xxm[ml] < kBigNorm) || neg));
                const unsigned need = (unsigned)(bal | (bal >> 32));
                if (h == 0 && valid && !((need >> j) & 1u)) a.out[row * a.o_rs + m] = (uint8_t)bidx;
                if (need) {
                    opq_rows_slow(a.x, a.x_rs, a.P, d, a.out, a.o_rs, a.cb, a.cc, a.K, DP, a.k_pad, m, row0, need,
                                  reinterpret_cast<float*>(slot_s + wave * 64));
                    *slot = kKeyInit;            // the scratch overlaid the wave's slots
                }
            }
        }
        if (a.stamps) {
            const unsigned long long st_c = __builtin_amdgcn_s_memtime();
            st_tiles += 1; st_rot += st_b - st_a; st_enc += st_c - st_b;
        }
        prow = pnext;
        if (!has_next) break;
        row0 = next_row0;
        cur_tile = next_tile;
    }
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 5;
        o[0] = st_tiles; o[1] = st_rot; o[2] = st_enc;
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

}  // namespace pqhip
