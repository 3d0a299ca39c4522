// kernels_opq_fused.hip.h -- OPQ encode in ONE kernel: rx = x.dot(P) (pq.rs:276) never leaves the
// register file; the rotation's accumulators ARE the encode's matrix operands.
// (Template kernels, instantiated from pqhip.hip only.)
//
// Why this shape.  Both halves of the work need a stationary operand: the rotation a column block of
// P for all k (76.8 KB for 64 columns at d = 300), the encode the MFMA fragments of a sub-codebook
// (20 KB per subquantizer at K = 256, dsub = 20).  160 KB of LDS cannot hold both next to the x
// staging slabs, so the P block stays in LDS -- it is read 2 x 150 times per 32-row tile -- and the
// codebook fragments, which are read once per (tile, subquantizer) and are exactly one coalesced
// dword per lane per MFMA, come straight from L2 (0.6 MB working set, 8 waves of a workgroup walking
// the same sequence), two steps ahead of their use.
//
// Measured (MI355X, 10 M x 300, M = 15, K = 256; in-kernel s_memtime stamps, PQHIP_DEBUG_FUSED_STAMP):
// 36.5 ms per launch = 0.58 of the FP32-MFMA peak, against 34.5 ms for k_rotate_pblock5 + k_encode_mfma_lds3
// through a pooled 2 GiB scratch.  Per 32-row tile and wave (two waves share a SIMD; 69 k cycles would be
// 100 %): rotation 49 k, encode 54 k.  With the fragments read from LDS instead (wrong data, timing
// only) the launch takes 32.7 ms: the L2 round trip of the fragment stream costs 10 %, and LDS has no
// room for them at dsub = 20 (P block 76.8 KB + 8 double-buffered x slabs 73.7 KB + norms and slots
// 7 KB = 157.5 of 160 KB).  Deeper x prefetch (two register sets) and a vectorised P staging changed
// nothing measurable; 8192 rows per workgroup is 0.5 % better than 4096, 16384 is 7 % worse.
// The dispatcher therefore takes this kernel only on request (encode variant 5): no scratch buffer at
// all, 5 % slower than the two-kernel path.
//
// Orientation.  k_rotate_pblock5 computes a 32-row x 64-column tile with the rows in the accumulator
// REGISTERS and the columns on the LANES.  Swapping the two MFMA operands (P fragment as A, x fragment
// as B -- the LDS images are unchanged) transposes the result: lane (row j, half h) then holds, in
// register r of column tile t, the column in "slot" i = (r & 3) + 8 (r >> 2) + 4 h.  P's columns are
// permuted while they are staged so that slot i of tile t holds local column 32 t + 2 r + h: register r
// of lane (j, h) is rx[row j][k = 2 (16 t + r) + h] -- precisely the B operand of encode k-step
// S = 16 t + r of the v_mfma_f32_32x32x2_f32 distance chain (B[k = lane >> 5][j = lane & 31]).
// A column block holds NM = 64 / dsub whole subquantizers (3 at dsub = 20: 60 of 64 slots used).
//
// Arithmetic is CANON-F32 throughout: the rotation chains are rule 2 (k-ordered fmaf chain, restart
// at k = 256, blocks added with one rounded add), ||rx_m||^2 is rule 1 evaluated across the two lane
// halves (v_permlane32_swap exchanges the partial sums, the adds keep ndarray's order), distances and
// the argmin are the LDS-atomic epilogue of k_encode_mfma_lds3.  Rows that need the exact path (NaN /
// Inf / huge norms, a negative fast minimum) are re-rotated by a scalar rule-2 chain and scanned
// exactly, so codes equal the oracle's for every input.
#pragma once
#include "kernels_mfma.hip.h"

namespace pqhip {

struct OpqFusedArgs {
    const float* x;      // [n][x_rs]
    int64_t n;
    int64_t x_rs;
    const float* P;      // [d][d] row-major, applied as x.dot(P)
    int d;
    const float* frags;  // [M][T][S][64]
    const float* cc;     // [M][k_pad]
    const float* cb;     // [M][K][dsub]  (exact path)
    uint8_t* out;        // [n][o_rs]
    int64_t o_rs;
    int M, K, k_pad;     // (the number of 32-centroid tiles T is a template parameter)
    int rows_per_wg;     // multiple of 256
    int ncb;             // column blocks = ceil(M / NM)
    int64_t rg_per_xcd;
    unsigned long long* stamps;   // diagnostics only (PQHIP_DEBUG_FUSED_STAMP): per wave {tiles, rotation cycles, encode cycles, wave cycles, wave realtime ticks}
};

// lower / upper 32 lanes of v broadcast to both halves: e = v of lane (l & 31), o = v of lane (l | 32)
__device__ __forceinline__ void halves(float v, float& e, float& o)
{
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    e = __uint_as_float(r[0]);
    o = __uint_as_float(r[1]);
}

// Exact path for flagged rows of a tile: lanes 0 .. dsub-1 re-rotate the row's sub-vector with the
// literal rule-2 chain, park it in the wave's LDS scratch, then the whole wave scans the K centroids
// with the literal three-operation distance (as encode_rows_slow_v).
__device__ __noinline__ void opq_rows_slow(const float* x, int64_t x_rs, const float* P, int d, uint8_t* out, int64_t o_rs,
                                           const float* cb, const float* cc, int K, int dsub, int k_pad, int m,
                                           int64_t row0, unsigned need, float* scratch /* >= 64 floats, wave-private */)
{
    const int lane = threadIdx.x & 63;
    const float* cbm = cb + (int64_t)m * K * dsub;
    const float* ccm = cc + (int64_t)m * k_pad;
    while (need) {  // wave-uniform
        const int jr = __builtin_ctz(need);
        need &= need - 1;
        const int64_t row = row0 + jr;
        for (int e = lane; e < dsub; e += 64)
            scratch[e] = chain_dot_global(x + row * x_rs, 1, P + (int64_t)m * dsub + e, d, d);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the stores above are visible to the wave's reads below
        __builtin_amdgcn_wave_barrier();
        float xx;
        {
            float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int i = 0;
            for (; dsub - i >= 8; i += 8)
                for (int l = 0; l < 8; ++l) p[l] = fadd(p[l], fmul(scratch[i + l], scratch[i + l]));
            float s = 0.f;
            s = fadd(s, fadd(p[0], p[4]));
            s = fadd(s, fadd(p[1], p[5]));
            s = fadd(s, fadd(p[2], p[6]));
            s = fadd(s, fadd(p[3], p[7]));
            for (; i < dsub; ++i) s = fadd(s, fmul(scratch[i], scratch[i]));
            xx = s;
        }
        float bd = 0.f;
        int bj = 0x7fffffff;
        for (int j = lane; j < K; j += 64) {
            const float* c = cbm + (int64_t)j * dsub;
            float dp = 0.f;
            for (int k = 0; k < dsub; ++k) dp = ffma(scratch[k], c[k], dp);   // dsub <= 32 < 256: one chain
            const float dd = fsub(fadd(xx, ccm[j]), fadd(dp, dp));
            if (bj == 0x7fffffff || of_less(dd, bd)) { bd = dd; bj = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(bd, off);
            const int oj = __shfl_xor(bj, off);
            const bool take = oj != 0x7fffffff &&
                              (bj == 0x7fffffff || of_less(od, bd) || (of_equal(od, bd) && oj < bj));
            if (take) { bd = od; bj = oj; }
        }
        if (lane == 0) out[row * o_rs + m] = (uint8_t)bj;
        __builtin_amdgcn_wave_barrier();
    }
}

template <int DP, int T>
__global__ __launch_bounds__(512, 2) void k_opq_encode_fused(OpqFusedArgs a)
{
    static_assert(DP % 2 == 0 && DP >= 2 && DP <= 32, "even sub-dimension up to 32");
    static_assert(T >= 2 && T <= 8, "2 .. 8 centroid tiles");
    constexpr int S = DP / 2;            // k-steps of one distance chain
    constexpr int NM = 64 / DP;          // subquantizers per 64-slot column block
    constexpr int XS = 36;               // x slab row stride in floats (144 B: 16-B aligned, 2-way banks)
    constexpr long long kKeyInit = 0x7fffffffffffffffll;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int d = a.d;
    const int ngroups = (d + 3) >> 2;                       // 4-k groups of the P image
    float* pl = smem_f;                                     // [ngroups][64 slots][4]
    float* xs_all = pl + (size_t)ngroups * 256;             // [8 waves][2][32][XS]
    float* cc_s = xs_all + 8 * 2 * 32 * XS;                 // [NM][256]
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
    // 16-byte loads, four in flight per thread before the first LDS store: one L2 round trip per batch
    // instead of one per element (the element-wise loop cost ~90 k cycles per workgroup, 10 % of its life).
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
                    float* base = pl + (((k >> 2) << 6) << 2) + inner;
#pragma unroll
                    for (int e = 0; e < 4; ++e) base[slot_of(4 * c4 + e) << 2] = v[u][e];
                }
            }
        }
        // zero the unused slots (local columns NM * DP .. 63) of every k
        constexpr int NPAD = 64 - NM * DP;
        for (int idx = tid; idx < ngroups * 4 * NPAD; idx += 512) {
            const int k = idx / (NPAD > 0 ? NPAD : 1), lc = NM * DP + idx % (NPAD > 0 ? NPAD : 1);
            const int inner = ((k & 1) << 1) | ((k >> 1) & 1);
            pl[((((k >> 2) << 6) + slot_of(lc)) << 2) + inner] = 0.f;
        }
        // (k rows d .. 4 ngroups - 1 do not exist: d % 4 == 0 is a launch condition)
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
    float* xs = xs_all + (size_t)wave * 2 * 32 * XS;
    long long* slot = slot_s + wave * 64 + lane;
    const float* plane = pl + 4 * j + 2 * h;     // + q * 256 floats per group; + 128: second slot tile
    const int nslab = (d + 31) >> 5;
    const int tail_groups = (d - 32 * (nslab - 1) + 3) / 4;  // 4-k groups with real k in the last slab
    constexpr int SB = kKC / 32;                 // slabs per rule-2 block
    const int lr = lane >> 3, lc8 = lane & 7;    // staging role: rows lr + 8 i, 16-byte piece lc8

    const float* rp[4];
    auto set_rows = [&](int64_t row0) {
        const int left = (int)((a.n - row0 < 32) ? a.n - row0 : 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 8 * i;
            rp[i] = a.x + ((r < left) ? row0 + r : a.n - 1) * a.x_rs + 4 * lc8;
        }
    };
    f32x4 st[4];
    auto fetch = [&](int slab) {                 // 8 lanes x 16 B = one row's 128 contiguous bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 32 * slab + 4 * lc8;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < d) v = *reinterpret_cast<const f32x4*>(rp[i] + 32 * slab);
            st[i] = v;
        }
    };
    auto stash = [&](int buf) {                  // (k0, k1, k2, k3) -> (k0, k2, k1, k3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 w = {st[i][0], st[i][2], st[i][1], st[i][3]};
            *reinterpret_cast<f32x4*>(xs + ((size_t)buf * 32 + lr + 8 * i) * XS + 4 * lc8) = w;
        }
    };

    int lo[16];                                  // centroid offset of accumulator register r inside a 32-centroid tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        lo[r] = (r & 3) + 8 * (r >> 2);
        asm volatile("" : "+v"(lo[r]));
    }

    int64_t row0 = wg_row0 + 32 * wave;
    if (row0 >= wg_row1) return;
    set_rows(row0);
    fetch(0);
    stash(0);
    unsigned long long st_tiles = 0, st_rot = 0, st_enc = 0;
    const unsigned long long st_t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (; row0 < wg_row1; row0 += 256) {
        const unsigned long long st_a = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
        // ================= rotation: tot[t][r] = rx[row j][col0 + 2 (16 t + r) + h] =================
        f32x16 tot0 = zero, tot1 = zero;
        // slab s is multiplied out of LDS buffer s & 1 while slab s + 1 travels from HBM / L2 into registers
        auto multiply = [&](int slab, f32x16& c0, f32x16& c1) {
            const float* arow = xs + ((size_t)(slab & 1) * 32 + j) * XS + 2 * h;
            const float* pq = plane + slab * 8 * 256;
            // operands of group u + 1 are requested before the four MFMAs of group u are issued, so an
            // LDS round trip (2-way banked reads) hides behind 256 cycles of matrix work
            const int ng = (slab + 1 < nslab || tail_groups == 8) ? 8 : tail_groups;   // wave-uniform
            f32x2 xv = *reinterpret_cast<const f32x2*>(arow);
            f32x2 p0 = *reinterpret_cast<const f32x2*>(pq);
            f32x2 p1 = *reinterpret_cast<const f32x2*>(pq + 128);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u < ng) {
                    const f32x2 xc = xv, q0 = p0, q1 = p1;
                    if (u + 1 < 8) {
                        // (reads past the last real group stay inside the LDS allocation and are not used)
                        xv = *reinterpret_cast<const f32x2*>(arow + 4 * (u + 1));
                        p0 = *reinterpret_cast<const f32x2*>(pq + (u + 1) * 256);
                        p1 = *reinterpret_cast<const f32x2*>(pq + (u + 1) * 256 + 128);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[0], xc[0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[0], xc[0], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[1], xc[1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[1], xc[1], c1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        for (int sb = 0; sb < nslab; sb += SB) {
            const int se = (sb + SB < nslab) ? sb + SB : nslab;
            f32x16 c0 = zero, c1 = zero;
            for (int slab = sb; slab < se; ++slab) {
                const bool more = slab + 1 < nslab;
                if (more) fetch(slab + 1);
                multiply(slab, c0, c1);
                if (more) stash((slab + 1) & 1);
            }
            if (sb == 0) { tot0 = c0; tot1 = c1; }
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot0[r] = fadd(tot0[r], c0[r]); tot1[r] = fadd(tot1[r], c1[r]); }
            }
        }
        const bool has_next = row0 + 256 < wg_row1;
        unsigned long long st_b = 0;
        if (a.stamps) { asm volatile("" ::"v"(tot0), "v"(tot1)); st_b = __builtin_amdgcn_s_memtime(); }

        // ================= encode the NM sub-vectors held in tot0 / tot1 =================
        // One flattened, fully unrolled pipeline over the steps g = ml * T + t (t = centroid tile): while the
        // matrix core runs the chain of step g + 1, the VALU turns the 16 distances of step g into keys and the
        // LDS unit folds them (no-return ds_min_i64 into the lane's slot, read back and re-armed by one
        // ds_wrxchg behind them).  Codebook fragments come from L2 TWO steps ahead: step g's set lives in
        // fa (g even) / fb (g odd) and is reloaded with step g + 2 as soon as its chain has been issued.
        const int64_t row = row0 + j;
        const bool valid = row < a.n;
        const int nm_valid = (a.M - m0 < NM) ? a.M - m0 : NM;      // wave-uniform; >= 1
        const int G = nm_valid * T;                                 // real steps of this row tile
        const float* fpb = a.frags + (int64_t)m0 * T * S * 64 + lane;   // step g: fpb + g * S * 64
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
                const float v = (SS < 16) ? tot0[SS & 15] : tot1[SS & 15];
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
        for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], tot0[s], acc, 0, 0, 0);   // step 0: k-steps 0 .. S-1

#pragma unroll
        for (int ml = 0; ml < NM; ++ml) {
            // the next tile's first slab travels from HBM while the last sub-vector is encoded
            if (has_next && ml == nm_valid - 1) { set_rows(row0 + 256); fetch(0); }
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
                        const float* fl = fpb + (int64_t)gl * S * 64;
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
                            nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(FN[s], (SS < 16) ? tot0[SS & 15] : tot1[SS & 15], nacc, 0, 0, 0);
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
        if (has_next) stash(0);
    }
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 5;
        o[0] = st_tiles; o[1] = st_rot; o[2] = st_enc;
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

}  // namespace pqhip
