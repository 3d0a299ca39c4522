// kernels_rotate8.hip.h -- OPQ rotation GEMM, eighth version: P-block stationary, x rows straight from
// global memory into the MFMA operand registers, operands swapped so that the result tile leaves the
// accumulators as 16-byte row pieces.  (Template kernels: instantiated from pqhip_rotate.hip; the burst helpers are shared with kernels_opq_fused2.hip.h.)
//
//   out[n][c] = sum_k x[n][k] * Pm[k][c]   (pq.rs:276 with Pm = projection, pq.rs:324 with Pm = projection^T),
//   rule-2 chains: one k-ordered fmaf chain per output element, restarted every 256 k, blocks added in order.
//
// What round 2's stamps said about v6 (k_rotate_pblock6, removed from the library in round 4): its k loop is 90 % matrix issue, but every 32-row
// tile ends in an 8.5 k-cycle MFMA-free epilogue (32 ds_write_b32 + 8 ds_read_b128 to transpose the tile for
// 16-byte stores) and every 16-k slab of x goes global -> registers -> LDS -> registers with a vmcnt wait in
// front of the LDS store that exposes the fetch (-14 % without it).  This version removes both LDS round trips:
//   * x is the B operand and comes from global memory in the lane-per-row form of the encode kernel: lane
//     (row j, half h) loads 64 contiguous bytes of its row per 32-k burst (half 0: k 0..15, half 1: k 16..31 --
//     the two halves cover one whole 128-byte line per row), and one v_permlane32_swap per register pair turns
//     (x[k], x[k+1] | x[k+16], x[k+17]) into the operands of k-steps k/2 and k/2 + 8.  The next burst is in
//     flight while the current one feeds 32 MFMAs (16 k-steps x 2 column tiles), across tile boundaries too.
//   * Pm's 64-column block is the A operand (LDS image as in v5/v6), so the accumulator holds the tile with the
//     ROWS on the lanes and four consecutive COLUMNS in registers 4g..4g+3: the tile is stored with eight
//     global_store_dwordx4 straight from the accumulators, no LDS, no epilogue barrier or wait.
// LDS holds the P block only (76.8 KB at d = 300), so the kernel takes any d with ceil(d/4) + 1 KB-groups <= 160 KB
// (d <= 636; v6 stopped at 320).  Requires 16-byte aligned rows and d % 4 == 0, like v5/v6.
#pragma once
#include "kernels_mfma.hip.h"

// Timing ablations (results wrong by construction) exist only for tools/rot8_ablate.hip, which defines
// PQHIP_TIMING_ONLY_BUILD and never links into libpqhip.so: 1 no LDS operand re-reads, 2 no x loads after a wave's
// first burst, 3 no stores, 4 all three (matrix instructions and operand swaps only).
#ifndef ROT8_ABLATE
#define ROT8_ABLATE 0
#elif ROT8_ABLATE != 0 && !defined(PQHIP_TIMING_ONLY_BUILD)
#error "ROT8_ABLATE produces wrong results: timing-only tool builds (PQHIP_TIMING_ONLY_BUILD) only"
#endif

#ifndef ROT8_STORE
#define ROT8_STORE 1      // 1: plain 16-byte stores, 0: nontemporal ones (A/B switch of tools/rot8_ablate.hip).  Each store instruction
                          // writes 32 bytes of 32 different rows, so a line is completed by four instructions; with the streaming
                          // hint the launch takes 2.09 ms instead of 1.81 ms (1.18 M x 300 rows, same box)
#endif

namespace pqhip {

// GATHER form (pq.rs:323-326 without the intermediate matrix): the "x" of the rotation is the reconstruction of a code row,
// x[row][k] = cb[m][codes[row][m]][k - m dsub] (primitives.rs:141-147), fetched piece by piece from the L2-resident codebook
// while the previous burst multiplies -- the gathered rows never exist in memory.  Needs dsub % 4 == 0 (a 16-byte piece
// lies inside one sub-vector).  sel_rows != nullptr: output row i reconstructs code row sel_rows[i] (lookup form).
struct Rot8Gather {
    const uint8_t* codes = nullptr;   // [n_codes or n][c_rs], 1-byte codes
    int64_t c_rs = 0;
    const float* cb = nullptr;     // [M][K][dsub]
    int K = 0, dsub = 0;
    unsigned inv_dsub = 0;         // ceil(2^32 / dsub): k / dsub == umulhi(k, inv_dsub) for k < 65536
    const int64_t* sel_rows = nullptr;
    int64_t n_codes = 0;
    int* err = nullptr;            // "code >= K / row index out of range" flag (as k_reconstruct)
};

struct Rot8Ops {           // LDS operands of one 4-k group: two k-steps x two column tiles
    f32x2 p0, p1;
};

__device__ __forceinline__ void rot8_read(Rot8Ops& o, const float* plane_g)
{
    o.p0 = *reinterpret_cast<const f32x2*>(plane_g);
    o.p1 = *reinterpret_cast<const f32x2*>(plane_g + 128);
}

// One 32-k burst (NG = 8 groups) or a partial one (ng < 8 real groups, wave-uniform): the LDS operands of group
// g + 1 -- of the next burst's (or next tile's) first group after the last one -- are requested before the four
// MFMAs of group g issue.  xo[s] = B operand of k-step s (half 0: k = 2 s, half 1: k = 2 s + 1).
template <bool FULL>
__device__ __forceinline__ void rot8_burst(const float* plane_b, const float* plane_next, int ng, const float (&xo)[16],
                                           f32x16& c0, f32x16& c1, Rot8Ops& cur)
{
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (!FULL && g >= ng) break;
        Rot8Ops nxt = cur;
        const bool last = FULL ? (g == 7) : (g + 1 == ng);
        if (ROT8_ABLATE != 1 && ROT8_ABLATE != 4) rot8_read(nxt, last ? plane_next : plane_b + (g + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
        if (ROT8_ABLATE == 5) {
            // timing only: the same flop, registers and operands as 2 x v_mfma_f32_16x16x4_f32 per 32x32x2 instruction
            auto pair16 = [](float av, float bv, f32x16& c, int idx) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q4 = (2 * idx + u) & 3;
                    f32x4 part = {c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]};
                    part = __builtin_amdgcn_mfma_f32_16x16x4f32(u ? bv : av, u ? av : bv, part, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[4 * q4 + e] = part[e];
                }
            };
            pair16(cur.p0[0], xo[2 * g], c0, 0);
            pair16(cur.p1[0], xo[2 * g], c1, 0);
            pair16(cur.p0[1], xo[2 * g + 1], c0, 1);
            pair16(cur.p1[1], xo[2 * g + 1], c1, 1);
        } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.p0[0], xo[2 * g], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.p1[0], xo[2 * g], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.p0[1], xo[2 * g + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.p1[1], xo[2 * g + 1], c1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// raw burst (4 x 16 bytes of this lane's row half) -> the 16 k-step operands, in k order
__device__ __forceinline__ void rot8_swap(const f32x4 (&s)[4], float (&xo)[16])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[e][0]), __float_as_uint(s[e][1]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[e][2]), __float_as_uint(s[e][3]), false, false);
        xo[2 * e] = __uint_as_float(a[0]);          // (k = 4e     | 4e + 1)
        xo[2 * e + 1] = __uint_as_float(b[0]);      // (k = 4e + 2 | 4e + 3)
        xo[8 + 2 * e] = __uint_as_float(a[1]);      // (k = 16 + 4e     | 16 + 4e + 1)
        xo[8 + 2 * e + 1] = __uint_as_float(b[1]);  // (k = 16 + 4e + 2 | 16 + 4e + 3)
    }
}

// ODD: odd number of full 32-k bursts; TAIL: d % 32 != 0 (a partial last burst).  Compile-time facts so that the burst
// sequence of a tile is ONE straight code path: with a run-time choice between the full and the partial burst the
// register allocator gave the two paths different accumulator registers and copied all 32 of them at every join.
// GATHER runs 8 waves per workgroup (two per SIMD, 256 registers each): the gather's address arithmetic does not fit beside
// two accumulator pairs and two burst buffers in the 168 registers of the three-waves-per-SIMD form (58 spilled).
template <bool SPLITK, bool ODD, bool TAIL, bool GATHER>
__global__ __launch_bounds__(GATHER ? 512 : 768, GATHER ? 2 : 3) void k_rotate_pblock8(const float* __restrict__ x, int64_t n, int64_t x_rs,
                                                           const float* __restrict__ Pm, int d, float* __restrict__ out,
                                                           int64_t o_rs, int rows_per_wg, int ncb, int64_t rg_per_xcd, Rot8Gather ga,
                                                           unsigned long long* stamps /* diagnostics: PQHIP_DEBUG_ROT_STAMP */)
{
    constexpr int NWAVE = GATHER ? 8 : 12;
    constexpr int NT = 64 * NWAVE;               // threads per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem8[];
    float* pl = smem8;                           // [ceil(d/4) + 1 groups][64 cols][4]; the last group is never used as data
    const unsigned long long st_in = stamps ? __builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int cb = (int)(q % ncb);
    const int64_t rg_local = q / ncb;
    const int64_t rg = rg_local * 8 + xcd;
    const int col0 = cb * 64;

    // stage the P block: 16-byte loads, two in flight per thread before the LDS stores; image order (k0, k2, k1, k3)
    {
        const int total = d * 16;                // float4 per block: d rows x 16
        for (int i0 = tid; i0 < total; i0 += NT * 2) {
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + NT * u;
                const int k = idx >> 4, c = col0 + 4 * (idx & 15);
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < total && c < d) v[u] = *reinterpret_cast<const f32x4*>(Pm + (int64_t)k * d + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + NT * u;
                if (idx < total) {
                    const int k = idx >> 4, c4 = idx & 15;
                    const int inner = ((k & 1) << 1) | ((k >> 1) & 1);
                    float* dst = pl + ((((k >> 2) << 6) + 4 * c4) << 2) + inner;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[4 * e] = v[u][e];
                }
            }
        }
    }
    __syncthreads();
    if (rg_local >= rg_per_xcd) return;
    const int64_t wg_row0 = rg * rows_per_wg;
    if (wg_row0 >= n) return;
    int64_t wg_row1 = wg_row0 + rows_per_wg;
    if (wg_row1 > n) wg_row1 = n;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* plane = pl + 4 * j + 2 * h;     // + 256 floats per 4-k group; + 128: second column tile
    const int nfull = d >> 5;                    // bursts of 32 k in which every piece is real
    const int tail_groups = (d & 31) >> 2;       // 4-k groups of the partial last burst (TAIL: 1..7)
    const int nb = nfull + (TAIL ? 1 : 0);
    constexpr int KB = kKC / 32;                 // bursts per rule-2 block

    // Tiles are assigned statically (wave w takes tiles w, w + 12, ..).  The three waves of a SIMD do not share its matrix pipe
    // evenly -- the oldest wins the arbitration: 534 k to 1.1 M cycles of wave life for the same 12 tiles -- but that is harmless:
    // the last wave of a SIMD runs alone at the full rate.  Handing tiles out from an LDS counter was measured and removed
    // (7-17 tiles per wave, 70 k instead of 60 k cycles per tile, OPQ step 30.9 vs 30.5 ms).
    const int ntile = (int)((wg_row1 - wg_row0 + 31) >> 5);
    int64_t row0 = wg_row0 + 32 * wave;
    if (row0 >= wg_row1) return;
    int cur_tile = wave;
    bool bad = false;                            // GATHER: a code >= K or a row index out of range was met
    // this lane's row of the tile at r0 (clamped to the last row): pointer to its half's 16 k, or -- GATHER -- the byte offset
    // of its code row in the code matrix (32-bit offsets against uniform bases keep the gather's addressing in one VGPR each)
    struct RowRef { const float* p; unsigned co; };
    auto row_ref = [&](int64_t r0) -> RowRef {
        const int64_t r = (r0 + j < n) ? r0 + j : n - 1;
        if constexpr (GATHER) {
            int64_t src = r;
            if (ga.sel_rows) {
                src = ga.sel_rows[r];
                if (src < 0 || src >= ga.n_codes) { bad = true; src = 0; }
            }
            return RowRef{nullptr, (unsigned)(src * ga.c_rs)};
        } else {
            return RowRef{x + r * x_rs + 16 * h, 0u};
        }
    };
    bool loads_on = true;
    // one 16-byte piece of a gathered row: floats k0 .. k0 + 3 of the reconstruction of the code row at byte offset co
    auto gather_piece = [&](unsigned co, int k0) -> f32x4 {
        const unsigned m = __umulhi((unsigned)k0, ga.inv_dsub);
        const unsigned off = (unsigned)k0 - m * (unsigned)ga.dsub;
        unsigned code = *(ga.codes + (co + m));                                   // (1-byte codes only: the host falls back otherwise)
        bad |= code >= (unsigned)ga.K;
        code = code < (unsigned)ga.K ? code : 0u;
        const unsigned boff = ((m * (unsigned)ga.K + code) * (unsigned)ga.dsub + off) * 4u;   // < 2^26: the codebook is at most 64 MB
        return *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(ga.cb) + boff);
    };
    auto load_burst = [&](f32x4 (&s)[4], const RowRef& rr, int bi, bool full) {   // burst bi of the tile whose row is rr
        if (!loads_on) { asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])); return; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k0 = 32 * bi + 16 * h + 4 * e;
            if (full || k0 < d) {
                if constexpr (GATHER) s[e] = gather_piece(rr.co, k0);
                else s[e] = *reinterpret_cast<const f32x4*>(rr.p + 32 * bi + 4 * e);
            } else {
                s[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto load_full = [&](f32x4 (&s)[4], const RowRef& rr, int bi) { load_burst(s, rr, bi, true); };
    auto load_tail = [&](f32x4 (&s)[4], const RowRef& rr) { load_burst(s, rr, nfull, false); };   // pieces past the row's end are zero
    // the tile in (t0, t1) leaves the accumulators as 16-byte row pieces: register 4 g + e of lane (row j, half h) is
    // column 32 ct + 8 g + 4 h + e
    auto store_tile = [&](const f32x16& t0, const f32x16& t1, int64_t r0) {
        if (ROT8_ABLATE == 3 || ROT8_ABLATE == 4) { asm volatile("" ::"v"(t0), "v"(t1)); return; }
        if (r0 + j < n) {
            float* orow = out + (r0 + j) * o_rs + col0 + 4 * h;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = col0 + 32 * ct + 8 * g + 4 * h;
                    const f32x16& c = ct ? t1 : t0;
                    const f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                    if (col < d) {
                        if (ROT8_STORE == 1) *reinterpret_cast<f32x4*>(orow + 32 * ct + 8 * g) = v;
                        else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(orow + 32 * ct + 8 * g));
                    }
                }
            }
        }
    };

    f32x4 sa[4], sb[4];
    RowRef prow = row_ref(row0);
    if (nfull > 0) load_full(sa, prow, 0); else load_tail(sa, prow);
    if (ROT8_ABLATE == 2 || ROT8_ABLATE == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sb[e] = sa[e];
        loads_on = false;
    }
    Rot8Ops ops;
    rot8_read(ops, plane);
    f32x16 t0 = zero, t1 = zero;                 // rule-2 block sums; between tiles: the finished tile on its way out
    bool pending = false;
    int64_t prev_row0 = 0;
    unsigned long long st_tiles = 0, st_k = 0, st_first = 0, st_last = 0;
    const unsigned long long st_t0 = stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (;;) {
        const unsigned long long st_a = stamps ? __builtin_amdgcn_s_memtime() : 0;
        const int next_tile = cur_tile + NWAVE;
        const bool has_next = next_tile < ntile;
        const int64_t next_row0 = wg_row0 + 32 * (int64_t)next_tile;
        const RowRef pnext = has_next ? row_ref(next_row0) : prow;
        f32x16 c0 = zero, c1 = zero;
        // One burst.  Order matters: the operands of THIS burst are formed first (the compiler's vmcnt(0) in front of
        // the swaps then waits for nothing younger), THEN burst bi + 1 (or the next tile's burst 0) is requested into
        // `nx` and has the 32 MFMAs of this burst to arrive; the previous tile's stores go out right behind the first
        // request of a tile, so the wait at the next burst finds them long done.
#define R8_STEP(cu, nx, bi, IS_TAIL)                                                                   \
        {                                                                                              \
            float xo_[16];                                                                             \
            rot8_swap(cu, xo_);                                                                        \
            if ((bi) + 1 < nfull) load_full(nx, prow, (bi) + 1);                                       \
            else if (TAIL && (bi) + 1 == nfull) load_tail(nx, prow);                                   \
            else if (has_next) { if (nfull > 0) load_full(nx, pnext, 0); else load_tail(nx, pnext); }  \
            if ((bi) == 0 && pending) store_tile(t0, t1, prev_row0);                                   \
            const float* pn_ = ((bi) + 1 < nb) ? plane + ((bi) + 1) * 8 * 256 : plane;                 \
            if (!(IS_TAIL)) rot8_burst<true>(plane + (bi) * 8 * 256, pn_, 8, xo_, c0, c1, ops);        \
            else rot8_burst<false>(plane + (bi) * 8 * 256, pn_, tail_groups, xo_, c0, c1, ops);        \
        }
#define R8_BLOCK(bi)                                                                                   \
        if (SPLITK && (bi) > 0 && ((bi) % KB) == 0) {                                                  \
            if ((bi) == KB) { t0 = c0; t1 = c1; }                                                      \
            else {                                                                                     \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) { t0[r] = fadd(t0[r], c0[r]); t1[r] = fadd(t1[r], c1[r]); } \
            }                                                                                          \
            c0 = zero; c1 = zero;                                                                      \
        }
        int bi = 0;
        for (; bi + 2 <= nfull; bi += 2) {
            R8_BLOCK(bi);
            R8_STEP(sa, sb, bi, false);
            R8_STEP(sb, sa, bi + 1, false);
        }
        if (ODD) {
            R8_BLOCK(bi);
            R8_STEP(sa, sb, bi, false);
            if (TAIL) {
                R8_STEP(sb, sa, bi + 1, true);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[e] = sb[e];
            }
        } else if (TAIL) {
            R8_BLOCK(bi);
            R8_STEP(sa, sb, bi, true);
#pragma unroll
            for (int e = 0; e < 4; ++e) sa[e] = sb[e];
        }
#undef R8_STEP
#undef R8_BLOCK
        if (SPLITK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t0[r] = fadd(t0[r], c0[r]); t1[r] = fadd(t1[r], c1[r]); }
        } else {
            t0 = c0; t1 = c1;
        }
        pending = true;
        prev_row0 = row0;
        prow = pnext;
        if (stamps) { asm volatile("" ::"v"(t0), "v"(t1)); const unsigned long long st_c = __builtin_amdgcn_s_memtime(); st_tiles += 1; st_k += st_c - st_a; if (st_tiles == 1) st_first = st_c - st_a; st_last = st_c - st_a; }
        if (!has_next) break;
        row0 = next_row0;
        cur_tile = next_tile;
    }
    if (pending) store_tile(t0, t1, prev_row0);
    if (GATHER && bad) atomicOr(ga.err, 1);
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * NWAVE + wave) * 8;
        o[0] = st_tiles; o[1] = st_k; o[2] = st_t0 - st_in;   // [2]: P staging + barrier
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[5] = st_r0; o[6] = st_first; o[7] = st_last;
    }
}

}  // namespace pqhip
