// kernels_gather_cg.hip.h -- the reconstruct gather for short sub-vectors (1 or 2 floats, K <= 256, 1-byte codes): the
// centroids come from LDS, one GROUP of subquantizers per workgroup (round 4).
//
// Why: finalfusion's usual quantizer cuts d = 300 into M = 150 two-float sub-vectors; reconstruct_batch (primitives.rs:137-147,
// 169-172) is then 150 eight-byte copies per row out of a 307 KB codebook.  k_reconstruct fetches them through the vector
// memory path -- 64 different cache lines per instruction -- and ran at 0.24 of HBM (128 one-float sub-vectors: 0.17) against
// 0.71-0.83 for sub-vectors of 16+ floats.  Here a workgroup keeps the centroids of `mg` subquantizers in LDS (64 KB: 32
// two-float subquantizers at K = 256), writes that group's 16-byte chunks of every row of its row block, and the groups of a
// row block share an XCD (ids of equal residue mod 8) so that the code rows and the partially written lines of the output meet
// in one L2.  Per thread the chain code (global) -> centroid (LDS) -> store is software-pipelined: the codes of the next four
// chunks are requested before the current four are stored, in a branch-free loop body (a wait for a load also waits for every
// store issued before it; with a load -> wait -> store body the kernel ran at 3.3 TB/s where it now reaches 4.4-5.4).
// 10 M rows: d = 128, M = 64 2.91 -> 1.16 ms (0.25 -> 0.62 of HBM), d = 128, M = 128 4.83 -> 1.18 (0.17 -> 0.68), d = 20, M = 10,
// K = 128 0.214 -> 0.153 (0.53 -> 0.73), d = 300, M = 150 7.10 -> 4.6-5.0 (0.24 -> 0.34-0.36), d = 300, M = 300 12.7 -> 5.2-5.8.
// Rows whose pitch is not a multiple of 64 bytes (d = 300: 1,200) stay at 2.4-2.6 TB/s -- the 256-byte pieces of a group then
// start and end inside 64-byte blocks (d = 320, M = 160: 0.54 in the run where d = 300 gave 0.34); plain instead of
// non-temporal stores, 256 .. 4,096 rows per workgroup and two 156 KB groups of 1,024 threads all measured the same or worse.
// Four-float sub-vectors stay on k_reconstruct (one 16-byte gather per chunk: 4.29 ms at d = 300, M = 75 against 4.7-5.0 here).
// A code >= K raises *err and is clamped (reference: index_axis panic); the lookup form (SEL) reads row sel_rows[i] of a
// resident code matrix and multiplies by its scale exactly like k_reconstruct<.., SEL = true>.
#pragma once
#include "common.hip.h"
#include <type_traits>

namespace pqhip {

struct RecCgArgs {
    const uint8_t* codes;   // [n or n_codes][c_rs]
    int64_t n;              // output rows
    int64_t c_rs;
    float* out;             // [n][o_rs], 16-byte aligned, o_rs % 4 == 0
    int64_t o_rs;
    const float* cb;        // [M][K][DSUB]
    int M, K;
    int mg;                 // subquantizers per group (a multiple of 4 / DSUB)
    int n_groups;
    int rows_per_wg;
    int sb_rows;            // rows per inner block: sb_rows * (chunks per row of a group) < 2^16
    int64_t n_row_blocks;
    int* err;
    const int64_t* sel_rows;
    int64_t n_codes;
    const float* sel_scales;
    int64_t s_rs;
};

template <int DSUB, bool SEL>
__global__ __launch_bounds__(512) void k_reconstruct_cg(RecCgArgs a)
{
    static_assert(DSUB == 1 || DSUB == 2, "no such instantiation");
    constexpr int NT = 512;
    constexpr int CPC = 4 / DSUB;                                  // codes per 16-byte chunk
    extern __shared__ __attribute__((aligned(16))) float cg_s[];   // [nm][K][DSUB]
    const int64_t wg = blockIdx.x;
    const int64_t wl = wg >> 3;
    const int64_t rbl = wl / a.n_groups;
    const int grp = (int)(wl - rbl * a.n_groups);
    const int64_t row_block = rbl * 8 + (wg & 7);
    if (row_block >= a.n_row_blocks) return;
    const int m0 = grp * a.mg;
    const int nm = (a.M - m0 < a.mg) ? a.M - m0 : a.mg;
    {
        const float* src = a.cb + (int64_t)m0 * a.K * DSUB;
        const int total = nm * a.K * DSUB;
        if ((a.K * DSUB) % 4 == 0) {
            for (int i = threadIdx.x; i < total / 4; i += NT)
                reinterpret_cast<f32x4*>(cg_s)[i] = reinterpret_cast<const f32x4*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < total; i += NT) cg_s[i] = src[i];
        }
    }
    __syncthreads();
    const int cpr = nm / CPC;                                      // 16-byte chunks per row of this group
    const unsigned inv_cpr = (cpr == 1) ? 0u : (unsigned)(((1ull << 32) + cpr - 1) / cpr);
    const int64_t row0_wg = row_block * a.rows_per_wg;
    const int rows_wg = (a.n - row0_wg < a.rows_per_wg) ? (int)(a.n - row0_wg) : a.rows_per_wg;
    const bool wide_codes = (a.c_rs % CPC == 0) && (m0 % CPC == 0) && ((reinterpret_cast<uintptr_t>(a.codes) & (CPC - 1)) == 0);
    float* const out_g = a.out + (int64_t)m0 * DSUB;
    const uint8_t* const codes_g = a.codes + m0;
    bool bad = false;
    // The chain of a chunk is code (global) -> centroid (LDS) -> store, and a wait for a load also waits for every store issued
    // before it: the codes of the NEXT batch of U chunks per thread are requested before the stores of the current one, and the
    // loop body has no branch (address forms are chosen outside; indices past the end are clamped and their stores skipped).
    constexpr int U = 4;
    auto run = [&](auto wide_c, auto one_c) {
        constexpr bool WIDE = decltype(wide_c)::value, ONE = decltype(one_c)::value;
        struct Batch { unsigned cw[U]; int64_t orow[U]; int c[U]; float sc[U]; };
        for (int r0 = 0; r0 < rows_wg; r0 += a.sb_rows) {
            const int nch = ((rows_wg - r0 < a.sb_rows) ? rows_wg - r0 : a.sb_rows) * cpr;   // < 2^16: L / cpr == umulhi(L, inv_cpr)
            auto load_batch = [&](int base, Batch& b) {
                int64_t src_row[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int L = base + NT * u;
                    L = L < nch ? L : nch - 1;
                    const int row = ONE ? L : (int)__umulhi((unsigned)L, inv_cpr);
                    b.c[u] = L - row * cpr;
                    b.orow[u] = row0_wg + r0 + row;
                    src_row[u] = b.orow[u];
                    if constexpr (SEL) {
                        if (a.sel_rows) src_row[u] = a.sel_rows[b.orow[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    b.sc[u] = 1.0f;
                    if constexpr (SEL) {
                        if (a.sel_rows && (src_row[u] < 0 || src_row[u] >= a.n_codes)) { bad = true; src_row[u] = 0; }
                        if (a.sel_scales) b.sc[u] = a.sel_scales[src_row[u] * a.s_rs];
                    }
                    const uint8_t* cp = codes_g + src_row[u] * a.c_rs + b.c[u] * CPC;
                    if constexpr (CPC == 1) b.cw[u] = cp[0];
                    else if constexpr (WIDE) {
                        if constexpr (CPC == 2) b.cw[u] = *reinterpret_cast<const uint16_t*>(cp);
                        else b.cw[u] = *reinterpret_cast<const uint32_t*>(cp);
                    } else {
                        b.cw[u] = 0;
#pragma unroll
                        for (int i = 0; i < CPC; ++i) b.cw[u] |= (unsigned)cp[i] << (8 * i);
                    }
                }
            };
            Batch cur, nxt;
            load_batch(threadIdx.x, cur);                                  // (nch >= 1; indices are clamped)
            for (int base = threadIdx.x; base < nch; base += NT * U) {
                load_batch(base + NT * U, nxt);                           // unconditional: one path, exact wait counts
                f32x4 q[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int i = 0; i < CPC; ++i) {
                        unsigned code = (cur.cw[u] >> (8 * i)) & 255u;
                        if (code >= (unsigned)a.K) { bad = true; code = 0; }
                        const float* p = cg_s + ((size_t)(cur.c[u] * CPC + i) * a.K + code) * DSUB;
                        if constexpr (DSUB == 2) { const f32x2 t = *reinterpret_cast<const f32x2*>(p); q[u][2 * i] = t[0]; q[u][2 * i + 1] = t[1]; }
                        else q[u][i] = *p;
                    }
                    if (SEL && a.sel_scales) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) q[u][v] = fmul(q[u][v], cur.sc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + NT * u < nch)
                        __builtin_nontemporal_store(q[u], reinterpret_cast<f32x4*>(out_g + cur.orow[u] * a.o_rs + (int64_t)cur.c[u] * 4));
                cur = nxt;
            }
        }
    };
    if (cpr == 1) { if (wide_codes) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
    else { if (wide_codes) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
    if (bad) atomicOr(a.err, 1);
}

}  // namespace pqhip
