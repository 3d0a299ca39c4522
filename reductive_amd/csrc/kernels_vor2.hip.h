// kernels_vor2.hip.h -- PQ encode for 2-float (and 1-float) sub-vectors (K <= 256): only the centroids that can win in the point's grid cell
// are evaluated (round 4; the tables and the argument why the first minimum is always among them: vor2_prep.h).
//
// One LANE owns one row and walks the subquantizers of the workgroup's group; the group's tables -- grid
// parameters, 16-bit cell entries (list offset, length in words), the lists of centroid indices -- its centroids and their norms live in LDS.
// Per (row, m): the cell from two subtract-multiply pairs (exactly the operations the tables were built for; the grid parameters
// come through the scalar path), one LDS word for the cell, then the list four indices (one word) at a time: four independent
// 16-byte records {c0, c1, ||c||^2}, and per candidate, in ascending centroid order, the distance by the literal
// CANON-F32 operations -- xx by rule 1, dp = fma(x1, c1, fl(x0 c0)) (rule 2), fl(fl(xx + cc) - 2 dp) as fma(dp, -2, t)
// (no overflow: the tables exist only for |c| <= 2^40 and the grids end at 8.5 box sizes) -- and a strict `<` keeps the first
// minimum (kmeans.rs:149-156).  The candidate loop runs to the longest list of the wave.  Rows outside both grids (or NaN /
// Inf) take encode_rows_slow_v.  The reference's statistical test shape d = 20, M = 10, K = 128 (pq.rs:431-440) evaluates
// 3-9 of its 128 centroids per sub-vector this way.
#pragma once
#include "kernels_mfma.hip.h"
#include "vor2_prep.h"

namespace pqhip {

struct Vor2Args {
    const float* x;       // [n][x_rs]
    int64_t n;
    int64_t x_rs;
    uint8_t* out;         // [n][o_rs]
    int64_t o_rs;
    const float* cb;      // [M][K][2]
    const float* cc;      // [M][k_pad]
    const uint32_t* tab;  // the regions of vor2_prep.h, back to back
    const uint32_t* off;  // [M + 1] word offsets into tab
    int M, K, k_pad;
    int dsub;             // 1 or 2 floats per sub-vector (1: the second coordinate is 0 everywhere)
    int mg;               // subquantizers per workgroup
    int rows_per_thread;
    int n_groups;         // groups of mg subquantizers
    int64_t n_row_blocks; // blocks of (threads per workgroup) * rows_per_thread rows
};

// read-only for the whole launch: loads through the constant address space stay on the scalar path (see kernels_smallk.hip.h)
typedef const uint32_t __attribute__((address_space(4)))* vor2_cptr;

// MG: subquantizers per workgroup, at most (a.mg <= MG).  NT: threads per workgroup -- 256, or 512 when the row is cut into eight groups or
// more: a group then reads a short piece of a long row, the wait for those lines is what the waves of a CU have to cover, and
// eight waves share one copy of the tables (d = 300, M = 150, K = 256: 14.3 -> 11.1 ms per 10 M rows; 128 one-float
// sub-vectors: 8.2 -> 4.9 ms; with up to four groups 256 threads are 5-10 % faster)
template <int MG, int NT>
__global__ __launch_bounds__(NT) void k_encode_vor2(Vor2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t vor2_s[];
    const vor2_cptr off_c = (vor2_cptr)a.off;
    const vor2_cptr tab_c = (vor2_cptr)a.tab;
    // workgroup -> (row block, group), XCD-aware: workgroups are handed to the eight XCDs round robin, so the groups of one row
    // block are given ids of the same residue -- they run on one XCD at about the same time and its L2 fetches the rows' lines
    // once for all of them (a group reads only its 8 to 64 bytes of every row)
    const int64_t wg = blockIdx.x;
    const int64_t wl = wg >> 3;
    const int64_t rbl = wl / a.n_groups;
    const int grp = (int)(wl - rbl * a.n_groups);
    const int64_t row_block = rbl * 8 + (wg & 7);
    if (row_block >= a.n_row_blocks) return;
    const int m0 = grp * a.mg;
    const int nm = (a.M - m0 < a.mg) ? a.M - m0 : a.mg;
    const uint32_t w0 = off_c[m0];
    const uint32_t nw_raw = off_c[m0 + nm] - w0;
    const uint32_t nw = (nw_raw + 3u) & ~3u;                       // table words of the group (the records below start on 16 bytes)
    // LDS: [tables of the group][records: nm x K x {c0, c1, ||c||^2, -}]
    f32x4* const rec_s = reinterpret_cast<f32x4*>(vor2_s + nw);
    for (uint32_t i = threadIdx.x; i < nw_raw; i += NT) vor2_s[i] = a.tab[w0 + i];
    for (int i = threadIdx.x; i < nm * a.K; i += NT) {
        const int g = i / a.K, j = i - g * a.K;
        const float* c = a.cb + ((int64_t)(m0 + g) * a.K + j) * a.dsub;
        rec_s[i] = (f32x4){c[0], a.dsub == 2 ? c[1] : 0.f, a.cc[(int64_t)(m0 + g) * a.k_pad + j], 0.f};
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const bool word_stores = (a.o_rs % 4 == 0) && (m0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 3) == 0);
    const int64_t block_row0 = row_block * NT * a.rows_per_thread;
    // the row's floats of the group: requested one row ahead
    float xn[2 * MG];
    const int nfl = a.dsub * nm;                                   // floats of the group per row
    // 16-byte loads when the group's floats start on 16 bytes in every row (row stride, base and first column)
    const bool vec4 = (a.x_rs % 4 == 0) && ((a.dsub * m0) % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    auto fetch = [&](int it) {
        int64_t row = block_row0 + (int64_t)it * NT + threadIdx.x;
        if (row >= a.n) row = a.n - 1;
        const float* xr = a.x + row * a.x_rs + a.dsub * m0;
        if (vec4) {                                                // (wave-uniform; whole pieces: the row has them, x_rs >= d)
#pragma unroll
            for (int e = 0; e < 2 * MG; e += 4) {
                if (e < nfl) {
                    const int64_t left = a.x_rs - (int64_t)a.dsub * m0 - e;   // floats of the row from here on: >= 1
                    if (left >= 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + e);
                        xn[e] = v[0]; xn[e + 1] = v[1]; xn[e + 2] = v[2]; xn[e + 3] = v[3];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) xn[e + k] = (k < left) ? xr[e + k] : 0.f;
                    }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2 * MG; ++e)
                if (e < nfl) xn[e] = xr[e];
        }
    };
    fetch(0);
    for (int it = 0; it < a.rows_per_thread; ++it) {
        const int64_t wave_row0 = block_row0 + (int64_t)it * NT + (threadIdx.x & ~63);
        if (wave_row0 >= a.n) break;                               // wave-uniform
        const int64_t row = wave_row0 + lane;
        const bool valid = row < a.n;
        float xs[2 * MG];
#pragma unroll
        for (int e = 0; e < 2 * MG; ++e) xs[e] = xn[e];
        fetch((it + 1 < a.rows_per_thread) ? it + 1 : it);
        uint8_t* orow = a.out + row * a.o_rs + m0;
        unsigned codes_lo = 0, codes_hi = 0, bad = 0;                // the group's code bytes of this row; bit g of bad: exact path
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            if (g >= nm) break;                                    // wave-uniform
            // grid parameters: uniform, read from the global copy of the header through the scalar path
            const uint32_t hoff = off_c[m0 + g];
            const vor2_cptr hdr = tab_c + hoff;
            const uint32_t r0 = hoff - w0;                         // the region's first word in LDS
            const float x0 = a.dsub == 2 ? xs[2 * g] : xs[g], x1 = a.dsub == 2 ? xs[2 * g + 1] : 0.f;
            // cell: fine grid, else coarse grid, else the exact path.  (Every header word is fetched as a scalar BEFORE the
            // lane-dependent choice: a load of a selected address would be a vector load whose latency nothing hides.)
            const float t0 = fmul(fsub(x0, __uint_as_float(hdr[0])), __uint_as_float(hdr[1]));
            const float t1 = fmul(fsub(x1, __uint_as_float(hdr[2])), __uint_as_float(hdr[3]));
            const float gf0 = __uint_as_float(hdr[4]), gf1 = __uint_as_float(hdr[16]);
            const int base_f = (int)hdr[10], G = (int)hdr[13], sub_base = (int)hdr[15];
            const uint32_t list_w = hdr[12] >> 2;
            const bool in_f = (t0 >= 0.f) & (t0 < gf0) & (t1 >= 0.f) & (t1 < gf1);
            bool in_any = in_f;
            int ci = base_f + (in_f ? (int)t0 * G + (int)t1 : 0);
            if (__builtin_amdgcn_ballot_w64(!in_f) != 0) {                      // some lane of the wave is outside the fine grid
                const float u0 = fmul(fsub(x0, __uint_as_float(hdr[5])), __uint_as_float(hdr[6]));
                const float u1 = fmul(fsub(x1, __uint_as_float(hdr[7])), __uint_as_float(hdr[8]));
                const float cgf0 = __uint_as_float(hdr[9]), cgf1 = __uint_as_float(hdr[17]);
                const int base_c = (int)hdr[11], CG = (int)hdr[14];
                const bool in_c = (u0 >= 0.f) & (u0 < cgf0) & (u1 >= 0.f) & (u1 < cgf1);
                if (!in_f && in_c) ci = base_c + (int)u0 * CG + (int)u1;
                in_any = in_f || in_c;
            }
            const uint16_t* cells = reinterpret_cast<const uint16_t*>(vor2_s + r0);
            uint32_t cw = cells[ci];
            const bool split = in_f && (cw & 15u) == 15u;              // a dense fine cell: one of its four half cells
            if (__builtin_amdgcn_ballot_w64(split) != 0) {
                const float f0 = fsub(t0, (float)(int)t0), f1 = fsub(t1, (float)(int)t1);    // exact
                const int si = sub_base + 4 * (int)(cw >> 4) + 2 * (f0 >= 0.5f ? 1 : 0) + (f1 >= 0.5f ? 1 : 0);
                const uint32_t cw2 = cells[split ? si : 0];
                if (split) cw = cw2;
            }
            const int cnt = in_any ? (int)(cw & 15u) + 1 : 0;       // words of the list
            const uint32_t* lp = vor2_s + r0 + list_w + (cw >> 4);    // (lists start on words)
            const f32x4* rec = rec_s + (size_t)g * a.K;
            const float xx = fadd(fmul(x0, x0), fmul(x1, x1));                 // rule 1, two elements
            float best = __builtin_inff();
            int bj = -1;
            for (int i = 0; __builtin_amdgcn_ballot_w64(i < cnt) != 0; ++i) {      // to the longest list of the wave, a word (four indices) at a time
                const uint32_t four = (i < cnt) ? lp[i] : 0u;
                f32x4 r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = rec[(four >> (8 * e)) & 255u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dp = ffma(x1, r[e][1], fmul(x0, r[e][0]));     // rule 2: k = 0, 1 from +0
                    const float d = ffma(dp, -2.0f, fadd(xx, r[e][2]));        // fl(fl(xx + cc) - fl(dp + dp))
                    // Ascending index inside the list: the first minimum stays.  No test of i + e against the list length: the
                    // padding of a list's last word repeats its last index (never `<`), and a lane whose list is exhausted
                    // evaluates centroid 0 -- a real centroid that, not being on the list, is strictly farther than the winner.
                    if (d < best) { best = d; bj = (int)((four >> (8 * e)) & 255u); }
                }
            }
            if (cnt == 0) bj = -1;                                             // outside both grids / NaN / Inf: the exact path
            const unsigned long long need = __builtin_amdgcn_ballot_w64(valid && bj < 0);
            if (bj < 0) bad |= 1u << g;
            else if (g < 4) codes_lo |= (unsigned)bj << (8 * g);
            else codes_hi |= (unsigned)bj << (8 * (g - 4));
            if (need) {                                                       // wave-uniform
                const unsigned l32 = (unsigned)need, h32 = (unsigned)(need >> 32);
                if (l32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, a.dsub, a.k_pad, 0, m0 + g, wave_row0, l32);
                if (h32) encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, a.dsub, a.k_pad, 0, m0 + g, wave_row0 + 32, h32);
            }
        }
        // the row's codes of the group: whole words where the code matrix allows it (one byte store per (row, m) is 64 cache
        // lines per instruction and a partial write each), bytes otherwise and for rows the exact path has written to
        if (valid) {
            if (word_stores && bad == 0 && (nm & 3) == 0) {
                *reinterpret_cast<unsigned*>(orow) = codes_lo;
                if (nm == 8) *reinterpret_cast<unsigned*>(orow + 4) = codes_hi;
            } else {
#pragma unroll
                for (int g = 0; g < MG; ++g)
                    if (g < nm && !((bad >> g) & 1u)) orow[g] = (uint8_t)((g < 4 ? codes_lo : codes_hi) >> (8 * (g & 3)));
            }
        }
    }
}

}  // namespace pqhip
