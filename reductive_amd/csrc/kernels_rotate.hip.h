// kernels_rotate.hip.h -- OPQ rotation GEMM and the MFMA self-test (non-template kernels: include
// from exactly one translation unit, pqhip.hip).
#pragma once
#include <type_traits>
#include "kernels_mfma.hip.h"

#ifndef ROT_ABLATE
#define ROT_ABLATE 0      // timing experiments only (results wrong): 1 no global stores, 2 no epilogue at all, 3 no x fetch/stash, 4 no LDS operand reads after the first group
#elif ROT_ABLATE != 0 && !defined(PQHIP_TIMING_ONLY_BUILD)
#error "ROT_ABLATE produces wrong results: only `make TIMING=1` (libpqhip_timing.so, -DPQHIP_TIMING_ONLY_BUILD) may set it"
#endif
namespace pqhip {

// ---------------------------------------------------------------------------------------------
// K2/K4 v2  rotation GEMM with LDS-staged P slabs.
//   out[n][c] = sum_k x[n][k] * Pm[k][c], rule-2 chains (restart every 256 k, blocks summed).
// Workgroup = 4 waves = 64 rows x (2 * CT * 32) columns; wave (rg, ch) owns 32 rows x CT column
// tiles.  x is the A operand and comes straight from global memory (each lane streams its own
// row, 16 floats per slab, exactly like the encode kernel's B operand); the P slab [16 k][NC cols]
// is staged through registers into a double-buffered LDS image shared by the four waves and read
// back conflict-free (consecutive lanes = consecutive columns).  VALU work is ~8 selects per
// 8*CT MFMAs, so the FP32 pipe is spent almost entirely on the matrix instruction.
// SPLIT = d > 256 (kept as a template flag for the dispatcher; the k-block loop handles both).
// VEC   = 16-byte aligned rows and d % 4 == 0.
// ---------------------------------------------------------------------------------------------
template <int CT, bool SPLIT, bool VEC>
__global__ __launch_bounds__(256, 2) void k_rotate_gemm(const float* __restrict__ x, int64_t n,
                                                        int64_t x_rs, const float* __restrict__ Pm,
                                                        int d, float* __restrict__ out, int64_t o_rs)
{
    constexpr int KB = 16;            // k per slab (8 k-steps); 256 / KB slabs per rule-2 block
    constexpr int NC = 2 * CT * 32;   // columns per workgroup
    constexpr int NV = KB * NC / 4 / 256;  // float4 staged per thread per slab
    static_assert(KB * NC / 4 % 256 == 0, "slab must divide over the workgroup");
    __shared__ __attribute__((aligned(16))) float ps[2][KB][NC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int rg = wave >> 1, ch = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * 64 + rg * 32;
    const int col0 = blockIdx.y * NC;
    const int cbase = ch * CT * 32 + j;  // column (inside the slab) of this lane's tile 0

    int64_t arow = row0 + j;
    if (arow >= n) arow = n - 1;
    const float* xr = x + arow * x_rs;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 tot[CT], cur[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { tot[ct] = zero; cur[ct] = zero; }

    const int nslabs = (d + KB - 1) / KB;

    // -- staging helpers: P slab -> registers -> LDS; x slab -> registers.
    // Everything that depends only on (thread, i) is computed once: element offset of the
    // thread's i-th float4 inside a slab (global and LDS) and whether its columns exist.
    f32x4 pst[NV];
    int goff[NV], loff[NV], kk_[NV];
    bool cok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i;
        const int kk = idx / (NC / 4), c4 = idx % (NC / 4);
        kk_[i] = kk;
        goff[i] = kk * d + col0 + 4 * c4;
        loff[i] = kk * NC + 4 * c4;
        cok[i] = col0 + 4 * c4 < d;
    }
    const float* pslab = Pm;  // advances by KB rows per slab
    auto load_p = [&](int slab) {
        const int krem = d - slab * KB;  // rows of P left (wave-uniform)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kk_[i] < krem) {
                const float* p = pslab + goff[i];
                if (VEC) {
                    if (cok[i]) v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    const int c = goff[i] - kk_[i] * d;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < d) v[e] = p[e];
                }
            }
            pst[i] = v;
        }
        pslab += (int64_t)KB * d;
    };
    auto store_p = [&](int buf) {
        float* base = &ps[buf][0][0];
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(base + loff[i]) = pst[i];
    };
    // each lane fetches only the k = 2s + h elements of its row that it feeds to the MFMA
    auto load_x = [&](int slab, float (&xv)[KB / 2]) {
        const int kb = slab * KB + h;
#pragma unroll
        for (int s = 0; s < KB / 2; ++s) xv[s] = (kb + 2 * s < d) ? xr[kb + 2 * s] : 0.f;
    };

    float xv[KB / 2], xn[KB / 2];
    load_p(0);
    load_x(0, xv);
    store_p(0);
    __syncthreads();

    // rule 2: the k range is cut into blocks of kKC; inside a block every output element is one
    // fmaf chain (accumulated in `cur`), blocks are summed into `tot` with one rounded add each.
    // (Nested loops keep the slab loop free of accumulator copies.)
    constexpr int SPB = kKC / KB;  // slabs per block
    for (int sb = 0; sb < nslabs; sb += SPB) {
        const int se = (sb + SPB < nslabs) ? sb + SPB : nslabs;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) cur[ct] = zero;
        for (int slab = sb; slab < se; ++slab) {
            const int buf = slab & 1;
            const bool more = slab + 1 < nslabs;
            if (more) { load_p(slab + 1); load_x(slab + 1, xn); }
            // B fragments are fetched one k-step ahead (two register sets), so a ds_read's
            // latency sits behind the five MFMAs of the step before it instead of in front
            float bf[2][CT];
            {
                const float* prow = &ps[buf][h][cbase];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bf[0][ct] = prow[ct * 32];
            }
#pragma unroll
            for (int s = 0; s < KB / 2; ++s) {
                const float av = xv[s];
                if (s + 1 < KB / 2) {
                    const float* prow = &ps[buf][2 * (s + 1) + h][cbase];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bf[(s + 1) & 1][ct] = prow[ct * 32];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this step's MFMAs
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    cur[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[s & 1][ct], cur[ct], 0, 0, 0);
            }
            if (more) {
                store_p(buf ^ 1);
#pragma unroll
                for (int e = 0; e < KB / 2; ++e) xv[e] = xn[e];
            }
            __syncthreads();
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (sb == 0) tot[ct] = cur[ct];
            else
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[ct][r] = fadd(tot[ct][r], cur[ct][r]);
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = col0 + cbase + ct * 32;
        if (col < d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < n) __builtin_nontemporal_store(tot[ct][r], out + row * o_rs + col);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2/K4 v3  rotation GEMM, P-block stationary.
// A workgroup keeps a 64-column block of Pm for ALL k in LDS (d x 64 floats, 76.8 KB at d = 300) and
// streams 32-row tiles of x past it: per k-step a wave issues one ds_read2_b32 (both column
// tiles' B fragments), half a global_load_dwordx4 (its row's x, the A operand) and two MFMAs,
// so the FP32 pipe is spent almost entirely on the matrix instruction; the fixed per-tile work
// (rule-2 fold, 32 stores) is amortised over 300 MFMAs.  The column blocks of one row range are
// given to consecutive workgroups of ONE XCD so x is pulled from HBM once.
// Rule 2 (k-blocks of 256) is honoured with one extra accumulator pair.
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256, 2) void k_rotate_pblock(const float* __restrict__ x, int64_t n,
                                                          int64_t x_rs, const float* __restrict__ Pm,
                                                          int d, float* __restrict__ out, int64_t o_rs,
                                                          int rows_per_wg, int ncb, int64_t rg_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) float pl[];  // [kpad][64]
    const int kpad = (d + 3) & ~3;
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

    // stage the P block (zero padded beyond d in both directions).  LDS image: [k / 4][col][4] with the
    // four k of a group stored in the order (0, 2, 1, 3): lane half h then finds its two operands of
    // the group, k = 4q + h and k = 4q + 2 + h, as ONE 8-byte word at [q][col][2h] (one ds_read_b64 per
    // column tile and group, addressed by immediate offsets).
    for (int idx = tid; idx < kpad * 16; idx += 256) {
        const int k = idx >> 4, c4 = idx & 15, c = col0 + 4 * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < d) {
            const float* p = Pm + (int64_t)k * d + c;
            if (VEC) {
                if (c < d) v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < d) v[e] = p[e];
            }
        }
        const int inner = ((k & 1) << 1) | ((k >> 1) & 1);
        float* dst = pl + ((((k >> 2) << 6) + 4 * c4) << 2) + inner;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * e] = v[e];
    }
    __syncthreads();
    if (rg_local >= rg_per_xcd) return;
    const int64_t wg_row0 = rg * rows_per_wg;
    if (wg_row0 >= n) return;
    int64_t wg_row1 = wg_row0 + rows_per_wg;
    if (wg_row1 > n) wg_row1 = n;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* plane = pl + 4 * j + 2 * h;  // + q * 256 floats per group; + 128 for the second column tile
    const int nq = kpad / 4;                 // groups of 4 k (two k-steps)
    constexpr int QB = kKC / 4;              // groups per rule-2 block

    for (int64_t row0 = wg_row0 + 32 * wave; row0 < wg_row1; row0 += 128) {
        const int left = (int)((n - row0 < 32) ? n - row0 : 32);
        const float* xr = x + ((j < left) ? row0 + j : n - 1) * x_rs;
        auto load_x4 = [&](int qq) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(xr + 4 * qq);  // kpad == d when VEC (d % 4 == 0)
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * qq + e < d) v[e] = xr[4 * qq + e];
            }
            return v;
        };
        f32x16 tot0 = zero, tot1 = zero;
        for (int qb = 0; qb < nq; qb += QB) {
            const int qe = (qb + QB < nq) ? qb + QB : nq;
            f32x16 c0 = zero, c1 = zero;
            // One "group" = 4 k = two k-steps = four MFMAs.  x is streamed through an 8-deep register
            // ring (a group's 16 bytes are requested 32 MFMAs before they are used; loads unconditional
            // so the compiler counts vmcnt across the loop); B operands are read one group ahead as
            // 8-byte words.  Steady-state chunks carry no index clamps and address everything by
            // immediate offsets: per group 1 global load, 2 LDS reads, 2 lane-half selects, 4 MFMAs.
            constexpr int RD = 8;
            const int qlast = qe - 1;
            f32x4 ring[RD];
#pragma unroll
            for (int u = 0; u < RD; ++u) ring[u] = load_x4((qb + u < qlast) ? qb + u : qlast);
            f32x2 bA, bB;  // column tile 0 / 1: (k = 4q + h, k = 4q + 2 + h)
            auto read_b = [&](const float* pq, int u) {
                bA = *reinterpret_cast<const f32x2*>(pq + u * 256);
                bB = *reinterpret_cast<const f32x2*>(pq + u * 256 + 128);
            };
            read_b(plane + qb * 256, 0);
            int q0 = qb;
            // steady state: every refill index q0 + RD + u and every B prefetch q0 + u + 1 is in range
            for (; q0 + 2 * RD <= qe; q0 += RD) {
                const float* xq = xr + 4 * (q0 + RD);
                const float* pq = plane + q0 * 256;
#pragma unroll
                for (int u = 0; u < RD; ++u) {
                    const float a0 = sel_half(ring[u][0], ring[u][1]);
                    const float a1 = sel_half(ring[u][2], ring[u][3]);
                    const f32x2 b0 = bA, b1 = bB;
                    ring[u] = VEC ? *reinterpret_cast<const f32x4*>(xq + 4 * u) : load_x4(q0 + RD + u);
                    read_b(pq, u + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[0], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], c1, 0, 0, 0);
                }
            }
            // last full chunk(s): same body with clamped indices
            for (; q0 + RD <= qe; q0 += RD) {
#pragma unroll
                for (int u = 0; u < RD; ++u) {
                    const int qq = q0 + u;
                    const float a0 = sel_half(ring[u][0], ring[u][1]);
                    const float a1 = sel_half(ring[u][2], ring[u][3]);
                    const f32x2 b0 = bA, b1 = bB;
                    ring[u] = load_x4((qq + RD < qlast) ? qq + RD : qlast);
                    read_b(plane + ((qq + 1 < qlast) ? qq + 1 : qlast) * 256, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[0], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], c1, 0, 0, 0);
                }
            }
            // tail (< RD groups of this k-block): plain loads; bA/bB already hold group q0's operands
            for (int qq = q0; qq < qe; ++qq) {
                const f32x4 xa = load_x4(qq);
                const float a0 = sel_half(xa[0], xa[1]);
                const float a1 = sel_half(xa[2], xa[3]);
                const f32x2 b0 = bA, b1 = bB;
                read_b(plane + ((qq + 1 < qlast) ? qq + 1 : qlast) * 256, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[0], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[1], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], c1, 0, 0, 0);
            }
            if (qb == 0) { tot0 = c0; tot1 = c1; }
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot0[r] = fadd(tot0[r], c0[r]); tot1[r] = fadd(tot1[r], c1[r]); }
            }
        }
        const int cA = col0 + j, cB = col0 + 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (rr < left) {
                float* o = out + (row0 + rr) * o_rs;
                if (cA < d) __builtin_nontemporal_store(tot0[r], o + cA);
                if (cB < d) __builtin_nontemporal_store(tot1[r], o + cB);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2/K4 v5  rotation GEMM, P-block stationary, x staged through LDS in full 128-byte lines.
// Workgroup = 8 waves sharing one 64-column block of Pm in LDS; every wave owns 32-row tiles and a
// private double-buffered LDS slab [32 rows][32 k].  A slab is fetched with 4 global_load_dwordx4 per
// lane in which 8 consecutive lanes cover one row's 128 contiguous bytes (full lines, no reliance on
// L1 to merge row-strided 16-byte pieces), written to LDS with the four k of a group in the order
// (0, 2, 1, 3), and consumed as one ds_read_b64 per group: lane (row j, half h) gets k = 4q + h and
// k = 4q + 2 + h at once -- no lane-half selects, no barriers in the loop (slabs are wave-private).
// Per 4 MFMAs: 1 ds_read_b64 (A) + 1 ds_read2st64_b64 (B of both column tiles) + 1/2 global load +
// 1/2 ds_write_b128.  Requires 16-byte aligned rows and d % 4 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_rotate_pblock5(const float* __restrict__ x, int64_t n,
                                                           int64_t x_rs, const float* __restrict__ Pm,
                                                           int d, float* __restrict__ out, int64_t o_rs,
                                                           int rows_per_wg, int ncb, int64_t rg_per_xcd)
{
    constexpr int XS = 36;                       // slab row stride in floats (144 B: 16-B aligned, 2-way banks)
    extern __shared__ __attribute__((aligned(16))) float smem5[];
    const int kpad = (d + 31) & ~31;             // whole 32-k slabs (zero padded)
    float* pl = smem5;                           // [kpad / 4][64 cols][4]
    float* xs_all = smem5 + (size_t)kpad * 64;   // [8 waves][2][32][XS]
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

    for (int idx = tid; idx < kpad * 16; idx += 512) {
        const int k = idx >> 4, c4 = idx & 15, c = col0 + 4 * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < d && c < d) v = *reinterpret_cast<const f32x4*>(Pm + (int64_t)k * d + c);
        const int inner = ((k & 1) << 1) | ((k >> 1) & 1);
        float* dst = pl + ((((k >> 2) << 6) + 4 * c4) << 2) + inner;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * e] = v[e];
    }
    __syncthreads();
    if (rg_local >= rg_per_xcd) return;
    const int64_t wg_row0 = rg * rows_per_wg;
    if (wg_row0 >= n) return;
    int64_t wg_row1 = wg_row0 + rows_per_wg;
    if (wg_row1 > n) wg_row1 = n;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* xs = xs_all + (size_t)wave * 2 * 32 * XS;
    const float* plane = pl + 4 * j + 2 * h;     // + q * 256 floats per group; + 128: second column tile
    const int nslab = kpad / 32;
    const int tail_groups = (d - 32 * (nslab - 1) + 3) / 4;  // 4-k groups with real k in the last slab
    constexpr int SB = kKC / 32;                 // slabs per rule-2 block
    const int lr = lane >> 3, lc = lane & 7;     // staging role: rows lr + 8 i, 16-byte piece lc

    const float* rp[4];
    auto set_rows = [&](int64_t row0) {
        const int left = (int)((n - row0 < 32) ? n - row0 : 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 8 * i;
            rp[i] = x + ((r < left) ? row0 + r : n - 1) * x_rs + 4 * lc;
        }
    };
    f32x4 st[4];
    auto fetch = [&](int slab) {                 // 8 lanes x 16 B = one row's 128 contiguous bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 32 * slab + 4 * lc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < d) v = *reinterpret_cast<const f32x4*>(rp[i] + 32 * slab);
            st[i] = v;
        }
    };
    auto stash = [&](int buf) {                  // (k0, k1, k2, k3) -> (k0, k2, k1, k3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 w = {st[i][0], st[i][2], st[i][1], st[i][3]};
            *reinterpret_cast<f32x4*>(xs + ((size_t)buf * 32 + lr + 8 * i) * XS + 4 * lc) = w;
        }
    };

    int64_t row0 = wg_row0 + 32 * wave;
    if (row0 >= wg_row1) return;
    set_rows(row0);
    fetch(0);
    stash(0);
    const int ob = (nslab - 1) & 1;              // buffer of the last slab: free once the k loop is done
    for (; row0 < wg_row1; row0 += 256) {
        const int left = (int)((n - row0 < 32) ? n - row0 : 32);
        f32x16 tot0 = zero, tot1 = zero;
        for (int sb = 0; sb < nslab; sb += SB) {
            const int se = (sb + SB < nslab) ? sb + SB : nslab;
            f32x16 c0 = zero, c1 = zero;
            for (int slab = sb; slab < se; ++slab) {
                const int buf = slab & 1;
                const bool more = slab + 1 < nslab;
                if (more) fetch(slab + 1);
                const float* arow = xs + ((size_t)buf * 32 + j) * XS + 2 * h;
                const float* pq = plane + slab * 8 * 256;
                auto group = [&](int u) {        // 4 k of both column tiles
                    const f32x2 a = *reinterpret_cast<const f32x2*>(arow + 4 * u);
                    const f32x2 b0 = *reinterpret_cast<const f32x2*>(pq + u * 256);
                    const f32x2 b1 = *reinterpret_cast<const f32x2*>(pq + u * 256 + 128);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0[0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b1[0], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b0[1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1[1], c1, 0, 0, 0);
                };
                if (more || tail_groups == 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) group(u);
                } else {
                    // last slab of a k that is not a multiple of 32 (d = 300: 12 of 32): only the
                    // groups that hold real k (zero k-padding is exact, but it is not free)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < tail_groups) group(u);
                }
                if (more) stash(buf ^ 1);
            }
            if (sb == 0) { tot0 = c0; tot1 = c1; }
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot0[r] = fadd(tot0[r], c0[r]); tot1[r] = fadd(tot1[r], c1[r]); }
            }
        }
        // the next tile's first slab starts its trip from HBM before the stores of this one
        const bool has_next = row0 + 256 < wg_row1;
        if (has_next) { set_rows(row0 + 256); fetch(0); }
        // epilogue: the 32 x 64 result goes through the free slab buffer so that every lane stores 16
        // contiguous bytes (8 store instructions per tile instead of 64 dword stores: the dword tail
        // was store-issue bound, ~7,500 cycles per tile)
        float* os = xs + (size_t)ob * 32 * XS;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                os[((r & 3) + 8 * (r >> 2) + 4 * h) * XS + j] = ct ? tot1[r] : tot0[r];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = lr + 8 * i;
                const f32x4 v = *reinterpret_cast<const f32x4*>(os + rr * XS + 4 * lc);
                const int col = col0 + 32 * ct + 4 * lc;
                if (rr < left && col < d)
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (row0 + rr) * o_rs + col));
            }
        }
        if (has_next) stash(0);
    }
}

// ---------------------------------------------------------------------------------------------
// K2/K4 v6  rotation GEMM, P-block stationary, THREE waves per SIMD.
// Same data flow as v5 -- a 64-column block of Pm for all k in LDS, wave-private x slabs staged in the
// (k0, k2, k1, k3) order and consumed by ds_read_b64 -- with two changes that the in-kernel stamps of
// round 2 asked for (v5 at two waves per SIMD: k loop 48 k cycles per 32-row tile against 38.4 k of
// matrix issue, plus a 7-8 k cycle MFMA-free tile epilogue that two lock-stepped waves cannot hide):
//   * 12 waves per workgroup instead of 8.  The third wave per SIMD covers the other two's epilogues and
//     LDS / HBM waits.  To fit, a slab is 16 k deep ([32 rows][16 k], 2 x 2.5 KB per wave) and the P image
//     has exactly ceil(d / 4) groups: 76.8 + 61.4 KB of LDS at d = 300; <= 168 VGPRs.
//   * the LDS operands of group u + 1 are requested before the four MFMAs of group u are issued, across
//     slab boundaries too (two statically named operand sets);
// Stamps (PQHIP_DEBUG_ROT_STAMP, 10 M x 300, clock 2.24 GHz under this load): k loop 64 k + epilogue 8.5 k
// cycles per tile and wave where three waves sharing a SIMD need 57.6 k of matrix issue (79 %); 9 % of the
// launch lies outside the tile loops (P staging per workgroup, workgroup turnover, chunk tails).  Measured
// equal to v5 within 1 % (34.1-34.5 vs 34.3-34.8 ms for rotate + encode of 10 M rows); starting the three
// waves of a SIMD a third of a tile apart changed nothing.
// Timing ablations (-DROT_ABLATE=n, results wrong by construction; tools/rot_time.py, 1.18 M x 300 rows, one box):
// shipped 2.34 ms; no global stores 2.20; no x fetch from global memory 2.02 (-14 %: every row is fetched by the
// five column-block workgroups, 64 bytes = half a cache line per slab); no LDS operand reads after the first
// group 2.42 (the LDS reads are NOT what binds); a slab row stride without the 2-way bank conflict of the x
// operand reads (18 instead of 20 floats): no change.
// Requires 16-byte aligned rows and d % 4 == 0, like v5.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768, 3) void k_rotate_pblock6(const float* __restrict__ x, int64_t n,
                                                           int64_t x_rs, const float* __restrict__ Pm,
                                                           int d, float* __restrict__ out, int64_t o_rs,
                                                           int rows_per_wg, int ncb, int64_t rg_per_xcd,
                                                           unsigned long long* stamps /* diagnostics: PQHIP_DEBUG_ROT_STAMP */)
{
    constexpr int NWAVE = 12;
    constexpr int KS = 16;                       // k per slab (4 groups of 4)
    constexpr int XS = 20;                       // slab row stride in floats (80 B: 16-B aligned)
    constexpr int OS = 36;                       // row stride of the output staging image (both slab buffers: 2 x 640 >= 32 x 36 floats)
    extern __shared__ __attribute__((aligned(16))) float smem6[];
    const int ngroups = (d + 3) >> 2;
    float* pl = smem6;                           // [ngroups][64 cols][4]
    float* xs_all = smem6 + (size_t)ngroups * 256;   // [12 waves][2][32][XS]  (2 x 640 floats per wave)
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

    // stage the P block: 16-byte loads, two in flight per thread before the LDS stores
    {
        const int total = d * 16;                // float4 per block: d rows x 16
        for (int i0 = tid; i0 < total; i0 += 768 * 2) {
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + 768 * u;
                const int k = idx >> 4, c = col0 + 4 * (idx & 15);
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < total && c < d) v[u] = *reinterpret_cast<const f32x4*>(Pm + (int64_t)k * d + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + 768 * u;
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
    float* xs = xs_all + (size_t)wave * 2 * 32 * XS;
    const float* plane = pl + 4 * j + 2 * h;     // + q * 256 floats per group; + 128: second column tile
    const int nslab = (d + KS - 1) / KS;
    const int tail_groups = (d - KS * (nslab - 1) + 3) / 4;  // 4-k groups with real k in the last slab (1..4)
    constexpr int SB = kKC / KS;                 // slabs per rule-2 block
    const int lr = lane >> 2, lc = lane & 3;     // staging role: rows lr + 16 i (i = 0, 1), 16-byte piece lc

    const float* rp[2];
    auto set_rows = [&](int64_t row0) {
        const int left = (int)((n - row0 < 32) ? n - row0 : 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lr + 16 * i;
            rp[i] = x + ((r < left) ? row0 + r : n - 1) * x_rs + 4 * lc;
        }
    };
    f32x4 st[2];
    auto fetch = [&](int slab) {                 // 4 lanes x 16 B = 64 contiguous bytes of one row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = KS * slab + 4 * lc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < d && ROT_ABLATE != 3) v = *reinterpret_cast<const f32x4*>(rp[i] + KS * slab);
            st[i] = v;
        }
    };
    auto stash = [&](int buf) {                  // (k0, k1, k2, k3) -> (k0, k2, k1, k3)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 w = {st[i][0], st[i][2], st[i][1], st[i][3]};
            *reinterpret_cast<f32x4*>(xs + ((size_t)buf * 32 + lr + 16 * i) * XS + 4 * lc) = w;
        }
    };

    // k loop, straight-line per slab.  Operand registers ping-pong between two sets (A, B): the LDS reads of
    // group u + 1 are in flight while the four MFMAs of group u issue, ACROSS slab boundaries too -- slab
    // s + 1 is already in LDS when slab s starts (it was fetched during slab s - 1 and is stored first thing
    // in slab s), so the last group of a slab pre-reads the first group of the next one.  (The first build
    // of this kernel rotated one register set through copies inside a loop with run-time trip counts: the
    // compiler then waited lgkmcnt(0) in front of every group's MFMAs and the prefetch hid nothing.)
    const int nfull = (d % KS == 0) ? nslab : nslab - 1;     // slabs with all four groups
    auto xaddr = [&](int slab) { return xs + ((size_t)(slab & 1) * 32 + j) * XS + 2 * h; };
    f32x2 xa, pa0, pa1, xb, pb0, pb1;
#define PQ6_RD(X, P0, P1, SLAB, U)                                                  \
    if (ROT_ABLATE != 4 || ((SLAB) == 0 && (U) < 2)) {                              \
        const float* ar_ = xaddr(SLAB) + 4 * (U);                                   \
        const float* pq_ = plane + ((SLAB) * (KS / 4) + (U)) * 256;                 \
        X = *reinterpret_cast<const f32x2*>(ar_);                                   \
        P0 = *reinterpret_cast<const f32x2*>(pq_);                                  \
        P1 = *reinterpret_cast<const f32x2*>(pq_ + 128);                            \
    }
#define PQ6_MM(X, P0, P1)                                                           \
    {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[0], P0[0], c0, 0, 0, 0);        \
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[0], P1[0], c1, 0, 0, 0);        \
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[1], P0[1], c0, 0, 0, 0);        \
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[1], P1[1], c1, 0, 0, 0);        \
        __builtin_amdgcn_sched_barrier(0);                                          \
    }

    int64_t row0 = wg_row0 + 32 * wave;
    if (row0 >= wg_row1) return;
    set_rows(row0);
    fetch(0);
    stash(0);
    if (nslab > 1) fetch(1);                     // st: slab 1, stored in mid-slab 0
    unsigned long long st_tiles = 0, st_k = 0, st_e = 0;
    const unsigned long long st_t0 = stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (; row0 < wg_row1; row0 += 32 * NWAVE) {
        const unsigned long long st_a = stamps ? __builtin_amdgcn_s_memtime() : 0;
        const int left = (int)((n - row0 < 32) ? n - row0 : 32);
        f32x16 tot0 = zero, tot1 = zero;
        PQ6_RD(xa, pa0, pa1, 0, 0);
        for (int sb = 0; sb < nslab; sb += SB) {             // rule-2 blocks of 256 k
            const int se = (sb + SB < nslab) ? sb + SB : nslab;
            f32x16 c0 = zero, c1 = zero;
            for (int slab = sb; slab < se; ++slab) {
                if (slab < nfull) {
                    PQ6_RD(xb, pb0, pb1, slab, 1);
                    PQ6_MM(xa, pa0, pa1);
                    // slab + 1 (requested one slab ago) goes to LDS in mid-slab: nothing but the pre-read of
                    // the first group is outstanding at the loop head, and the store is long retired when the
                    // last group pre-reads the next slab
                    if (slab + 1 < nslab) stash((slab + 1) & 1);
                    if (slab + 2 < nslab) fetch(slab + 2);
                    PQ6_RD(xa, pa0, pa1, slab, 2);
                    PQ6_MM(xb, pb0, pb1);
                    PQ6_RD(xb, pb0, pb1, slab, 3);
                    PQ6_MM(xa, pa0, pa1);
                    if (slab + 1 < nslab) PQ6_RD(xa, pa0, pa1, slab + 1, 0);
                    PQ6_MM(xb, pb0, pb1);
                } else {                         // last, partial slab: 1 .. 3 groups (wave-uniform)
                    PQ6_RD(xb, pb0, pb1, slab, 1);   // (reads past the last real group stay inside LDS, unused)
                    PQ6_MM(xa, pa0, pa1);
                    if (tail_groups > 1) {
                        PQ6_RD(xa, pa0, pa1, slab, 2);
                        PQ6_MM(xb, pb0, pb1);
                    }
                    if (tail_groups > 2) PQ6_MM(xa, pa0, pa1);
                }
            }
            if (sb == 0) { tot0 = c0; tot1 = c1; }
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot0[r] = fadd(tot0[r], c0[r]); tot1[r] = fadd(tot1[r], c1[r]); }
            }
        }
        unsigned long long st_b = 0;
        if (stamps) { asm volatile("" ::"v"(tot0), "v"(tot1)); st_b = __builtin_amdgcn_s_memtime(); }
        // the next tile's first slab starts its trip from HBM before the stores of this one
        const bool has_next = row0 + 32 * NWAVE < wg_row1;
        if (has_next) { set_rows(row0 + 32 * NWAVE); fetch(0); }
        // epilogue: the 32 x 64 result goes through the wave's (now free) slab buffers so that every lane
        // stores 16 contiguous bytes.  (Leaving the tile in its registers and storing it as dwords during
        // the next k loop -- two 128-byte row segments per instruction, no extra registers -- was tried:
        // 64 predicated stores per tile made the k loop 30 % longer.)
        float* os = xs;
        const int er = lane >> 3, ec = lane & 7;     // rows er + 8 i, 16-byte piece ec of a 32-column tile
        if (ROT_ABLATE == 2) asm volatile("" ::"v"(tot0), "v"(tot1));
#pragma unroll
        for (int ct = 0; ct < 2 && ROT_ABLATE != 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                os[((r & 3) + 8 * (r >> 2) + 4 * h) * OS + j] = ct ? tot1[r] : tot0[r];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = er + 8 * i;
                const f32x4 v = *reinterpret_cast<const f32x4*>(os + rr * OS + 4 * ec);
                const int col = col0 + 32 * ct + 4 * ec;
                if (ROT_ABLATE == 1) { asm volatile("" ::"v"(v)); continue; }
                if (rr < left && col < d)
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (row0 + rr) * o_rs + col));
            }
        }
        if (has_next) { stash(0); if (nslab > 1) fetch(1); }
        if (stamps) { const unsigned long long st_c = __builtin_amdgcn_s_memtime(); st_tiles += 1; st_k += st_b - st_a; st_e += st_c - st_b; }
    }
#undef PQ6_RD
#undef PQ6_MM
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * NWAVE + wave) * 5;
        o[0] = st_tiles; o[1] = st_k; o[2] = st_e;
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

// ---------------------------------------------------------------------------------------------
// Device self-test of the MFMA == fmaf-chain property (pqhip_selftest_mfma_chain).
// One wave per trial: random A[32][k], B[k][32]; compares the MFMA tile with a scalar chain.
// ---------------------------------------------------------------------------------------------
__device__ inline float hash_unit(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    // value in (-4, 4) with a random exponent spread so that rounding really happens
    const float u = (float)(int)((x >> 40) & 0xFFFFFF) / 16777216.0f - 0.5f;
    const int e = (int)((x >> 8) & 7) - 3;
    return ldexpf(u, e);
}

__global__ void k_selftest_mfma_chain(int k, uint64_t seed, unsigned long long* mismatches)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const uint64_t base = seed + (uint64_t)blockIdx.x * 1000003ull;
    auto A = [&](int i, int kk) { return hash_unit(base * 31 + (uint64_t)i * 4099 + kk); };
    auto B = [&](int kk, int jj) { return hash_unit(base * 17 + (uint64_t)jj * 8209 + kk + 77777); };
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < k; k0 += 2) {
        const int kk = k0 + h;
        const float av = (kk < k) ? A(j, kk) : 0.f;
        const float bv = (kk < k) ? B(kk, j) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    unsigned long long bad = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float ref = 0.f;
        for (int kk = 0; kk < k; ++kk) ref = ffma(A(i, kk), B(kk, j), ref);
        if (__float_as_uint(ref) != __float_as_uint(acc[r])) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace pqhip
