// kernels_atb.hip.h -- C = A^T . B over the rows (`instances.t().dot(&reconstructed)`, opq.rs:191) with rule-2
// arithmetic: per output element one fmaf chain over the rows of each 256-row block (matrixmultiply's KC), the
// block results added to C in block order with one rounded add each.
//
// Round 4 (the 23.9x traffic of the OPQ training step).  Rounds 1-3 gave every (row block,
// 64 x 64 macro tile) pair its own wave: a 256-row block of A and of B was fetched once per macro-tile row / column,
// 5 + 5 times at d = 300 (120 GB of fetches per 10 M rows), and B = the reconstructed matrix R had to exist in memory.
// Now ONE workgroup owns a whole row block and computes ALL of its output (up to 320 x 320 per workgroup; wider
// matrices are cut into 320-column blocks): eight waves, two per SIMD, each holding a 10 x 5 grid of 16 x 16
// accumulator tiles (200 of its 256 registers; v_mfma_f32_16x16x4_f32: the k = 0..3 chain on top of C, so 64 chained
// instructions are the block's 256-row chain).  A and B rows are read ONCE, 8 rows at a time, through a
// double-buffered LDS slab that all eight waves share (15 operand reads per 50 matrix instructions); the next slab's
// rows are in flight while the 6,400 matrix cycles of the current one issue.  (First build: four waves of 10 x 10 tiles
// and 4-row slabs -- 3,200 cycles of matrix work per slab did not cover an HBM round trip with one wave per SIMD: 55 ms
// per 10 M x 300 rows against 28 ms for the old kernel.)
// GATHER: the rows of B are not read but assembled from the codebook -- B[r] = concat_m cb[m][codes[r][m]] -- so R
// is never written (12 GB written and 12 GB read back per 10 M rows at d = 300 until round 3).
// Every part (= row range whose chain starts at +0) leaves one partial matrix; k_atb_fold adds them in part order.
// Exact mode: parts are the 256-row blocks of rule 2.  The float-tolerance mode (context option
// "cross_product_exact" = 0) makes a part many blocks long -- a plain split-K product, no per-block partials (16 GB
// written and read back per 10 M rows), within 1e-5 relative of the exact result but not bit-equal to it; every
// reductive build that can train OPQ links a BLAS (Cargo.toml:37-42), whose summation order is not
// matrixmultiply's either.
#pragma once
#include "common.hip.h"

namespace pqhip {

struct AtbArgs {
    const float* A;          // [n][a_rs], da columns used
    int64_t a_rs;
    int da;
    const float* B;          // [n][b_rs], db columns used (plain form)
    int64_t b_rs;
    int db;
    // gather form: B[r][c] = cb[m][codes[r][m]][c - m dsub], m = c / dsub
    const void* codes;       // [n][c_rs], 1- or 4-byte codes
    int64_t c_rs;
    const float* cb;         // [M][K][dsub]
    int K, dsub;
    unsigned inv_dsub;       // ceil(2^32 / dsub)
    int64_t n;
    int64_t row0;            // first row of this launch's first part
    int nparts;              // parts of this launch
    int64_t rows_per_part;   // multiple of 256
    int nba, nbb;            // 320-column output blocks per side: grid = nparts * nba * nbb
    int pa, pb;              // padded output dimensions (multiples of 16) = layout of a partial matrix
    float* part;             // [nparts][pa][pb]
};

constexpr int kAtbSR = 8;        // rows per slab = 2 k-groups (100 matrix instructions per wave between barriers)
constexpr int kAtbW = 320;       // output columns per workgroup and side
constexpr int kAtbXS = 336;      // slab row stride in floats: 336 = 16 (mod 64), the four k rows of a group hit disjoint banks
constexpr int kAtbNT = 10;       // 16 x 16 tile rows per wave (two wave rows)
constexpr int kAtbNTC = 5;       // 16 x 16 tile columns per wave (four wave columns)

template <bool GATHER, typename IdxT>
__global__ __launch_bounds__(512, 1) void k_atb_rowblock(AtbArgs a)
{
    __shared__ __attribute__((aligned(16))) float slab[2][2][kAtbSR][kAtbXS];   // [buffer][A | B][row][column]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;

    const int nblk = a.nba * a.nbb;
    const int p = blockIdx.x / nblk;
    const int ob = blockIdx.x - p * nblk;
    const int ia = ob / a.nbb, jb = ob - ia * a.nbb;
    const int ca0 = kAtbW * ia, cb0 = kAtbW * jb;                 // first output row / column of this workgroup
    const int wa = min(kAtbW, a.da - ca0), wb = min(kAtbW, a.db - cb0);   // real columns of A / B in this block
    const int64_t r_begin = a.row0 + (int64_t)p * a.rows_per_part;
    const int64_t r_end = min(a.n, r_begin + a.rows_per_part);

    // staging roles: 8 rows x 80 16-byte pieces per matrix = 640 pieces over 512 threads
    constexpr int NP = 2;
    int srow[NP], scol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = threadIdx.x + 512 * i;
        srow[i] = idx / 80;
        scol[i] = 4 * (idx - 80 * srow[i]);
    }
    // GATHER: the codes of a slab are requested one slab BEFORE its codebook rows (code -> codebook row is a dependent chain
    // of two memory round trips, and a slab lasts only ~2.8 us): cpre holds the codes of the slab that fetch() is about to use
    unsigned cpre[NP] = {};
    auto fetch_codes = [&](int64_t rs) {
        if constexpr (GATHER) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int64_t r = rs + srow[i];
                const int c = scol[i];
                unsigned code = 0;
                if (srow[i] < kAtbSR && r < r_end && c < wb) {
                    const unsigned m = __umulhi((unsigned)(cb0 + c), a.inv_dsub);
                    code = (unsigned)reinterpret_cast<const IdxT*>(a.codes)[r * a.c_rs + m];
                }
                cpre[i] = code;
            }
        }
    };
    auto fetch = [&](int64_t rs, f32x4 (&sa)[NP], f32x4 (&sb)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f};
            const int64_t r = rs + srow[i];
            if (srow[i] < kAtbSR && r < r_end) {
                const int c = scol[i];
                if (c < wa) {
                    const float* ar = a.A + r * a.a_rs + ca0 + c;
                    if (c + 4 <= wa) qa = *reinterpret_cast<const f32x4_u*>(ar);
                    else { for (int e = 0; e < 4; ++e) if (c + e < wa) qa[e] = ar[e]; }
                }
                if (c < wb) {
                    if constexpr (!GATHER) {
                        const float* br = a.B + r * a.b_rs + cb0 + c;
                        if (c + 4 <= wb) qb = *reinterpret_cast<const f32x4_u*>(br);
                        else { for (int e = 0; e < 4; ++e) if (c + e < wb) qb[e] = br[e]; }
                    } else {
                        // (dsub % 4 == 0 and 4-byte aligned codebook rows: a piece lies inside one sub-vector; the
                        // dispatcher takes the plain form otherwise)
                        const unsigned col = (unsigned)(cb0 + c);
                        const unsigned m = __umulhi(col, a.inv_dsub);
                        const unsigned off = col - m * (unsigned)a.dsub;
                        const unsigned code = cpre[i] < (unsigned)a.K ? cpre[i] : 0u;
                        const float* br = a.cb + ((int64_t)m * a.K + code) * a.dsub + off;
                        if (c + 4 <= wb) qb = *reinterpret_cast<const f32x4_u*>(br);
                        else { for (int e = 0; e < 4; ++e) if (c + e < wb) qb[e] = br[e]; }
                    }
                }
            }
            sa[i] = qa; sb[i] = qb;
        }
    };
    auto stash = [&](int buf, const f32x4 (&sa)[NP], const f32x4 (&sb)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (srow[i] < kAtbSR) {
                *reinterpret_cast<f32x4*>(&slab[buf][0][srow[i]][scol[i]]) = sa[i];
                *reinterpret_cast<f32x4*>(&slab[buf][1][srow[i]][scol[i]]) = sb[i];
            }
    };

    // this wave's tiles: rows [10 wr, 10 wr + nr) x columns [5 wc, 5 wc + nc) of the workgroup's 16 x 16 tile grid
    const int tiles_a = (wa + 15) >> 4, tiles_b = (wb + 15) >> 4;
    const int nr = max(0, min(kAtbNT, tiles_a - kAtbNT * wr)), nc = max(0, min(kAtbNTC, tiles_b - kAtbNTC * wc));   // wave-uniform
    f32x4 acc[kAtbNT][kAtbNTC];
#pragma unroll
    for (int r = 0; r < kAtbNT; ++r)
#pragma unroll
        for (int c = 0; c < kAtbNTC; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 sa[NP], sb[NP];
    fetch_codes(r_begin);
    fetch(r_begin, sa, sb);
    fetch_codes(r_begin + kAtbSR);
    stash(0, sa, sb);
    __syncthreads();
    int buf = 0;
    for (int64_t rs = r_begin; rs < r_end; rs += kAtbSR, buf ^= 1) {
        const bool more = rs + kAtbSR < r_end;                     // workgroup-uniform
        if (more) {
            fetch(rs + kAtbSR, sa, sb);                            // next slab in flight behind this slab's matrix instructions
            fetch_codes(rs + 2 * kAtbSR);                          // (rows past the end of the part: guarded inside)
        }
#pragma unroll
        for (int g = 0; g < kAtbSR / 4; ++g) {
            const float* ra = &slab[buf][0][4 * g + q][16 * kAtbNT * wr + i16];
            const float* rb = &slab[buf][1][4 * g + q][16 * kAtbNTC * wc + i16];
            // (no per-tile guards in here: a branch per matrix instruction starves the pipe.  Columns beyond nc get a
            // zero operand from the slab's zero padding -- the waves of a workgroup meet at the barrier after every
            // slab, so an edge wave that did fewer instructions would only wait)
            if (nr > 0 && nc > 0) {                                // wave-uniform
                float bo[kAtbNTC];
#pragma unroll
                for (int c = 0; c < kAtbNTC; ++c) bo[c] = rb[16 * c];
#pragma unroll
                for (int r = 0; r < kAtbNT; ++r) {
                    if (r < nr) {                                  // wave-uniform
                        const float ao = ra[16 * r];
#pragma unroll
                        for (int c = 0; c < kAtbNTC; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao, bo[c], acc[r][c], 0, 0, 0);
                    }
                }
            }
        }
        if (more) stash(buf ^ 1, sa, sb);
        __syncthreads();
    }
    // partial matrix of this part: lane (i16, q) holds D[4 q + v][i16] of every tile
    float* out = a.part + (int64_t)p * a.pa * a.pb;
#pragma unroll
    for (int r = 0; r < kAtbNT; ++r) {
        if (r < nr) {
#pragma unroll
            for (int c = 0; c < kAtbNTC; ++c) {
                if (c < nc) {
                    const int64_t row = ca0 + 16 * (kAtbNT * wr + r) + 4 * q, col = cb0 + 16 * (kAtbNTC * wc + c) + i16;
#pragma unroll
                    for (int v = 0; v < 4; ++v) out[(row + v) * a.pb + col] = acc[r][c][v];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_atb_fold(const float* __restrict__ part, int nparts, int64_t total,
                                                  int first, float* __restrict__ C)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float c = first ? part[idx] : C[idx];
#pragma unroll 8
    for (int b = first ? 1 : 0; b < nparts; ++b) c = fadd(c, part[(int64_t)b * total + idx]);
    C[idx] = c;
}

}  // namespace pqhip
