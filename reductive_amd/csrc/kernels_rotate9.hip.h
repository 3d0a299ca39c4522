// kernels_rotate9.hip.h -- OPQ rotation GEMM, ninth version: v8's data flow (P block stationary in LDS, x rows straight
// from global memory into the MFMA operand registers, the result tile leaving the accumulators as 16-byte row pieces)
// on v_mfma_f32_16x16x4_f32.
//
//   out[n][c] = sum_k x[n][k] * Pm[k][c]   (pq.rs:276 with Pm = projection, pq.rs:324 with Pm = projection^T),
//   rule-2 chains: one k-ordered fmaf chain per output element, restarted every 256 k, blocks added in order.
//
// Why (round 3, tools/rot8_ablate.sh variant 5 and tools/enc_power_ab.sh): with the same flop, registers and operands the
// 16x16x4 form of the matrix instruction holds a 3-4.5 % higher clock under the power cap than the 32x32x2 form, and 16-wide
// column tiles execute 304 columns for d = 300 where the 32-wide tiles of v8 execute 320.
//
// Layout.  Lane (i16 = lane & 15, q = lane >> 4).  A = 16 columns of the P block: lane supplies Pm[4 g + q][col i16] of
// k-group g; B = 16 rows of x: lane supplies x[row i16][4 g + q]; the lane receives out[row i16][4 q + v], v = 0..3 -- four
// consecutive columns of one row, one global_store_dwordx4.  A wave owns 32 rows (two row blocks) x the workgroup's column
// block (CT <= 4 tiles of 16 columns): per k-group one ds_read_b128 (the four tiles' A operands) feeds 2 x CT matrix
// instructions.  x arrives as one 16-byte load per lane and 16-k burst -- lane group q fetches floats [16 b + 4 q, + 4) of
// its row, the four groups together one 64-byte piece -- and a 4 x 4 transpose over the lane groups (two v_permlane16_swap,
// two v_permlane32_swap, in place) turns the four registers into the B operands of the burst's four k-groups.  Every burst
// executes all four k-groups: P's image and the x pieces are zero beyond d (at most 12 of 16 k, once per tile).
// The workgroups of the last column block execute only the tiles that hold real columns (CT = ceil((d - 64 (ncb - 1)) / 16)):
// the whole wave loop is instantiated per CT, so there is no join inside it.
#pragma once
#include "kernels_rotate8.hip.h"

namespace pqhip {

// the wave loop of one workgroup: CT column tiles of the staged P block
template <int CW, int CT, bool SPLITK, bool ODD, bool TAIL, bool GATHER, int NWAVE>
__device__ __forceinline__ void rot9_run(const float* __restrict__ x, int64_t n, int64_t x_rs, int d, float* __restrict__ out,
                                         int64_t o_rs, int64_t wg_row0, int64_t wg_row1, int col0, const float* pl, const Rot8Gather& ga,
                                         unsigned long long* stamps, unsigned long long st_in)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int nb = (d + 15) >> 4;                // 16-k bursts per tile (TAIL: the last one is partial)
    constexpr int KB = kKC / 16;                 // bursts per rule-2 block
    typedef float pav_t __attribute__((ext_vector_type(CW)));
    const float* plane = pl + lane * CW;         // + 64 CW floats per k-group: the block's CW tiles' A operands of this lane

    const int ntile = (int)((wg_row1 - wg_row0 + 31) >> 5);
    int cur_tile = wave;
    if (cur_tile >= ntile) return;
    int64_t row0 = wg_row0 + 32 * (int64_t)cur_tile;
    bool bad = false;                            // GATHER: a code >= K or a row index out of range was met
    // A row is addressed as a uniform base (the workgroup's first row; scalar registers, advanced per burst by the scalar
    // unit) plus a 32-bit byte offset per lane -- one vector add per load instead of 64-bit pointer arithmetic: every vector
    // instruction of this loop costs matrix issue time, and a 16-k burst is only 32 matrix instructions long.
    // GATHER: the offset is that of the lane's code row in the code matrix.
    const char* const xb = reinterpret_cast<const char*>(x + wg_row0 * x_rs);
    auto row_off = [&](int64_t r0, int rb) -> unsigned {
        const int64_t r = (r0 + 16 * rb + i16 < n) ? r0 + 16 * rb + i16 : n - 1;
        if constexpr (GATHER) {
            int64_t src = r;
            if (ga.sel_rows) {
                src = ga.sel_rows[r];
                if (src < 0 || src >= ga.n_codes) { bad = true; src = 0; }
            }
            return (unsigned)(src * ga.c_rs);
        } else {
            return (unsigned)((r - wg_row0) * x_rs * 4);
        }
    };
    // one 16-byte piece of a gathered row: floats k0 .. k0 + 3 of the reconstruction of the code row at byte offset co
    auto gather_piece = [&](unsigned co, unsigned k0) -> f32x4 {
        const unsigned m = __umulhi(k0, ga.inv_dsub);
        const unsigned off = k0 - m * (unsigned)ga.dsub;
        unsigned code = *(ga.codes + (co + m));
        bad |= code >= (unsigned)ga.K;
        code = code < (unsigned)ga.K ? code : 0u;
        const unsigned boff = ((m * (unsigned)ga.K + code) * (unsigned)ga.dsub + off) * 4u;   // < 2^26: the codebook is at most 64 MB
        return *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(ga.cb) + boff);
    };
    // Branch-free on purpose: a load behind a run-time choice makes its destination a phi, and the compiler then waits for it
    // (vmcnt(0)) and copies it at the join -- no prefetch left.  Lane group q fetches floats [16 b + 4 q, + 4) of its row;
    // in the partial last burst the groups past the row's end re-read its last piece (clamped) and are zeroed when used.
    const unsigned q16 = 16u * (unsigned)q, last_piece = 4u * (unsigned)d - 16u;
    auto load_burst = [&](f32x4 (&s)[2], unsigned o0, unsigned o1, int b) {
        unsigned kb = 64u * (unsigned)b + q16;                       // byte offset of the piece in the row
        if (TAIL) kb = kb < last_piece ? kb : last_piece;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned o = rb ? o1 : o0;
            if constexpr (GATHER) s[rb] = gather_piece(o, kb >> 2);
            else s[rb] = *reinterpret_cast<const f32x4*>(xb + (o + kb));
        }
    };
    // raw pieces -> B operands: xo[rb][s] of lane group q = float 16 b + 4 s + q of the row (a 4 x 4 transpose over the
    // lane groups, in place); MASK: the partial last burst, whose pieces past the row's end become zero
    auto transpose = [&](const f32x4 (&s)[2], bool mask, float (&xo)[2][4]) {
        const bool real = !mask || 16 * (nb - 1) + 4 * q < d;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = real ? __float_as_uint(s[rb][e]) : 0u;
            const auto a = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
            const auto c = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
            const auto e = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
            const auto o = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
            xo[rb][0] = __uint_as_float(e[0]);
            xo[rb][1] = __uint_as_float(o[0]);
            xo[rb][2] = __uint_as_float(e[1]);
            xo[rb][3] = __uint_as_float(o[1]);
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&](const f32x4 (&t)[2][CT], int64_t r0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int64_t r = r0 + 16 * rb + i16;
            if (r < n) {
                float* orow = out + r * o_rs + col0 + 4 * q;
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if (col0 + 16 * c + 4 * q < d) *reinterpret_cast<f32x4*>(orow + 16 * c) = t[rb][c];
            }
        }
    };

    f32x4 sa[2], sb[2];                          // the two bursts in flight (nb >= 2: the host's condition for this kernel)
    unsigned po0 = row_off(row0, 0), po1 = row_off(row0, 1);
    load_burst(sa, po0, po1, 0);
    load_burst(sb, po0, po1, 1);
    pav_t pa = *reinterpret_cast<const pav_t*>(plane);          // A operands of the k-group about to issue
    f32x4 t[2][CT];                                              // rule-2 block sums; between tiles: the finished tile on its way out
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < CT; ++c) t[rb][c] = zero4;
    bool pending = false;
    int64_t prev_row0 = 0;
    unsigned long long st_tiles = 0, st_k = 0, st_first = 0, st_last = 0;
    const unsigned long long st_t0 = stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (;;) {
        const unsigned long long st_a = stamps ? __builtin_amdgcn_s_memtime() : 0;
        const int next_tile = cur_tile + NWAVE;
        const bool has_next = next_tile < ntile;
        const int64_t next_row0 = wg_row0 + 32 * (int64_t)next_tile;
        unsigned pn0 = po0, pn1 = po1;
        if (has_next) { pn0 = row_off(next_row0, 0); pn1 = row_off(next_row0, 1); }
        f32x4 acc[2][CT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[rb][c] = zero4;
        // One burst.  The operands of THIS burst are formed first; its buffer is free then and takes burst b + 2 (or burst
        // 0 / 1 of the next tile; of this tile again when there is none), which has two bursts of matrix instructions to
        // arrive -- one burst (32 x 32 cycles x three waves per SIMD ~ 1.3 us) was not enough.  The previous tile's stores
        // go out behind the first request of a tile.
#define R9_STEP(cu, b, MASK)                                                                           \
        {                                                                                              \
            float xo_[2][4];                                                                           \
            transpose(cu, MASK, xo_);                                                                  \
            {                                                                                          \
                const bool wrap_ = (b) + 2 >= nb;                                                      \
                load_burst(cu, wrap_ ? pn0 : po0, wrap_ ? pn1 : po1, wrap_ ? (b) + 2 - nb : (b) + 2);  \
            }                                                                                          \
            if ((b) == 0 && pending) store_tile(t, prev_row0);                                         \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                            \
                const bool last_ = (g == 3) && ((b) + 1 == nb);                                        \
                const pav_t pn_ = *reinterpret_cast<const pav_t*>(last_ ? plane : plane + (4 * (b) + g + 1) * (64 * CW)); \
                __builtin_amdgcn_sched_barrier(0);                                                     \
                _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                       \
                    acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[c], xo_[0][g], acc[0][c], 0, 0, 0); \
                    acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[c], xo_[1][g], acc[1][c], 0, 0, 0); \
                }                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                     \
                pa = pn_;                                                                              \
            }                                                                                          \
        }
        // rule 2: the chains restart every KB bursts; the blocks are summed in order (the first one is taken as it is).
        // R9_SETTLE: the vector unit must not read an accumulator before the matrix instruction that writes it has retired
        // (8 passes).  The compiler's own wait-state insertion was one short for the LAST accumulator pair when the sums
        // follow the burst loop across a block boundary (observed: elements 2, 3 of acc[1][CT - 1] summed too early -- 8
        // wrong columns per row block, d = 272 / 336), so the wait is spelled out; it runs once per block, not per burst.
#define R9_SETTLE()                                                                                    \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                               \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) asm volatile("s_nop 15" : "+v"(acc[rb][c]));
#define R9_FOLD(first)                                                                                 \
        R9_SETTLE()                                                                                    \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                               \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                               \
            if (first) t[rb][c] = acc[rb][c];                                                          \
            else { _Pragma("unroll") for (int e = 0; e < 4; ++e) t[rb][c][e] = fadd(t[rb][c][e], acc[rb][c][e]); } \
            acc[rb][c] = zero4;                                                                        \
        }
        // The bursts of a tile: pairs (the two buffers alternate) up to np, where np leaves out the partial last burst
        // (TAIL) so that only that one pays for the zeroing selects; then the leftovers, each one straight code.
        // The block loop is the OUTER loop so that the sums are not if-converted into the burst loop.
        const int np = (TAIL ? nb - 1 : nb) & ~1;
        int b = 0;
        for (int blk = 0; b < np; ++blk) {
            if (SPLITK && blk > 0) { R9_FOLD(blk == 1); }
            const int bend = (SPLITK && b + KB < np) ? b + KB : np;
            for (; b < bend; b += 2) {
                R9_STEP(sa, b, false);
                R9_STEP(sb, b + 1, false);
            }
        }
        // leftovers: nb - np is 0 (no TAIL, even), 1 (ODD without TAIL: a full burst; TAIL with odd nb: the partial one) or
        // 2 (TAIL with even nb: a full burst, then the partial one)
        if (nb - np >= 1) {
            if (SPLITK && b > 0 && (b % KB) == 0) { R9_FOLD(b == KB); }
            R9_STEP(sa, b, (TAIL && ODD));
        }
        if (TAIL && !ODD) {
            R9_STEP(sb, b + 1, true);
        }
        if (ODD) {
            // an odd number of bursts leaves the next tile's burst 0 in sb and its burst 1 in sa
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { const f32x4 tmp = sa[rb]; sa[rb] = sb[rb]; sb[rb] = tmp; }
        }
        R9_SETTLE()
#undef R9_STEP
#undef R9_FOLD
#undef R9_SETTLE
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (SPLITK) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[rb][c][e] = fadd(t[rb][c][e], acc[rb][c][e]);
                } else {
                    t[rb][c] = acc[rb][c];
                }
            }
        pending = true;
        prev_row0 = row0;
        po0 = pn0; po1 = pn1;
        if (stamps) {
            const unsigned long long st_c = __builtin_amdgcn_s_memtime();
            st_tiles += 1; st_k += st_c - st_a; if (st_tiles == 1) st_first = st_c - st_a; st_last = st_c - st_a;
        }
        if (!has_next) break;
        row0 = next_row0;
        cur_tile = next_tile;
    }
    if (pending) store_tile(t, prev_row0);
    if (GATHER && bad) atomicOr(ga.err, 1);
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * NWAVE + wave) * 8;
        o[0] = st_tiles; o[1] = st_k; o[2] = st_t0 - st_in;   // [2]: P staging + barrier
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[5] = st_r0; o[6] = st_first; o[7] = st_last;
    }
}

// ODD: odd number of 16-k bursts, TAIL: d % 16 != 0 (compile-time, so that the burst ring of a tile is one straight code path).
// Launch geometry as v8: workgroup b -> XCD b % 8, column block (b / 8) % ncb, row group ((b / 8) / ncb) * 8 + xcd.
// CW: column tiles per workgroup block -- 4 (64 columns, P image 4 KB per 16 k: d <= 640) or 2 (32 columns, 2 KB per 16 k:
// d <= 1280, e.g. 768; each B operand then feeds half as many matrix instructions).
template <int CW, bool SPLITK, bool ODD, bool TAIL, bool GATHER>
__global__ __launch_bounds__(768, 3) void k_rotate_pblock9(const float* __restrict__ x, int64_t n, int64_t x_rs,
                                                           const float* __restrict__ Pm, int d, float* __restrict__ out,
                                                           int64_t o_rs, int rows_per_wg, int ncb, int64_t rg_per_xcd, Rot8Gather ga,
                                                           unsigned long long* stamps /* diagnostics: PQHIP_DEBUG_ROT_STAMP */)
{
    constexpr int NWAVE = 12;                    // also for the gather form: 157 registers, where v8's needed 221 and ran 8 waves
    constexpr int NT = 64 * NWAVE;
    extern __shared__ __attribute__((aligned(16))) float smem9[];
    float* pl = smem9;                           // [4 ceil(d/16) k-groups][64 lanes (col i16, k q)][CW column tiles]
    const unsigned long long st_in = stamps ? __builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x;

    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t qq = b >> 3;
    const int cb = (int)(qq % ncb);
    const int64_t rg_local = qq / ncb;
    const int64_t rg = rg_local * 8 + xcd;
    const int col0 = cb * 16 * CW;

    // stage the P block: Pm[k][col0 + 16 c + i16] -> image[(k >> 2) * 64 + (k & 3) * 16 + i16][c], zero beyond d in both directions
    {
        const int kpad = ((d + 15) >> 4) << 4;
        constexpr int Q = 4 * CW;                // float4 per P row of the block
        const int total = kpad * Q;
        for (int i0 = tid; i0 < total; i0 += NT * 2) {
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + NT * u;
                const int k = idx / Q, c = col0 + 4 * (idx % Q);
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < total && k < d && c < d) v[u] = *reinterpret_cast<const f32x4*>(Pm + (int64_t)k * d + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + NT * u;
                if (idx < total) {
                    const int k = idx / Q, c4 = idx % Q;
                    float* dst = pl + ((((k >> 2) << 6) + ((k & 3) << 4) + 4 * (c4 & 3)) * CW) + (c4 >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[CW * e] = v[u][e];
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
    const int ct = (d - col0 >= 16 * CW) ? CW : (d - col0 + 15) >> 4;   // column tiles with real columns (workgroup-uniform)
    if constexpr (CW == 4) {
        switch (ct) {
        case 4: rot9_run<4, 4, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in); break;
        case 3: rot9_run<4, 3, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in); break;
        case 2: rot9_run<4, 2, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in); break;
        default: rot9_run<4, 1, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in); break;
        }
    } else {
        if (ct == 2) rot9_run<2, 2, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in);
        else rot9_run<2, 1, SPLITK, ODD, TAIL, GATHER, NWAVE>(x, n, x_rs, d, out, o_rs, wg_row0, wg_row1, col0, pl, ga, stamps, st_in);
    }
}

}  // namespace pqhip
