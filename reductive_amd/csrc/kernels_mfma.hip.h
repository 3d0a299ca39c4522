// kernels_mfma.hip.h -- the gfx950 matrix-core kernels: PQ encode and the OPQ rotation GEMM.
//
// Why MFMA can be bit-exact here: on gfx950 v_mfma_f32_32x32x2_f32 computes, per output
// element, D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) -- a k-ordered f32 fmaf chain with one
// rounding per product and no wider internal accumulation.  Chaining the instruction over
// k-steps s = 0,1,.. with k = 2s + (lane >> 5) therefore reproduces CANON-F32 rule 2
// (matrixmultiply's sequential FMA micro-kernel) exactly.  pqhip_selftest_mfma_chain checks
// that property on the device; the parity tests check the whole kernel against the oracle.
//
// Layouts of v_mfma_f32_32x32x2_f32 (wave64):
//   A operand: lane l holds A[i = l & 31][k = l >> 5]          (one f32 VGPR)
//   B operand: lane l holds B[k = l >> 5][j = l & 31]          (one f32 VGPR)
//   C/D      : lane l, register r holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
#pragma once
#include "common.hip.h"

namespace pqhip {

// Above this squared norm (or for NaN/Inf) a tile leaves the fast epilogue: fma(-2, dp, t)
// equals fl(t - fl(dp + dp)) only while dp + dp cannot overflow; |dp| <= sqrt(xx * cc) keeps
// that true with a wide margin below 2^100.
constexpr float kBigNorm = 1.2676506e30f;  // 2^100
constexpr int kMfma16MaxTiles = 64;        // k_encode_mfma16: row tiles per wave (rows_per_item <= 2048), one bit each in a 64-bit mask

struct EncodeArgs {
    const float* x;      // [n][x_rs] rows, unit column stride
    int64_t n;
    int64_t x_rs;
    void* out;           // [n][o_rs] codes
    int64_t o_rs;
    const float* frags;  // [M][T][S][64]  (k_build_frags)
    const float* cc;     // [M][T*32]      (k_centroid_norms, +inf padded)
    const float* cb;     // [M][K][dsub]   row-major codebook (slow path only)
    int M, K, dsub;
    int k_pad;           // T * 32 (row length of cc)
    int rows_per_item;   // multiple of 32
    int64_t n_chunks;    // ceil(n / rows_per_item)
    int64_t chunks_per_xcd;  // ceil(n_chunks / 8)
    // K > 256 ("grouped" codebooks): every subquantizer is presented as `groups` virtual ones of
    // 256 centroids each (M above is then M_real * groups, K the real K); the kernel writes one
    // 64-bit key {ordered distance, global centroid index} per (row, virtual m) and k_merge_keys
    // takes the minimum over the groups.  groups == 1: plain codes.
    int groups;
    // optional device flag "some ||c||^2 is not finite or too large" (k_check_norms): when it is set
    // every row takes the exact path.  Lets a captured k-means iteration (hipGraph) stay correct
    // without the host looking at the flag between iterations.  nullptr: the host has checked.
    const int* bad_flag;
    // diagnostics only (PQHIP_DEBUG_ENC_STAMP): per wave {tiles, step-loop cycles, seam cycles, wave cycles, realtime ticks}
    unsigned long long* stamps = nullptr;
};

// Order-preserving map of an f32 distance onto u32 under ordered-float's total order
// (kmeans.rs:149-156): NaN greatest, -0 == +0.  Smaller key <=> smaller distance.
__device__ __forceinline__ unsigned ord_key(float d)
{
    if (d != d) return 0xffffffffu;
    if (d == 0.f) return 0x80000000u;
    const unsigned u = (unsigned)__float_as_int(d);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}

// Exact-by-construction evaluation of single rows, used when a row holds NaN/Inf/huge values or its
// fast minimum came out negative (a row that coincides with a centroid up to rounding -- every
// instance that is its own cluster, every first k-means iteration started from instances).  For
// each flagged row of the tile the WHOLE wave scans the K centroids (lane l takes l, l + 64, ..)
// with the literal three-operation distance, then reduces under the ordered-float total order with
// the lower index winning ties.  A flagged row costs a few microseconds, the other rows of the tile
// keep their fast result.  (An earlier version re-did the whole tile with one lane per row: 0.4 ms
// per affected tile, which made a 16 k-row k-means step ten times slower than it had to be.)
// Out of line and with everything passed BY VALUE: taking the address of the kernel-argument struct
// would force a copy of it into scratch memory in every wave (measured: 1.3 GB of spill writes per
// 10 M rows).  groups > 0 selects the 64-bit key output of grouped codebooks (mv = virtual m).
__device__ __forceinline__ bool of_equal(float a, float b) { return (a != a && b != b) || a == b; }

template <typename IdxT>
__device__ __noinline__ void encode_rows_slow_v(const float* x, int64_t x_rs, void* out, int64_t o_rs,
                                                const float* cb, const float* cc, int K, int dsub,
                                                int k_pad, int groups, int mv, int64_t row0, unsigned need)
{
    const int lane = threadIdx.x & 63;
    const int m = groups > 0 ? mv / groups : mv;
    const float* cbm = cb + (int64_t)m * K * dsub;
    const float* ccm = cc + (int64_t)m * k_pad;
    while (need) {  // wave-uniform
        const int jr = __builtin_ctz(need);
        need &= need - 1;
        const int64_t row = row0 + jr;
        const float* xs = x + row * x_rs + (int64_t)m * dsub;
        const float xx = norm_unrolled_global(xs, dsub);
        float bd = 0.f;
        int bj = 0x7fffffff;  // "no candidate yet"
        for (int j = lane; j < K; j += 64) {
            const float dp = chain_dot_global(xs, 1, cbm + (int64_t)j * dsub, 1, dsub);
            const float d = fsub(fadd(xx, ccm[j]), fadd(dp, dp));
            if (bj == 0x7fffffff || of_less(d, bd)) { bd = d; bj = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(bd, off);
            const int oj = __shfl_xor(bj, off);
            const bool take = oj != 0x7fffffff &&
                              (bj == 0x7fffffff || of_less(od, bd) || (of_equal(od, bd) && oj < bj));
            if (take) { bd = od; bj = oj; }
        }
        if (lane == 0) {
            if (groups > 0)
                reinterpret_cast<unsigned long long*>(out)[row * o_rs + mv] =
                    ((unsigned long long)ord_key(bd) << 32) | (unsigned long long)(unsigned)bj;
            else
                reinterpret_cast<IdxT*>(out)[row * o_rs + mv] = (IdxT)bj;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1  pq_encode  (primitives.rs:89-103 -> kmeans.rs:141-156 -> linalg.rs:167-176, fused)
//
// Codebook-stationary: one WAVE owns one (row-chunk, subquantizer m) item and keeps the whole
// sub-codebook of m -- T tiles of 32 centroids x S k-steps -- in T*S VGPRs as MFMA A operands
// for the lifetime of the item.  It then streams 32-row tiles of x: the 32 sub-vectors are the
// B operand (x row on the lane), so each lane ends up with the 16 x T distances of ITS row in
// its own accumulator registers and the argmin over centroids is lane-local; only the two
// half-waves (centroid rows +0..3 vs +4..7 of every group of 8) are merged by one shuffle.
// No LDS traffic for operands, no barriers in the loop; ||c||^2 is the only LDS resident.
//
// Item -> wave mapping is XCD-aware: workgroup b runs on XCD (b % 8) (observed round-robin;
// speed only), and all M items of one row chunk are given to consecutive waves of ONE XCD, so
// the M sub-vector slices of an x row are pulled from HBM once and then hit that XCD's L2.
//
// T  : centroid tiles (K padded to 32 T with +inf-norm dummies)
// DP : dsub padded to an even number (zero k-padding is exact: fma(0, 0, acc) == acc)
// VEC: x rows are 16-byte aligned and dsub % 4 == 0 -> global_load_dwordx4
// ---------------------------------------------------------------------------------------------
template <int T, int DP, bool VEC, typename IdxT>
__global__ __launch_bounds__(256, 2) void k_encode_mfma(EncodeArgs a)
{
    constexpr int S = DP / 2;
    __shared__ float cc_s[4][T * 32];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;

    // ---- item mapping (wave-uniform) ----
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t sidx = (b >> 3) * 4 + wave;  // position in this XCD's item stream
    const int64_t chunk_local = sidx / a.M;
    const int m = (int)(sidx - chunk_local * a.M);
    const int64_t chunk = chunk_local * 8 + xcd;
    const bool active = (chunk_local < a.chunks_per_xcd) && (chunk < a.n_chunks);

    // ---- resident operands ----
    float af[T][S];
    if (active) {
        const float* fp = a.frags + (int64_t)m * T * S * 64 + lane;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < S; ++s) af[t][s] = fp[(t * S + s) * 64];
        const float* ccm = a.cc + (int64_t)m * T * 32;
        for (int i = lane; i < T * 32; i += 64) cc_s[wave][i] = ccm[i];
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < S; ++s) af[t][s] = 0.f;
    }
    __syncthreads();  // every wave reaches this exactly once
    if (!active) return;

    const int64_t row_begin = chunk * a.rows_per_item;
    int64_t row_end = row_begin + a.rows_per_item;
    if (row_end > a.n) row_end = a.n;
    const float* xcol = a.x + (int64_t)m * a.dsub;
    const bool bad_codebook = a.bad_flag != nullptr && *a.bad_flag != 0;  // wave-uniform

    auto load_tile = [&](float (&v)[DP], int64_t tile_row0) {
        int64_t row = tile_row0 + j;
        if (row >= a.n) row = a.n - 1;  // clamp: loads stay in bounds, result is not stored
        const float* p = xcol + row * a.x_rs;
        // VEC: all DP floats are real (dsub == DP); otherwise dsub == DP - 1 and the last one is padding
        load_row_floats<VEC ? DP : DP - 1, DP>(p, v);
    };

    // operands of one 32-row tile: B fragments (k = 2s + h of the lane's row) and ||x||^2
    auto prep_tile = [&](const float (&v)[DP], float (&bop)[S], float& xx) {
        if (VEC) {
            xx = norm_unrolled_static<DP>(v);
        } else {  // dsub == DP - 1, known at compile time
            float w[DP - 1];
#pragma unroll
            for (int e = 0; e < DP - 1; ++e) w[e] = v[e];
            xx = norm_unrolled_static<DP - 1>(w);
        }
#pragma unroll
        for (int s = 0; s < S; ++s) bop[s] = h ? v[2 * s + 1] : v[2 * s];
    };
    auto chain = [&](int t, const float (&bop)[S]) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][s], bop[s], acc, 0, 0, 0);
        return acc;
    };

    // ---- software pipeline over steps (x tile i, centroid tile t) ----
    // While the VALU turns the 16 distances of step (i, t) into a running (min, argmin), the
    // matrix core already runs the fmaf chains of step (i, t+1) -- or of (i+1, 0) at the seam --
    // into a second accumulator set; sched_group_barrier pins the 1-MFMA : 8-VALU interleave.
    const int64_t last_tile0 = row_begin + ((row_end - row_begin - 1) / 32) * 32;
    float vn[DP];
    float bop[S];
    float xx;
    load_tile(vn, row_begin);
    prep_tile(vn, bop, xx);
    load_tile(vn, (row_begin + 32 <= last_tile0) ? row_begin + 32 : last_tile0);
    f32x16 acc = chain(0, bop);

    for (int64_t row0 = row_begin; row0 < row_end; row0 += 32) {
        float bop_n[S];
        float xx_n;
        prep_tile(vn, bop_n, xx_n);  // tile i+1
        load_tile(vn, (row0 + 64 <= last_tile0) ? row0 + 64 : last_tile0);  // tile i+2, in flight

        float best = __builtin_inff();
        int bidx = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x16 acc_n = (t + 1 < T) ? chain(t + 1, bop) : chain(0, bop_n);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // ||c||^2 of centroids 32t + 8g + 4h + {0,1,2,3}: one ds_read_b128
                const f32x4 c4 =
                    *reinterpret_cast<const f32x4*>(&cc_s[wave][32 * t + 8 * g + 4 * h]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float tt = fadd(xx, c4[q]);
                    // == fl(tt - fl(dp + dp)): 2*dp is exact and cannot overflow here
                    const float d = ffma(acc[4 * g + q], -2.0f, tt);
                    const bool lt = d < best;
                    best = lt ? d : best;
                    bidx = lt ? (32 * t + 8 * g + q) : bidx;
                }
            }
            // (the group-barrier solver is slow to compile: pinned only for the headline shape)
            if constexpr (T == 8 && DP == 20) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
                    if (s < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, (80 + S - 1) / S, 0);  // VALU share
                }
            }
            acc = acc_n;
        }
        bidx += 4 * h;
        // merge the two half-waves: lexicographic (distance, index) minimum
        const float od = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bidx, 32);
        if (od < best || (od == best && oi < bidx)) bidx = oi;

        const int64_t row = row0 + j;
        const bool valid = row < a.n;
        // NaN / Inf / huge rows: the fast epilogue's fma shortcut is not valid -> exact slow path
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && (bad_codebook || !(xx < kBigNorm)));
        const unsigned need = (unsigned)(bal | (bal >> 32));  // rows of this tile that need the exact path
        if (h == 0 && valid && !((need >> j) & 1u)) reinterpret_cast<IdxT*>(a.out)[row * a.o_rs + m] = (IdxT)bidx;
        if (need) encode_rows_slow_v<IdxT>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, a.dsub, a.k_pad, 0, m, row0, need);
#pragma unroll
        for (int s = 0; s < S; ++s) bop[s] = bop_n[s];
        xx = xx_n;
    }
}

}  // namespace pqhip
