// kernels_opq_common.hip.h -- what the fused OPQ rotate -> encode kernels share: the argument block, the lane-half
// broadcast and the exact path for flagged rows.  (The first-generation kernel k_opq_encode_fused of round 2 lived here;
// the second generation, kernels_opq_fused2.hip.h, superseded it for every shape it is dispatched for and round 4
// removed it from the library.)
//
// Orientation shared by the fused kernels.  The rotation computes a 32-row x 64-column tile with P's fragment as the A
// operand and x as the B operand, so lane (row j, half h) holds, in register r of column tile t, the column in "slot"
// i = (r & 3) + 8 (r >> 2) + 4 h.  P's columns are permuted while they are staged so that slot i of tile t holds local
// column 32 t + 2 r + h: register r of lane (j, h) is rx[row j][k = 2 (16 t + r) + h] -- precisely the B operand of encode
// k-step S = 16 t + r of the v_mfma_f32_32x32x2_f32 distance chain (B[k = lane >> 5][j = lane & 31]).  A column block
// holds NM = 64 / dsub whole subquantizers (3 at dsub = 20: 60 of 64 slots used).
//
// Arithmetic is CANON-F32 throughout: the rotation chains are rule 2 (k-ordered fmaf chain, restart at k = 256, blocks
// added with one rounded add), ||rx_m||^2 is rule 1 evaluated across the two lane halves (v_permlane32_swap exchanges
// the partial sums, the adds keep ndarray's order), distances and the argmin are the LDS-atomic epilogue of
// k_encode_mfma_lds3.  Rows that need the exact path (NaN / Inf / huge norms, a negative fast minimum) are re-rotated by
// a scalar rule-2 chain and scanned exactly, so codes equal the oracle's for every input.
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

}  // namespace pqhip
