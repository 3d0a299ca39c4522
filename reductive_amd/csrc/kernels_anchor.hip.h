// kernels_anchor.hip.h -- the scalar "anchor" encode kernel (exact by construction, any shape, any special value)
// and the key merge of grouped codebooks.  Templates only.
#pragma once
#include "common.hip.h"

namespace pqhip {


// ---------------------------------------------------------------------------------------------
// Scalar anchor encode: one thread per (row, subquantizer); literal CANON-F32 including the
// three-operation distance and the NaN-aware total order.  Any M, K, dsub.  This is the
// correctness anchor for the MFMA kernel and the fallback for shapes it does not cover.
// primitives.rs:89-103 -> kmeans.rs:141-156 -> linalg.rs:167-176
// ---------------------------------------------------------------------------------------------
__device__ inline int assign_scalar(const float* __restrict__ xs, const float* __restrict__ cbm,
                                    const float* __restrict__ ccm, int K, int dsub)
{
    const float xx = norm_unrolled_global(xs, dsub);
    int best = 0;
    float bestd = 0.f;
    for (int j = 0; j < K; ++j) {
        const float dp = chain_dot_global(xs, 1, cbm + (int64_t)j * dsub, 1, dsub);
        const float t = fadd(xx, ccm[j]);
        const float u = fadd(dp, dp);
        const float d = fsub(t, u);
        if (j == 0 || of_less(d, bestd)) { bestd = d; best = j; }
    }
    return best;
}

template <typename IdxT>
__global__ void k_encode_scalar(const float* __restrict__ x, int64_t n, int64_t x_rs,
                                IdxT* __restrict__ out, int64_t o_rs,
                                const float* __restrict__ cb, const float* __restrict__ cc, int M,
                                int K, int dsub, int k_pad)
{
    const int64_t total = n * M;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / M;
        const int m = (int)(idx % M);
        const int best = assign_scalar(x + row * x_rs + (int64_t)m * dsub,
                                       cb + (int64_t)m * K * dsub, cc + (int64_t)m * k_pad, K, dsub);
        out[row * o_rs + m] = (IdxT)best;
    }
}


// Grouped codebooks (K > 256): codes[row][m] = index part of the minimum over the groups of the
// 64-bit keys {ordered distance, global index} the MFMA kernel left per (row, virtual m).
template <typename IdxT>
__global__ __launch_bounds__(256) void k_merge_keys(const unsigned long long* __restrict__ keys, int64_t n,
                                                    int M, int groups, IdxT* __restrict__ codes, int64_t o_rs)
{
    const int64_t total = n * M;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / M;
        const int m = (int)(idx - row * M);
        const unsigned long long* k = keys + (row * M + m) * groups;
        unsigned long long best = k[0];
        for (int g = 1; g < groups; ++g) best = (k[g] < best) ? k[g] : best;
        codes[row * o_rs + m] = (IdxT)(unsigned)best;
    }
}

}  // namespace pqhip
