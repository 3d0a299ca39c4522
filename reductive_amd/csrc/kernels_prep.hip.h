// kernels_prep.hip.h -- codebook preparation (runs once per pqhip_codebook_create and after every k-means update):
// squared centroid norms, the finite-norm flag, the MFMA fragment image, the transposed image of the small-codebook
// kernel and the block-diagonal pair fragments.  Non-template kernels: include from exactly one translation unit
// (pqhip_codebook.hip).
#pragma once
#include "common.hip.h"

namespace pqhip {

// ---------------------------------------------------------------------------------------------
// Codebook preparation (runs once per pqhip_codebook_create)
// ---------------------------------------------------------------------------------------------

// cc[m][j] = c_j . c_j (linalg.rs:168; rule 1).  Entries j in [K, k_pad) are +inf so that a
// padded centroid can never win the argmin.  One thread per (m, j).
__global__ void k_centroid_norms(const float* __restrict__ cb, int M, int K, int dsub, int k_pad,
                                 float* __restrict__ cc)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * k_pad) return;
    const int m = idx / k_pad, j = idx % k_pad;
    cc[idx] = (j < K) ? norm_unrolled_global(cb + ((int64_t)m * K + j) * dsub, dsub)
                      : __builtin_inff();
}

// MFMA A-operand image of the codebook for v_mfma_f32_32x32x2_f32:
//   frags[m][t][s][lane] = cb[m][32 t + (lane & 31)][2 s + (lane >> 5)]   (0 outside K / dsub)
// so that a wave fetches the fragment of (tile t, k-step s) with one coalesced dword load.
__global__ void k_build_frags(const float* __restrict__ cb, int M, int K, int dsub, int T, int S,
                              float* __restrict__ frags)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)M * T * S * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    int64_t r = idx >> 6;
    const int s = (int)(r % S); r /= S;
    const int t = (int)(r % T);
    const int m = (int)(r / T);
    const int j = 32 * t + (lane & 31);
    const int k = 2 * s + (lane >> 5);
    frags[idx] = (j < K && k < dsub) ? cb[((int64_t)m * K + j) * dsub + k] : 0.f;
}

// Transposed image for the small-codebook VALU kernel (kernels_smallk.hip.h):
// cbt[m][k][j] = cb[m][j][k] for j < K, 0 for K <= j < KP
__global__ void k_build_cbt(const float* __restrict__ cb, int M, int K, int dsub, int KP, float* __restrict__ cbt)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * dsub * KP) return;
    const int j = (int)(idx % KP);
    const int64_t r = idx / KP;
    const int k = (int)(r % dsub), m = (int)(r / dsub);
    cbt[idx] = (j < K) ? cb[((int64_t)m * K + j) * dsub + k] : 0.f;
}

// tells the fast paths whether every centroid norm is finite and far from overflow
__global__ void k_check_norms(const float* __restrict__ cc, int M, int K, int k_pad, float big,
                              int* __restrict__ flag_bad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * k_pad) return;
    const int j = idx % k_pad;
    if (j < K && !(cc[idx] < big)) atomicOr(flag_bad, 1);
}

// fragp[p][s][lane = (i, h)] = A[i][2 s + h] of pair p (see the header); ccp[p][hh][r] = ||c_{2p+hh}[r]||^2
__global__ void k_build_pair_frags(const float* __restrict__ cb, const float* __restrict__ cc, int M, int K, int dsub, int k_pad,
                                   float* __restrict__ fragp, float* __restrict__ ccp)
{
    const int NP = (M + 1) / 2;
    const int total = NP * dsub * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total + NP * 32; idx += gridDim.x * blockDim.x) {
        if (idx < total) {
            const int lane = idx & 63, s = (idx >> 6) % dsub, p = (idx >> 6) / dsub;
            const int i = lane & 31, h = lane >> 5;
            const int hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
            const int k = 2 * s + h, m = 2 * p + hh, kk = k - hh * dsub;
            float v = 0.f;
            if (m < M && r < K && kk >= 0 && kk < dsub) v = cb[((int64_t)m * K + r) * dsub + kk];
            fragp[idx] = v;
        } else {
            const int q = idx - total, p = q >> 5, hh = (q >> 4) & 1, r = q & 15, m = 2 * p + hh;
            ccp[q] = (m < M && r < K) ? cc[(int64_t)m * k_pad + r] : __builtin_inff();
        }
    }
}

}  // namespace pqhip
