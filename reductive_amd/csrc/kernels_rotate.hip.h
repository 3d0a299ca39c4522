// kernels_rotate.hip.h -- OPQ rotation GEMM and the MFMA self-test (non-template kernels: include
// from exactly one translation unit, pqhip.hip).
#pragma once
#include "kernels_mfma.hip.h"

namespace pqhip {

// ---------------------------------------------------------------------------------------------
// K2/K4  rotation GEMM   out[n][c] = sum_k x[n][k] * Pm[k][c]     (pq.rs:276, pq.rs:324)
// with the matrixmultiply k-blocking of rule 2: chain(0..255) + chain(256..511) + ...
// Wave tile 32 rows x 64 columns (two 32x32 accumulators + two more for the current k-block),
// workgroup = 4 waves stacked on rows.  x is the A operand (row on the lane, k on the
// half-wave), Pm the B operand (coalesced 128-B rows).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_rotate_mfma(const float* __restrict__ x, int64_t n,
                                                        int64_t x_rs,
                                                        const float* __restrict__ Pm, int d,
                                                        float* __restrict__ out, int64_t o_rs)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (row0 >= n) return;
    const int c0 = blockIdx.y * 64;

    int64_t arow = row0 + j;
    if (arow >= n) arow = n - 1;
    const float* xr = x + arow * x_rs;
    const int cA = c0 + j, cB = c0 + 32 + j;
    const bool okA = cA < d, okB = cB < d;

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 totA = zero, totB = zero;
    for (int kb = 0; kb < d; kb += kKC) {
        const int ke = (kb + kKC < d) ? kb + kKC : d;
        f32x16 accA = zero, accB = zero;
        for (int k0 = kb; k0 < ke; k0 += 2) {
            const int k = k0 + h;
            const bool kok = k < ke;
            const float av = kok ? xr[k] : 0.f;
            const float* prow = Pm + (int64_t)(kok ? k : 0) * d;
            const float bA = (kok && okA) ? prow[cA] : 0.f;
            const float bB = (kok && okB) ? prow[cB] : 0.f;
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bA, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bB, accB, 0, 0, 0);
        }
        if (kb == 0) {
            totA = accA; totB = accB;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { totA[r] = fadd(totA[r], accA[r]); totB[r] = fadd(totB[r], accB[r]); }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < n) {
            if (okA) out[row * o_rs + cA] = totA[r];
            if (okB) out[row * o_rs + cB] = totB[r];
        }
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
