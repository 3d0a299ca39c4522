// kernels_rotate.hip.h -- the slab rotation GEMM (fallback of the P-block kernels v8 / v9: rows that are not
// 16-byte aligned, d % 4 != 0, d > 1,280) and the MFMA self-test (a non-template kernel: include from exactly one
// translation unit, pqhip_rotate.hip).  Rounds 1-3 also shipped the P-block generations v3 / v5 / v6 here; no
// dispatch path reached them once v8 / v9 existed and round 4 removed them (git history keeps them).
#pragma once
#include <type_traits>
#include "kernels_mfma.hip.h"

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
