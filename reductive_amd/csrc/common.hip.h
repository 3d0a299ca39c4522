// common.hip.h -- device helpers shared by the gfx950 kernels of libpqhip.
//
// CANON-F32 distinguishes fused from unfused arithmetic, so this whole
// library is compiled with -ffp-contract=off and the helpers below spell every rounding out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma STDC FP_CONTRACT OFF

namespace pqhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKC = 256;  // matrixmultiply sgemm k-block: chains restart every 256 k (rule 2)

__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float ffma(float a, float b, float c) { return __fmaf_rn(a, b, c); }

// ordered-float 2 total order used by kmeans.rs:149-156: NaN greatest, NaN == NaN, -0 == +0.
__device__ __forceinline__ bool of_less(float a, float b)
{
    if (a != a) return false;
    if (b != b) return true;
    return a < b;
}

// ndarray numeric_util::unrolled_dot(x, x) for a strided global vector (rule 1), runtime length.
__device__ inline float norm_unrolled_global(const float* __restrict__ x, int n)
{
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; n - i >= 8; i += 8) {
#pragma unroll
        for (int l = 0; l < 8; ++l) p[l] = fadd(p[l], fmul(x[i + l], x[i + l]));
    }
    float s = 0.f;
    s = fadd(s, fadd(p[0], p[4]));
    s = fadd(s, fadd(p[1], p[5]));
    s = fadd(s, fadd(p[2], p[6]));
    s = fadd(s, fadd(p[3], p[7]));
    for (; i < n; ++i) s = fadd(s, fmul(x[i], x[i]));
    return s;
}

// same, for a register-resident vector of compile-time length D.
template <int D>
__device__ __forceinline__ float norm_unrolled_static(const float (&v)[D])
{
    constexpr int NF = (D / 8) * 8;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NF; ++e) p[e & 7] = fadd(p[e & 7], fmul(v[e], v[e]));
    float s = 0.f;
    s = fadd(s, fadd(p[0], p[4]));
    s = fadd(s, fadd(p[1], p[5]));
    s = fadd(s, fadd(p[2], p[6]));
    s = fadd(s, fadd(p[3], p[7]));
#pragma unroll
    for (int e = NF; e < D; ++e) s = fadd(s, fmul(v[e], v[e]));
    return s;
}

// register-resident vector padded to DP, logical length n <= DP (wave-uniform).
template <int DP>
__device__ __forceinline__ float norm_unrolled_padded(const float (&v)[DP], int n)
{
    if (n == DP) return norm_unrolled_static<DP>(v);
    const int nf = (n / 8) * 8;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < DP; ++e)
        if (e < nf) p[e & 7] = fadd(p[e & 7], fmul(v[e], v[e]));
    float s = 0.f;
    s = fadd(s, fadd(p[0], p[4]));
    s = fadd(s, fadd(p[1], p[5]));
    s = fadd(s, fadd(p[2], p[6]));
    s = fadd(s, fadd(p[3], p[7]));
#pragma unroll
    for (int e = 0; e < DP; ++e)
        if (e >= nf && e < n) s = fadd(s, fmul(v[e], v[e]));
    return s;
}


typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 r;  // two independent IEEE binary32 adds
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b)
{
    f32x2 r;  // two independent IEEE binary32 multiplies
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


// lane-half select: lanes 32..63 take `hi`, lanes 0..31 take `lo`.  The two values are first made
// opaque so that the optimiser cannot turn selects on vector elements into a runtime-indexed
// extract (which it lowers to a compare/select chain over the whole vector); the select itself is
// left to the compiler so that it also inserts the VALU -> MFMA operand wait states.
__device__ __forceinline__ float sel_half(float lo, float hi)
{
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (threadIdx.x & 32) ? hi : lo;
}

// rule 1 (ndarray unrolled_dot(x, x)) for a register vector of compile-time length D held as D/2
// pairs, with packed arithmetic: identical roundings, about half the instructions of the scalar form.
template <int D>
__device__ __forceinline__ float norm_unrolled_packed(const f32x2 (&v2)[D / 2])
{
    static_assert(D % 2 == 0, "pairs");
    constexpr int C = D / 8;          // full chunks of 8
    constexpr int NF = C * 8;
    f32x2 sq[D / 2];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) sq[i] = pk_mul(v2[i], v2[i]);
    float s = 0.f;
    if (C > 0) {
        f32x2 pp[4];  // pp[l2] = (p[2 l2], p[2 l2 + 1]);  p[l] = 0 + sq[l] (+ sq[8 + l] ...)
#pragma unroll
        for (int l2 = 0; l2 < 4; ++l2) {
            pp[l2] = sq[l2];  // 0 + x == x exactly for x >= +0 or NaN
#pragma unroll
            for (int c = 1; c < C; ++c) pp[l2] = pk_add(pp[l2], sq[4 * c + l2]);
        }
        const f32x2 a = pk_add(pp[0], pp[2]);  // (p0 + p4, p1 + p5)
        const f32x2 b = pk_add(pp[1], pp[3]);  // (p2 + p6, p3 + p7)
        s = fadd(fadd(fadd(a[0], a[1]), b[0]), b[1]);  // 0 + (p0 + p4) is exact
    }
#pragma unroll
    for (int e = NF; e < D; ++e) s = fadd(s, sq[e / 2][e & 1]);
    return s;
}

// Dword-aligned wide loads.  ROCm runs the GPU in unaligned-access mode: global_load_dwordx4 / x2
// accept any 4-byte-aligned address, and the compiler emits them for these under-aligned vector
// types.  So a row of floats never needs a scalar-load variant because of where it starts.
typedef f32x4 f32x4_u __attribute__((aligned(4)));
typedef f32x2 f32x2_u __attribute__((aligned(4)));

// v[0 .. CNT) = p[0 .. CNT) with the widest loads available, v[CNT .. N) = 0.
template <int CNT, int N>
__device__ __forceinline__ void load_row_floats(const float* __restrict__ p, float (&v)[N])
{
    static_assert(CNT <= N, "row longer than its register image");
    constexpr int N4 = (CNT / 4) * 4, N2 = N4 + ((CNT - N4) / 2) * 2;
#pragma unroll
    for (int e = 0; e < N4; e += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4_u*>(p + e);
        v[e] = q[0]; v[e + 1] = q[1]; v[e + 2] = q[2]; v[e + 3] = q[3];
    }
    if (N2 > N4) {
        const f32x2 q = *reinterpret_cast<const f32x2_u*>(p + N4);
        v[N4] = q[0]; v[N4 + 1] = q[1];
    }
    if (CNT > N2) v[N2] = p[N2];
#pragma unroll
    for (int e = CNT; e < N; ++e) v[e] = 0.f;
}

// Same with a run-time (wave-uniform) count: used by the wide sub-vector instantiations, whose
// padded length is a multiple of 8 and whose real length is any value below it.
template <int N>
__device__ __forceinline__ void load_row_floats_rt(const float* __restrict__ p, int cnt, float (&v)[N])
{
#pragma unroll
    for (int e = 0; e < N; e += 4) {
        if (e + 4 <= cnt) {
            const f32x4 q = *reinterpret_cast<const f32x4_u*>(p + e);
            v[e] = q[0]; v[e + 1] = q[1]; v[e + 2] = q[2]; v[e + 3] = q[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[e + i] = (e + i < cnt) ? p[e + i] : 0.f;
        }
    }
}

// One sequential fmaf chain with restarts every kKC (rule 2), global operands with strides.
__device__ inline float chain_dot_global(const float* __restrict__ a, int64_t as,
                                         const float* __restrict__ b, int64_t bs, int n)
{
    float total = 0.f;
    for (int kb = 0; kb < n; kb += kKC) {
        const int ke = (kb + kKC < n) ? kb + kKC : n;
        float ab = 0.f;
        for (int k = kb; k < ke; ++k) ab = ffma(a[k * as], b[k * bs], ab);
        total = (kb == 0) ? ab : fadd(total, ab);
    }
    return total;
}

}  // namespace pqhip
