// Micro-benchmark (diagnostic, not product): sustained FP32 MFMA rate of the whole chip on RANDOM operands for
// the two f32 shapes, v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, at equal FLOP per wave, plus the chain-order
// self-test of the 16x16x4 form (is D = fma(a3 b3, fma(a2 b2, fma(a1 b1, fma(a0 b0, C))))?).
// Round 3 question: the OPQ pipeline is power-limited (encode clock 2.0 GHz after rotation v8, 2.15 after v6, same
// cycles) -- does the 16x16 shape, which moves half the accumulator bytes per MAC, hold a higher clock?
// usage: mb_shape [seconds per variant] [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float hash_unit(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = (float)(int)((x >> 40) & 0xFFFFFF) / 16777216.0f - 0.5f;
    const int e = (int)((x >> 8) & 7) - 3;
    return ldexpf(u, e);
}

// SHAPE 0: 32x32x2, 4 accumulators of 16 regs;  SHAPE 1: 16x16x4, 16 accumulators of 4 regs  (64 accumulator regs both)
template <int SHAPE>
__global__ __launch_bounds__(256) void burn(float* out, int iters, uint64_t seed, unsigned long long* clk)
{
    float a[8], b[8];
    const uint64_t t = seed + (uint64_t)(blockIdx.x * 256 + threadIdx.x) * 7919;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = hash_unit(t * 31 + i); b[i] = hash_unit(t * 17 + i + 100); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + q) & 7], b[(u + 3 * q) & 7], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    } else {
        f32x4 acc[16] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + q) & 7], b[(u + 3 * q) & 7], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// one wave per trial: random A[16][k], B[k][16]; compares the 16x16x4 MFMA chain with a scalar k-ordered fmaf chain
__global__ void selftest16(int k, uint64_t seed, unsigned long long* mismatches)
{
    const int lane = threadIdx.x & 63;
    const int i16 = lane & 15, qd = lane >> 4;
    const uint64_t base = seed + (uint64_t)blockIdx.x * 1000003ull;
    auto A = [&](int i, int kk) { return hash_unit(base * 31 + (uint64_t)i * 4099 + kk); };
    auto B = [&](int kk, int jj) { return hash_unit(base * 17 + (uint64_t)jj * 8209 + kk + 77777); };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < k; k0 += 4) {
        const int kk = k0 + qd;
        const float av = (kk < k) ? A(i16, kk) : 0.f;
        const float bv = (kk < k) ? B(kk, i16) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    unsigned long long bad = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * qd + r;           // D row (A row index); column = lane & 15
        float ref = 0.f;
        for (int kk = 0; kk < k; ++kk) ref = __fmaf_rn(A(i, kk), B(kk, i16), ref);
        if (__float_as_uint(ref) != __float_as_uint(acc[r])) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 1;
    unsigned long long* d_bad;
    hipMalloc(&d_bad, 8);
    for (int k : {4, 8, 20, 44, 256, 300}) {
        hipMemset(d_bad, 0, 8);
        hipLaunchKernelGGL(selftest16, dim3(2048), dim3(64), 0, 0, k, 12345ull + k, d_bad);
        unsigned long long h = 0;
        hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
        printf("selftest 16x16x4 chain order, k=%d: %llu mismatches of %d\n", k, h, 2048 * 256);
    }
    const int blocks = 256 * wps;
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;   // 32 (shape 0) / 64 (shape 1) MFMA per iteration per wave: 131072 MAC-lanes... equal FLOP
    for (int round = 0; round < 2; ++round)
    for (int shape = 0; shape < 2; ++shape) {
        double total_ms = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int k = 0; k < 4; ++k) {
                if (shape == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 99ull + k, clk);
                else hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 99ull + k, clk);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            total_ms += ms;
            unsigned long long h[2];
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flop = 4.0 * blocks * 4 * (double)iters * 32 * 4096;
            printf("shape %s wps=%d  %.1f TFLOP/s  %.1f ms per 4 launches  in-kernel clock %.0f MHz  cycles/MFMA-slot %.1f\n",
                   shape ? "16x16x4" : "32x32x2", wps, flop / (ms * 1e-3) / 1e12, ms, (double)h[0] / (double)h[1] * 100.0,
                   (double)h[0] / ((double)iters * 32 * wps));
            fflush(stdout);
        }
    }
    return 0;
}
