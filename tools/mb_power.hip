// Micro-benchmark (diagnostic, not product): SUSTAINED v_mfma_f32_32x32x2_f32 rate of the whole chip
// with non-trivial operands, for seconds, so that clock and power settle (sample rocm-smi beside it).
// Answers: what FP32-MFMA rate can this GPU actually hold -- is the 157.3 TFLOP/s figure (2.4 GHz)
// reachable under the power cap?   usage: mb_power [seconds] [waves_per_simd] [valu_per_mfma: 0|2|4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void burn(float* out, int iters, float seed)
{
    f32x16 a0 = {0}, a1 = {0};
    float a = seed + threadIdx.x * 0.37f, b = 1.0f - threadIdx.x * 0.011f, t0 = a, t1 = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, a1, 0, 0, 0);
            if (NV >= 2) { asm volatile("v_fma_f32 %0, %2, %3, %0\nv_fma_f32 %1, %3, %2, %1" : "+v"(t0), "+v"(t1) : "v"(a), "v"(b)); }
            if (NV >= 4) { asm volatile("v_fma_f32 %0, %2, %3, %0\nv_fma_f32 %1, %3, %2, %1" : "+v"(t0), "+v"(t1) : "v"(a), "v"(b)); }
        }
        a = a * 0.999f + 0.001f;   // keep operands moving (data-dependent power)
    }
    float s = t0 + t1;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 2;
    const int nv = argc > 3 ? atoi(argv[3]) : 0;
    const int blocks = 256 * wps;  // 256 CUs x (wps waves per SIMD): one 256-thread block = 1 wave per SIMD of a CU
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;   // 16 MFMA per iteration per wave
    double total_ms = 0; long launches = 0;
    while (total_ms < secs * 1e3) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) {
            if (nv == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
            else if (nv == 2) hipLaunchKernelGGL(burn<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
            else hipLaunchKernelGGL(burn<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; launches += 8;
        const double flop = 8.0 * blocks * 4 * (double)iters * 16 * 4096;
        printf("wps=%d nv=%d  %.1f TFLOP/s (MFMA only counted)  %.1f ms per 8 launches\n", wps, nv, flop / (ms * 1e-3) / 1e12, ms);
        fflush(stdout);
    }
    return 0;
}
