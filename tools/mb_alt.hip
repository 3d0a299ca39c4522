// Diagnostic micro-benchmark #4: v_mfma_f32_32x32x2_f32 issue rate with 1, 2 and 4 independent
// accumulators interleaved (no other instructions), 1-2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void kern(float* out, unsigned long long* cyc, int iters, float av)
{
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    float a = av + threadIdx.x * 1e-6f, b = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC> void run(int block)
{
    const int grid = 256, iters = 2000;
    float* out; unsigned long long* cyc; const int nw = grid * block / 64;
    (void)hipMalloc(&out, sizeof(float) * grid * block); (void)hipMalloc(&cyc, 8 * nw);
    kern<NACC><<<grid, block>>>(out, cyc, 10, 1.0f);
    kern<NACC><<<grid, block>>>(out, cyc, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, 8 * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("accumulators=%d waves/SIMD=%d: %.2f cycles per MFMA per SIMD\n", NACC, block / 256,
           h[nw / 2] / (iters * 8.0 * NACC) / (block / 256));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() { for (int b : {256, 512}) { run<1>(b); run<2>(b); run<4>(b); } return 0; }
