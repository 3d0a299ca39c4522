// mb_stream.hip -- how fast can [n][128] f32 rows be streamed from HBM with the access patterns the small-codebook encode
// kernels can use?  Every kernel reads all n x 512 bytes once and folds them into a checksum (one dword per wave written).
//   P0  linear: a wave reads 1 KiB contiguous per instruction, 8 instructions in flight, consecutive KiB per wave
//   P1  k_encode_smallk's pattern: wave = 64 rows; per instruction 8 rows x 128 B; 8 instructions = one 32-float chunk
//   P2  k_encode_small16's pattern: per instruction 16 rows x 64 B
//   P3  wave = 64 rows read as one contiguous 32 KiB block, 1 KiB per instruction, 8 in flight
//   P4  one row per lane: per instruction 64 rows x 16 B
// build: hipcc -O3 --offload-arch=gfx950 tools/mb/mb_stream.hip -o tools/mb/mb_stream ; run: tools/mb/mb_stream [n_rows] [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ x, int64_t n, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row0 = wave * 64;
    if (row0 >= n) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const char* base = reinterpret_cast<const char*>(x + row0 * 128);
    // 32 instructions per wave (64 rows x 512 B), issued in 4 groups of 8
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            size_t off;
            if (P == 0 || P == 3) off = (size_t)(8 * g + i) * 1024 + 16 * lane;
            else if (P == 1) { const int p = lane + 64 * i; const int r = p >> 3, c = p & 7; off = (size_t)r * 512 + 128 * g + 16 * c; }
            else if (P == 2) { const int rb = i & 3, bu = i >> 2; off = (size_t)(16 * rb + (lane & 15)) * 512 + 128 * g + 64 * bu + 16 * (lane >> 4); }
            else { off = (size_t)lane * 512 + 128 * g + 16 * i; }
            v[i] = *reinterpret_cast<const f32x4*>(base + off);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[wave] = s;   // never true for the test data: keeps the loads alive without a store stream
}

template <int P>
static float run(const float* x, int64_t n, float* out, int reps, size_t lds = 0)
{
    const unsigned grid = (unsigned)((n / 64 + 3) / 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_stream<P>, dim3(grid), dim3(256), lds, 0, x, n, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream<P>, dim3(grid), dim3(256), lds, 0, x, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10'000'000;
    float *x, *out;
    CK(hipMalloc(&x, (size_t)n * 512));
    CK(hipMalloc(&out, (size_t)(n / 64 + 8) * 4));
    CK(hipMemset(x, 0, (size_t)n * 512));
    const double gb = (double)n * 512 / 1e9;
    float ms;
    ms = run<0>(x, n, out, 20); printf("P0 linear                 %.4f ms  %.2f TB/s\n", ms, gb / ms);
    ms = run<1>(x, n, out, 20); printf("P1 8 rows x 128 B / instr %.4f ms  %.2f TB/s\n", ms, gb / ms);
    ms = run<2>(x, n, out, 20); printf("P2 16 rows x 64 B / instr %.4f ms  %.2f TB/s\n", ms, gb / ms);
    ms = run<3>(x, n, out, 20); printf("P3 = P0 (same mapping)    %.4f ms  %.2f TB/s\n", ms, gb / ms);
    ms = run<4>(x, n, out, 20); printf("P4 64 rows x 16 B / instr %.4f ms  %.2f TB/s\n", ms, gb / ms);
    // occupancy: dynamic LDS per workgroup limits the workgroups per CU (160 KB): 4 / 2 / 1 waves per SIMD
    for (size_t lds : {(size_t)40000, (size_t)80000, (size_t)150000}) {
        CK(hipFuncSetAttribute((const void*)k_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)k_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)k_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ms = run<0>(x, n, out, 20, lds); printf("lds %6zu  P0 %.4f ms  %.2f TB/s\n", lds, ms, gb / ms);
        ms = run<2>(x, n, out, 20, lds); printf("lds %6zu  P2 %.4f ms  %.2f TB/s\n", lds, ms, gb / ms);
        ms = run<4>(x, n, out, 20, lds); printf("lds %6zu  P4 %.4f ms  %.2f TB/s\n", lds, ms, gb / ms);
    }
    return 0;
}
