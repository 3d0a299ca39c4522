// Micro-benchmark (diagnostic, not product): HBM write rate of 16-byte nontemporal stores when a
// 12 GB [n][300] f32 matrix is written (a) as whole rows by one workgroup per row range, or (b) as
// NP column pieces of each row by NP different workgroup sets (what an LDS-resident codebook slice
// per workgroup would produce in the reconstruct kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void w(float* out, long n, int d, int np, int rows_per_wg)
{
    const int piece = blockIdx.x % np;
    const long rg = blockIdx.x / np;
    const int c0 = (d / 4) * piece / np, c1 = (d / 4) * (piece + 1) / np;  // 16-byte chunk range of the piece
    const int cw = c1 - c0;
    const long r0 = rg * rows_per_wg;
    const long r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
    const long total = (r1 - r0) * cw;
    const f4 v = {1.f, 2.f, 3.f, (float)piece};
    for (long L = threadIdx.x; L < total; L += 256) {
        const long r = L / cw;
        const int c = (int)(L - r * cw);
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (r0 + r) * d + 4 * (c0 + c)));
    }
}

// (c) interleaved: workgroup b writes 64-row blocks b, b + G, b + 2G, ... (all workgroups advance through
// memory together) instead of one contiguous range each
__global__ __launch_bounds__(256) void wi(float* out, long n, int d, int rows_per_block)
{
    const long nblocks = (n + rows_per_block - 1) / rows_per_block;
    const int cw = d / 4;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long r0 = blk * rows_per_block;
        const long r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
        const long total = (r1 - r0) * cw;
        for (long L = threadIdx.x; L < total; L += 256)
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + r0 * d + 4 * L));
    }
}

int main(int argc, char** argv)
{
    const long n = 10000000; const int d = 300;
    float* out; hipMalloc(&out, (size_t)n * d * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int np : {1, 2, 3, 5}) {
        for (int rpw : {2048, 8192}) {
            const long rgs = (n + rpw - 1) / rpw;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(w, dim3((unsigned)(rgs * np)), dim3(256), 0, 0, out, n, d, np, rpw);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("pieces=%d rows/wg=%d  %.2f TB/s\n", np, rpw, 5.0 * n * d * 4 / (ms * 1e-3) / 1e12);
            }
        }
    }
    for (int rpb : {16, 64, 256}) {
        for (int g : {1024, 2048, 4096}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(wi, dim3(g), dim3(256), 0, 0, out, n, d, rpb);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("interleaved rows/block=%d grid=%d  %.2f TB/s\n", rpb, g, 5.0 * n * d * 4 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
