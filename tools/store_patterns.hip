// Micro-benchmark behind DESIGN.md "K3 store pattern": which way of laying 16-byte streaming stores of a
// whole-buffer write over workgroups reaches the HBM write rate on MI355X, and how much does the answer move
// with the buffer (two 120 GB allocations A and B alive at once) and its size?  No reads at all: what is
// measured is the store pattern alone.
//   mode 0  "fill"     : workgroup b writes the 16 KB piece b (4 x 4 KB), grid = bytes / 16 KB       (torch fill_)
//   mode 1  "ranges"   : G persistent workgroups, workgroup i streams through its own contiguous 1/G of the buffer
//   mode 2  "strided"  : G persistent workgroups, workgroup i writes blocks i, i + G, i + 2 G, .. of BLK bytes
//   mode 3  "blocks"   : one workgroup per block of BLK bytes, grid = bytes / BLK (dispatch order = address order)
// build: hipcc -O3 --offload-arch=gfx950 tools/store_patterns.hip -o gpurun_out/store_patterns
// usage: store_patterns [GB ...]        (default 36 120)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ inline void st(f32x4* p, f32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT> __global__ __launch_bounds__(256) void k_fill(f32x4* out, long n16)
{
    const f32x4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    const long u0 = (long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (u0 + 256 * j < n16) st<NT>(out + u0 + 256 * j, v);
}
template <bool NT> __global__ __launch_bounds__(256) void k_ranges(f32x4* out, long n16, long per_wg16)
{
    const f32x4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    const long b = (long)blockIdx.x * per_wg16;
    const long e = (b + per_wg16 < n16) ? b + per_wg16 : n16;
#pragma unroll 4
    for (long u = b + threadIdx.x; u < e; u += 256) st<NT>(out + u, v);
}
template <bool NT> __global__ __launch_bounds__(256) void k_strided(f32x4* out, long n16, long blk16)
{
    const f32x4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    const long nblk = (n16 + blk16 - 1) / blk16;
    for (long bl = blockIdx.x; bl < nblk; bl += gridDim.x) {
        const long b = bl * blk16;
        const long e = (b + blk16 < n16) ? b + blk16 : n16;
#pragma unroll 4
        for (long u = b + threadIdx.x; u < e; u += 256) st<NT>(out + u, v);
    }
}
template <bool NT> __global__ __launch_bounds__(256) void k_blocks(f32x4* out, long n16, long blk16)
{
    const f32x4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    const long b = (long)blockIdx.x * blk16;
    const long e = (b + blk16 < n16) ? b + blk16 : n16;
#pragma unroll 4
    for (long u = b + threadIdx.x; u < e; u += 256) st<NT>(out + u, v);
}

static double run(int mode, bool nt, f32x4* buf, long n16, long G, long blk16)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a, 0));
        if (mode == 0) { const unsigned g = (unsigned)((n16 + 1023) / 1024); if (nt) k_fill<true><<<g, 256>>>(buf, n16); else k_fill<false><<<g, 256>>>(buf, n16); }
        if (mode == 1) { const long per = (((n16 + G - 1) / G) + 255) / 256 * 256; if (nt) k_ranges<true><<<(unsigned)G, 256>>>(buf, n16, per); else k_ranges<false><<<(unsigned)G, 256>>>(buf, n16, per); }
        if (mode == 2) { if (nt) k_strided<true><<<(unsigned)G, 256>>>(buf, n16, blk16); else k_strided<false><<<(unsigned)G, 256>>>(buf, n16, blk16); }
        if (mode == 3) { const unsigned g = (unsigned)((n16 + blk16 - 1) / blk16); if (nt) k_blocks<true><<<g, 256>>>(buf, n16, blk16); else k_blocks<false><<<g, 256>>>(buf, n16, blk16); }
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float t; CHECK(hipEventElapsedTime(&t, a, b));
        if (rep) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
    std::vector<long> gbs;
    for (int i = 1; i < argc; ++i) gbs.push_back(atol(argv[i]));
    if (gbs.empty()) gbs = {36, 120};
    const long cap = 120l * 1000 * 1000 * 1000;
    f32x4 *A, *B;
    CHECK(hipMalloc(&A, cap)); CHECK(hipMalloc(&B, cap));
    CHECK(hipMemset(A, 0, cap)); CHECK(hipMemset(B, 0, cap));
    CHECK(hipDeviceSynchronize());
    struct Cfg { int mode; long G; long blk; const char* name; };
    const long KB = 1024;
    std::vector<Cfg> cfgs = {
        {0, 0, 0, "fill 16K/wg"},
        {3, 0, 16 * KB, "blocks 16K"}, {3, 0, 76800, "blocks 76.8K"}, {3, 0, 256 * KB, "blocks 256K"}, {3, 0, 1024 * KB, "blocks 1M"}, {3, 0, 4096 * KB, "blocks 4M"},
        {1, 512, 0, "ranges G=512"}, {1, 1024, 0, "ranges G=1024"}, {1, 2048, 0, "ranges G=2048"}, {1, 4096, 0, "ranges G=4096"},
        {2, 1024, 16 * KB, "strided G=1024 16K"}, {2, 1024, 76800, "strided G=1024 76.8K"}, {2, 1024, 1024 * KB, "strided G=1024 1M"},
        {2, 2048, 16 * KB, "strided G=2048 16K"}, {2, 2048, 76800, "strided G=2048 76.8K"}, {2, 2048, 1024 * KB, "strided G=2048 1M"},
    };
    for (long gb : gbs) {
        const long bytes = gb * 1000 * 1000 * 1000;
        const long n16 = bytes / 16;
        for (const Cfg& c : cfgs)
            for (int nt = 1; nt >= 0; --nt) {
                const double a = run(c.mode, nt, A, n16, c.G, c.blk / 16);
                const double b = run(c.mode, nt, B, n16, c.G, c.blk / 16);
                printf("{\"GB\": %ld, \"pattern\": \"%s\", \"nontemporal\": %d, \"A_ms\": %.3f, \"A_TBps\": %.3f, \"B_ms\": %.3f, \"B_TBps\": %.3f}\n",
                       gb, c.name, nt, a, bytes / a / 1e9, b, bytes / b / 1e9);
                fflush(stdout);
            }
    }
    return 0;
}
