// Timing-only harness (diagnostic, not product; never linked into libpqhip.so): launches k_rotate_pblock8 on a
// 1,179,648 x 300 random batch for ~2 s per build and prints kernel ms, cycles per tile and the in-kernel clock.
// Built once per ablation:  hipcc -DPQHIP_TIMING_ONLY_BUILD -DROT8_ABLATE=n ...   (tools/rot8_ablate.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../reductive_amd/csrc/kernels_rotate9.hip.h"
using namespace pqhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const int d = 300; const int64_t n = 1179648;
    std::vector<float> hx((size_t)n * d), hp((size_t)d * d);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hp) v = rnd() * 0.06f;
    float *x, *P, *out; unsigned long long* st;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&P, hp.size() * 4)); CK(hipMalloc(&out, hx.size() * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(P, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    const int rows_per_wg = 4608, ncb = 5;
    const int64_t n_rg = (n + rows_per_wg - 1) / rows_per_wg, rg_per_xcd = (n_rg + 7) / 8;
    const dim3 grid((unsigned)(rg_per_xcd * ncb * 8));
    const size_t lds = (size_t)((d + 15) / 16) * 4096;
    const size_t n_stamp = (size_t)grid.x * 12 * 8;
    CK(hipMalloc(&st, n_stamp * 8));
    auto k = k_rotate_pblock9<4, true, true, true, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total = 0; float last = 0;
    while (total < secs * 1e3) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k, grid, dim3(768), lds, 0, x, n, (int64_t)d, P, d, out, (int64_t)d, rows_per_wg, ncb, rg_per_xcd, Rot8Gather{}, (unsigned long long*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&last, e0, e1)); total += last;
    }
    CK(hipMemset(st, 0, n_stamp * 8));
    hipLaunchKernelGGL(k, grid, dim3(768), lds, 0, x, n, (int64_t)d, P, d, out, (int64_t)d, rows_per_wg, ncb, rg_per_xcd, Rot8Gather{}, st);
    std::vector<unsigned long long> h(n_stamp);
    CK(hipMemcpy(h.data(), st, n_stamp * 8, hipMemcpyDeviceToHost));
    double tiles = 0, kc = 0, cyc = 0, rt = 0, wgmax = 0; size_t waves = 0, wgs = 0;
    for (size_t w0 = 0; w0 < n_stamp; w0 += 96) {
        double m = 0;
        for (size_t i = w0; i < w0 + 96; i += 8) if (h[i]) { tiles += h[i]; kc += h[i + 1]; cyc += h[i + 3]; rt += h[i + 4]; ++waves; if ((double)h[i + 3] > m) m = (double)h[i + 3]; }
        if (m > 0) { wgmax += m; ++wgs; }
    }
    printf("v9 (16x16x4) [%d]: %.3f ms per launch (%.1f TFLOP/s on 2 d^2), tile %.0f cyc, wave life %.0f cyc, slowest wave of a workgroup %.0f cyc, clock %.0f MHz\n",
           ROT8_ABLATE, last / 8, 2.0 * d * d * n / (last / 8 * 1e-3) / 1e12, kc / tiles, cyc / waves, wgmax / wgs, cyc / rt * 100.0);
    return 0;
}
