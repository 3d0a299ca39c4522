// pqhip_rotate.hip -- out = x . Pm on the device: the OPQ rotation x.dot(P) (pq.rs:276), the inverse r.dot(P^T)
// (pq.rs:324) and its GATHER form, which takes its rows straight from the codebook (OPQ reconstruct / lookup without an
// intermediate matrix).  Kernels: k_rotate_pblock8 / k_rotate_pblock9 (P block in LDS, x from global memory into the MFMA
// operands) for 16-byte aligned rows with d % 4 == 0 and d <= 1,280; k_rotate_gemm (P slabs through LDS) for everything else.
#include "pqhip_internal.h"

#include "kernels_rotate.hip.h"
#include "kernels_rotate8.hip.h"
#include "kernels_rotate9.hip.h"

using namespace pqhip;

namespace pqh {

static std::atomic<int> g_rotation_variant{0};   // pqhip_set_rotation_variant: 0 auto, 8 / 9 force k_rotate_pblock8 / 9 (test knob)

// out[n][d] = x[n][d] . Pm   on the device
// ga != nullptr: the rows are gathered from the codebook inside the rotation kernel; returns PQHIP_EUNSUPPORTED when
// the shape has no such kernel (the caller then gathers into a scratch buffer first).
int32_t rotate_dev(const float* d_x, int64_t n, int64_t x_rs, const float* Pm, int d, float* d_out,
                   int64_t o_rs, hipStream_t st, const RotGather* ga_in)
{
    if (n == 0) return PQHIP_OK;
    Rot8Gather gav;
    const Rot8Gather* ga = nullptr;
    if (ga_in) {
        gav.codes = ga_in->codes; gav.c_rs = ga_in->c_rs; gav.cb = ga_in->cb; gav.K = ga_in->K; gav.dsub = ga_in->dsub;
        gav.inv_dsub = ga_in->inv_dsub; gav.sel_rows = ga_in->sel_rows; gav.n_codes = ga_in->n_codes; gav.err = ga_in->err;
        ga = &gav;
    }
    const bool vec = ga ? (d % 4 == 0 && ga->dsub % 4 == 0)
                        : (d % 4 == 0) && (x_rs % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_x) & 15) == 0);
    const bool out_vec = (o_rs % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0);
    // v8 (kernels_rotate8.hip.h): a 64-column P block for all k in LDS (d <= 636), x rows straight from global memory into
    // the MFMA operands, direct 16-byte stores from the accumulators.  v9 (kernels_rotate9.hip.h): the same data flow on
    // v_mfma_f32_16x16x4_f32 with 16-wide column tiles (304 columns executed for d = 300 instead of 320), also with
    // 32-column blocks, which reach d = 1,280.
    const size_t lds8 = ((size_t)((d + 3) / 4) + 1) * 256 * sizeof(float);   // P image + one spare group (pre-reads past the last group)
    const int nb9 = (d + 15) / 16;
    const bool fits9 = (size_t)nb9 * 2048 <= 160 * 1024 && d >= 17;
    if (vec && out_vec && (lds8 <= 160 * 1024 || fits9)) {
        // v9 is the default of the GATHER form: 157 registers let it run 12 waves per workgroup where v8's gather needs 8
        // (OPQ reconstruct of 10 M codes: 16.4 vs 16.85 ms on one box).  For plain rotation it holds a higher clock but
        // pays twice the vector instructions per k (operand transposes, addressing): 1.87 vs 1.83 ms per 1.18 M x 300 rows
        // standalone -- so there it is taken only where v8's 64-column blocks execute >= 10 % more columns than 16-column
        // tiles (d = 272: 320 vs 272, 400: 448 vs 400, 96: 128 vs 96 ...: -5 .. -15 %; tools/rot_variants.py), and beyond
        // d = 640, where no 64-column block fits LDS.  pqhip_set_rotation_variant(9) / (8) force one or the other.
        const int rv = g_rotation_variant.load(std::memory_order_relaxed);
        const bool plain_v9 = 10 * 64 * ((d + 63) / 64) >= 11 * 16 * ((d + 15) / 16);
        const bool narrow9 = (size_t)nb9 * 4096 > 160 * 1024;
        const bool v9 = rv != 8 && (ga != nullptr || rv == 9 || plain_v9 || narrow9) && nb9 >= 2 &&
                        (size_t)nb9 * (narrow9 ? 2048 : 4096) <= 160 * 1024 &&
                        (ga != nullptr || (double)rot_rows_per_wg() * (double)x_rs * 4.0 < 2147483648.0);   // 32-bit row offsets inside a row group
        if (v9 || lds8 <= 160 * 1024) {
            const int rows_per_wg = (ga && !v9) ? rot_rows_per_wg() / 12 * 8 : rot_rows_per_wg();   // 12 (v8's gather form: 8) waves x 12 tiles of 32 rows
            const int ncb = (v9 && narrow9) ? (d + 31) / 32 : (d + 63) / 64;
            const int64_t n_rg = (n + rows_per_wg - 1) / rows_per_wg;
            const int64_t rg_per_xcd = (n_rg + 7) / 8;
            const dim3 grid((unsigned)(rg_per_xcd * ncb * 8));
            StampRun stamps;      // (diagnostic builds: in-kernel s_memtime summary of the launch)
            PQCHK(stamps.begin(diag().rot_stamp, (size_t)grid.x * 12 * 8, st));
            if (v9) {
                const size_t lds9 = (size_t)nb9 * (narrow9 ? 2048 : 4096);
                // template facts: rule-2 split (d > 256), odd number of 16-k bursts, partial last burst
                const bool splitk9 = d > kKC, odd9 = (nb9 & 1) != 0, tail9 = (d & 15) != 0;
#define LAUNCH_ROT9W(W, S, O, T, G)                                                                                 \
                do {                                                                                                \
                    HIPCHK(hipFuncSetAttribute((const void*)k_rotate_pblock9<W, S, O, T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    hipLaunchKernelGGL((k_rotate_pblock9<W, S, O, T, G>), grid, dim3(768), lds9, st, d_x, n, x_rs, Pm, d, d_out, o_rs, rows_per_wg, ncb, \
                                       rg_per_xcd, ga ? *ga : Rot8Gather{}, stamps.ptr());                          \
                } while (0)
#define LAUNCH_ROT9G(S, O, T, G) do { if (narrow9) LAUNCH_ROT9W(2, S, O, T, G); else LAUNCH_ROT9W(4, S, O, T, G); } while (0)
#define LAUNCH_ROT9(S, O, T) do { if (ga) LAUNCH_ROT9G(S, O, T, true); else LAUNCH_ROT9G(S, O, T, false); } while (0)
                if (splitk9) { if (odd9) { if (tail9) LAUNCH_ROT9(true, true, true); else LAUNCH_ROT9(true, true, false); }
                               else      { if (tail9) LAUNCH_ROT9(true, false, true); else LAUNCH_ROT9(true, false, false); } }
                else         { if (odd9) { if (tail9) LAUNCH_ROT9(false, true, true); else LAUNCH_ROT9(false, true, false); }
                               else      { if (tail9) LAUNCH_ROT9(false, false, true); else LAUNCH_ROT9(false, false, false); } }
#undef LAUNCH_ROT9
#undef LAUNCH_ROT9G
#undef LAUNCH_ROT9W
                note_kernel(ga ? "k_rotate_pblock9<gather>" : "k_rotate_pblock9");
            } else {
                // template facts: rule-2 split (d > 256), odd number of full 32-k bursts, partial last burst
                const bool splitk = d > kKC, odd = ((d >> 5) & 1) != 0, tail = (d & 31) != 0;
#define LAUNCH_ROT8G(S, O, T, G)                                                                                    \
                do {                                                                                                \
                    HIPCHK(hipFuncSetAttribute((const void*)k_rotate_pblock8<S, O, T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    hipLaunchKernelGGL((k_rotate_pblock8<S, O, T, G>), grid, dim3(G ? 512 : 768), lds8, st, d_x, n, x_rs, Pm, d, d_out, o_rs, rows_per_wg, ncb, \
                                       rg_per_xcd, ga ? *ga : Rot8Gather{}, stamps.ptr());                          \
                } while (0)
#define LAUNCH_ROT8(S, O, T) do { if (ga) LAUNCH_ROT8G(S, O, T, true); else LAUNCH_ROT8G(S, O, T, false); } while (0)
                if (splitk) { if (odd) { if (tail) LAUNCH_ROT8(true, true, true); else LAUNCH_ROT8(true, true, false); }
                              else     { if (tail) LAUNCH_ROT8(true, false, true); else LAUNCH_ROT8(true, false, false); } }
                else        { if (odd) { if (tail) LAUNCH_ROT8(false, true, true); else LAUNCH_ROT8(false, true, false); }
                              else     { if (tail) LAUNCH_ROT8(false, false, true); else LAUNCH_ROT8(false, false, false); } }
#undef LAUNCH_ROT8
#undef LAUNCH_ROT8G
                note_kernel(ga ? "k_rotate_pblock8<gather>" : "k_rotate_pblock8");
            }
            HIPCHK(hipGetLastError());
            if (stamps.ptr()) {   // diagnostics: synchronous summary on stderr (8 words per wave)
                std::vector<unsigned long long> h;
                PQCHK(stamps.fetch(st, h));
                double tiles = 0, kc = 0, ec = 0, cyc = 0, rt = 0, cmax = 0, cmin = 1e30, wgmax = 0, first = 0, last = 0; size_t waves = 0, wgs = 0;
                for (size_t w0 = 0; w0 < h.size(); w0 += 12 * 8) {
                    double m = 0;
                    for (size_t i = w0; i < w0 + 12 * 8; i += 8)
                        if (h[i]) {
                            tiles += (double)h[i]; kc += (double)h[i + 1]; ec += (double)h[i + 2]; cyc += (double)h[i + 3]; rt += (double)h[i + 4]; ++waves;
                            first += (double)h[i + 6]; last += (double)h[i + 7];
                            cmax = std::max(cmax, (double)h[i + 3]); cmin = std::min(cmin, (double)h[i + 3]); m = std::max(m, (double)h[i + 3]);
                        }
                    if (m > 0) { wgmax += m; ++wgs; }
                }
                if (tiles > 0)
                    fprintf(stderr, "[pqhip] rotate v8/v9 stamps: %zu waves, %.1f tiles/wave, tile %.0f cyc (first %.0f, last %.0f), P staging %.0f cyc/wave, wave life %.0f cyc (min %.0f, max %.0f; slowest wave of a workgroup %.0f), clock %.0f MHz\n",
                            waves, tiles / waves, kc / tiles, first / waves, last / waves, ec / waves, cyc / waves, cmin, cmax, wgmax / wgs, rt > 0 ? cyc / rt * 100.0 : 0.0);
                if (const char* f = diag().rot_stamp_file) {
                    if (FILE* fp = fopen(f, "ab")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fp); fclose(fp); }
                }
            }
            return PQHIP_OK;
        }
    }
    if (ga) return PQHIP_EUNSUPPORTED;           // only v8 / v9 gather inside the kernel
    // any d, any alignment: P slabs double-buffered through LDS (k-block loop restarts the chains every 256 k)
    const bool split = d > kKC;
    constexpr int CT = 5;                      // 320 columns per workgroup
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((d + 2 * CT * 32 - 1) / (2 * CT * 32)));
#define LAUNCH_ROT(SP, VE) \
    hipLaunchKernelGGL((k_rotate_gemm<CT, SP, VE>), grid, dim3(256), 0, st, d_x, n, x_rs, Pm, d, d_out, o_rs)
    if (split) { if (vec) LAUNCH_ROT(true, true); else LAUNCH_ROT(true, false); }
    else { if (vec) LAUNCH_ROT(false, true); else LAUNCH_ROT(false, false); }
#undef LAUNCH_ROT
    HIPCHK(hipGetLastError());
    note_kernel("k_rotate_gemm");
    return PQHIP_OK;
}

}  // namespace pqh

using namespace pqh;

extern "C" {

int32_t pqhip_set_rotation_variant(int32_t variant)
{
    if (variant != 0 && variant != 8 && variant != 9) return PQHIP_EINVAL;
    g_rotation_variant.store(variant, std::memory_order_relaxed);
    return PQHIP_OK;
}

int32_t pqhip_rotate_f32_dev(pqhip_ctx* ctx, int32_t slot, const float* d_x, int64_t n, int64_t x_rs, int64_t d,
                             const float* projection, float* d_out, int64_t o_rs, void* stream)
{
    if (!ctx || !projection || n < 0 || d <= 0 || d > (1 << 24)) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_x || !d_out || x_rs < d || o_rs < d)) return PQHIP_EINVAL;
    SET_DEVICE(ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    DevBuf dp;
    PQCHK(dp.alloc((size_t)d * d * sizeof(float)));
    HIPCHK(hipMemcpyAsync(dp.p, projection, (size_t)d * d * sizeof(float), hipMemcpyHostToDevice, st));
    PQCHK(rotate_dev(d_x, n, x_rs, (const float*)dp.p, (int)d, d_out, o_rs, st));
    HIPCHK(hipStreamSynchronize(st));
    return PQHIP_OK;
}

int32_t pqhip_selftest_mfma_chain(pqhip_ctx* ctx, int32_t slot, int32_t k, int32_t n_trials,
                                  uint64_t seed, int64_t* out_mismatches)
{
    if (!ctx || !out_mismatches || k <= 0 || n_trials <= 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    DeviceSlot& ds = *ctx->devs[slot];
    SET_DEVICE(ds.ordinal);
    unsigned long long* d_cnt = nullptr;
    HIPCHK(hipMalloc((void**)&d_cnt, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), ds.stream[0]));
    hipLaunchKernelGGL(k_selftest_mfma_chain, dim3((unsigned)n_trials), dim3(64), 0, ds.stream[0], (int)k,
                       seed, d_cnt);
    unsigned long long h = 0;
    hipError_t e = hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, ds.stream[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(ds.stream[0]);
    (void)hipFree(d_cnt);
    HIPCHK(e);
    *out_mismatches = (int64_t)h;
    return PQHIP_OK;
}

}  // extern "C"
