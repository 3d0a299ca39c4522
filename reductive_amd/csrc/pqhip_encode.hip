// pqhip_encode.hip -- PQ encode dispatch of libpqhip.so: which kernel family serves a (K, dsub, index width) shape,
// its launch geometry, and the grouped / wide-sub-vector forms that go through 64-bit keys.
// (primitives.rs:64-104 -> kmeans.rs:133-159 -> linalg.rs:150-180, fused in every kernel.)
#include "pqhip_internal.h"

#include "kernels_anchor.hip.h"
#include "kernels_pair16.hip.h"
#include "encode_launch.h"
#include "smallk_launch.h"
#include "small16_launch.h"
#include "vor2_launch.h"
#include "wide_launch.h"

using namespace pqhip;

namespace pqh {

// K > 256 on the MFMA path: every subquantizer is presented to the default kernel as `groups`
// virtual subquantizers of 256 centroids; the kernel leaves a 64-bit key {ordered distance, global
// index} per (row, virtual m) and k_merge_keys reduces them to u32 codes.  Rows are chunked so that
// the key buffer stays <= 1 GiB; the buffer is leased from the codebook's scratch pool for the call.
static int32_t encode_grouped_dev(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs,
                           void* d_codes, int64_t o_rs, hipStream_t st)
{
    CodebookDev& cd = cb->dev[slot];
    const int64_t Mv = cb->M * cb->groups;
    const int64_t chunk = std::min<int64_t>(n, std::max<int64_t>(4096, (1ll << 30) / (Mv * 8)));
    ScratchLease keys(cb, slot, st);
    PQCHK(keys.acquire((size_t)chunk * Mv * sizeof(unsigned long long)));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min<int64_t>(chunk, n - r0);
        EncodeArgs a;
        a.x = d_x + r0 * x_rs; a.n = rows; a.x_rs = x_rs; a.out = keys.ptr(); a.o_rs = Mv;
        a.frags = cd.frags; a.cc = cd.cc; a.cb = cd.cb;
        a.M = (int)Mv; a.K = (int)cb->K; a.dsub = (int)cb->dsub; a.k_pad = cb->k_pad;
        a.groups = cb->groups;
        a.bad_flag = nullptr;
        int64_t rpi = round_up((rows * Mv + 4 * 4096 - 1) / (4 * 4096), 32);
        rpi = std::max<int64_t>(32, std::min<int64_t>(1024, rpi));
        a.rows_per_item = (int)rpi;
        a.n_chunks = (rows + 4 * rpi - 1) / (4 * rpi);
        a.chunks_per_xcd = (a.n_chunks + 7) / 8;
        const dim3 grid((unsigned)(a.chunks_per_xcd * Mv * 8));
        if (!launch_encode_mfma(2, 8, cb->DP, cb->DP == cb->dsub, 8, a, grid, st, diag().lds_pad)) return PQHIP_EUNSUPPORTED;
        const unsigned mg = (unsigned)std::min<int64_t>((rows * cb->M + 255) / 256, 256 * 32);
        hipLaunchKernelGGL((k_merge_keys<uint32_t>), dim3(mg), dim3(256), 0, st, (const unsigned long long*)keys.ptr(), rows, (int)cb->M, cb->groups,
                           (uint32_t*)d_codes + r0 * o_rs, o_rs);
        HIPCHK(hipGetLastError());
        note_kernel("k_encode_mfma_lds3<grouped>");
        note_kernel("k_merge_keys");
    }
    cb->last_kernel = "k_encode_mfma_lds3<grouped>";
    return PQHIP_OK;
}

// 128 < dsub <= 1,024 (kernels_mfma_wide.hip.h; beyond 256 floats the multi-block kernel k_encode_mfma_wide2): squared norms by a pre-pass, one 64-bit key per (row, group of <= 128
// centroids) from the matrix-core kernel, k_merge_keys -> codes.  Keys and norms live in one leased scratch buffer, rows are
// chunked so that it stays <= 1 GiB.
static int32_t encode_wide_dev(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs,
                        void* d_codes, int code_bytes, int64_t o_rs, hipStream_t st)
{
    CodebookDev& cd = cb->dev[slot];
    const int64_t Mv = cb->M * cb->groups;
    const int64_t per_row = Mv * 8 + cb->M * 4;
    const int64_t chunk = std::min<int64_t>(n, std::max<int64_t>(4096, (1ll << 30) / per_row));
    ScratchLease buf(cb, slot, st);
    PQCHK(buf.acquire((size_t)chunk * per_row));
    unsigned long long* keys = (unsigned long long*)buf.ptr();
    float* xx = (float*)(keys + chunk * Mv);
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min<int64_t>(chunk, n - r0);
        launch_row_norms(d_x + r0 * x_rs, rows, x_rs, (int)cb->M, (int)cb->dsub, xx, st);
        EncodeArgs a;
        a.x = d_x + r0 * x_rs; a.n = rows; a.x_rs = x_rs; a.out = keys; a.o_rs = Mv;
        a.frags = cd.frags; a.cc = cd.cc; a.cb = cd.cb;
        a.M = (int)Mv; a.K = (int)cb->K; a.dsub = (int)cb->dsub; a.k_pad = cb->k_pad;
        a.groups = cb->groups;
        a.bad_flag = nullptr;
        // one wave per SIMD and one workgroup per CU: ~4 row streams per CU and round
        int64_t rpi = round_up((rows * Mv + 4 * 1024 - 1) / (4 * 1024), 32);
        rpi = std::max<int64_t>(32, std::min<int64_t>(512, rpi));
        a.rows_per_item = (int)rpi;
        a.n_chunks = (rows + 4 * rpi - 1) / (4 * rpi);
        a.chunks_per_xcd = (a.n_chunks + 7) / 8;
        const dim3 grid((unsigned)(a.chunks_per_xcd * Mv * 8));
        if (!launch_encode_wide(cb->T, cb->DP, a, xx, grid, st)) return PQHIP_EUNSUPPORTED;
        const unsigned mg = (unsigned)std::min<int64_t>((rows * cb->M + 255) / 256, 256 * 32);
        if (code_bytes == 1)
            hipLaunchKernelGGL((k_merge_keys<uint8_t>), dim3(mg), dim3(256), 0, st, (const unsigned long long*)keys, rows, (int)cb->M, cb->groups,
                               (uint8_t*)d_codes + r0 * o_rs, o_rs);
        else
            hipLaunchKernelGGL((k_merge_keys<uint32_t>), dim3(mg), dim3(256), 0, st, (const unsigned long long*)keys, rows, (int)cb->M, cb->groups,
                               (uint32_t*)d_codes + r0 * o_rs, o_rs);
        HIPCHK(hipGetLastError());
        note_kernel("k_row_norms");
        note_kernel("k_encode_mfma_wide");
        note_kernel("k_merge_keys");
    }
    cb->last_kernel = "k_encode_mfma_wide";
    return PQHIP_OK;
}

// PQ encode of device-resident, already rotated rows.
// bad_flag != nullptr: the matrix-core kernel is launched whatever the host last knew about the
// centroid norms and consults the device flag itself (captured k-means iterations).
int32_t encode_plain_dev(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs,
                         void* d_codes, int code_bytes, int64_t o_rs, hipStream_t st,
                         const int* bad_flag, bool beside_update)
{
    if (n == 0) return PQHIP_OK;
    if (cb->variant == 8 && !cb->has_proj) return PQHIP_EUNSUPPORTED;   // variant 8 = the fused OPQ kernel only
    CodebookDev& cd = cb->dev[slot];
    if (cb->wide) {
        if (cb->variant != 1 && cb->norms_ok && (code_bytes == 1 || code_bytes == 4))
            return encode_wide_dev(cb, slot, d_x, n, x_rs, d_codes, code_bytes, o_rs, st);
        // (anything else: the scalar anchor kernel below)
    } else
    if (cb->groups > 1 && cb->variant != 1 && cb->norms_ok && code_bytes == 4)
        return encode_grouped_dev(cb, slot, d_x, n, x_rs, d_codes, o_rs, st);
    // 2-float sub-vectors, K <= 256: only the centroids that can win in the point's grid cell are evaluated (kernels_vor2.hip.h;
    // the tables exist for Pq handles whose centroids are finite and within range).  Variant 11 forces it.
    // Auto above 16 centroids (tools: bench.py --d .. --variant 0 / 2 / 4, vectors/s against the best kernel that evaluates every
    // centroid): d=20 M=10 K=128 (the reference's test shape) 1.7e10 / 5.3e9, K=256 1.5e10 / 2.5e9, K=32 1.44e10 / 1.31e10;
    // d=64 M=32 K=128 6.3e9 / 1.75e9; d=300 M=150 K=256 9.3e8 / 1.7e8; one-float sub-vectors d=128 M=128 K=256 2.0e9 / 2.0e8.
    // Up to 16 centroids the pair kernel below is faster (d=128 M=64 K=16: 3.6e9 against 1.8e9 here).
    if ((cb->variant == 11 || (cb->variant == 0 && (cb->K > 16 || cb->dsub == 1))) && cb->vor2 && code_bytes == 1 && cb->norms_ok && bad_flag == nullptr) {
        Vor2Launch l;
        l.x = d_x; l.n = n; l.x_rs = x_rs; l.out = (uint8_t*)d_codes; l.o_rs = o_rs;
        l.cb = cd.cb; l.cc = cd.cc; l.tab = cd.vor2_tab; l.off = cd.vor2_off;
        l.M = (int)cb->M; l.K = (int)cb->K; l.k_pad = cb->k_pad; l.dsub = (int)cb->dsub; l.max_region_words = cb->vor2_max_region_words;
        l.n_cus = cb->ctx->devs[slot]->n_cus;
        if (launch_vor2(l, st)) {
            HIPCHK(hipGetLastError());
            cb->last_kernel = "k_encode_vor2";
            note_kernel("k_encode_vor2");
            return PQHIP_OK;
        }
    }
    if (cb->variant == 11) return PQHIP_EUNSUPPORTED;
    // K <= 16 with sub-vectors of 2 / 4 / 8 / 16 floats: one matrix tile serves two subquantizers, x is read once in whole
    // lines (kernels_pair16.hip.h).  Variant 7 forces it; variants 1..6 keep the others.
    // Measured (tools/smallk_ab.sh, one box, vectors/s pair / VALU kernel / default MFMA kernel): d=128 M=64 (dsub 2) 3.59e9 / 3.31e9 /
    // 1.50e9; d=300 M=75 (dsub 4) 2.14e9 / 1.41e9 / 1.21e9; d=128 M=32 (dsub 4) 5.09e9 / 5.61e9 / 2.89e9; d=128 M=16 (dsub 8, the
    // reference's bench shape) 5.95e9 / 6.69e9 / 4.76e9; d=768 M=48 (dsub 16) 1.14e9 / 0.70e9 / 1.18e9 -- the zero blocks double the
    // matrix time, which the shared FP32 pipe charges in full, so auto takes it only where it wins: dsub 2, and dsub 4 with many
    // subquantizers.
    const bool pair_auto = cb->variant == 0 && (cb->dsub == 2 || (cb->dsub == 4 && cb->M >= 48));
    if ((pair_auto || cb->variant == 7) && cb->pair16 && code_bytes == 1 && cb->norms_ok && bad_flag == nullptr) {
        Pair16Args a;
        const int NP = (int)((cb->M + 1) / 2);
        a.x = d_x; a.n = n; a.x_rs = x_rs; a.out = (uint8_t*)d_codes; a.o_rs = o_rs;
        a.fragp = cd.fragp; a.ccp = cd.fragp + (int64_t)NP * cb->dsub * 64; a.cb = cd.cb; a.cc = cd.cc;
        a.M = (int)cb->M; a.K = (int)cb->K; a.k_pad = cb->k_pad; a.NP = NP;
        a.n_tiles = (n + 31) / 32;
        const size_t lds = ((size_t)NP * cb->dsub * 64 + (size_t)NP * 32 + 4 * 2 * 32 * 36) * sizeof(float);
        const int per_cu = std::max<int>(1, std::min<int>(3, (int)(160 * 1024 / lds)));
        const unsigned grid = (unsigned)std::min<int64_t>((a.n_tiles + 3) / 4, (int64_t)cb->ctx->devs[slot]->n_cus * per_cu);
#define LAUNCH_P16(D)                                                                                                   \
        do {                                                                                                            \
            HIPCHK(hipFuncSetAttribute((const void*)k_encode_pair16<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            hipLaunchKernelGGL((k_encode_pair16<D>), dim3(grid), dim3(256), lds, st, a);                                 \
        } while (0)
        switch ((int)cb->dsub) {
        case 2: LAUNCH_P16(2); break;
        case 4: LAUNCH_P16(4); break;
        case 8: LAUNCH_P16(8); break;
        default: LAUNCH_P16(16); break;
        }
#undef LAUNCH_P16
        HIPCHK(hipGetLastError());
        cb->last_kernel = "k_encode_pair16";
        note_kernel("k_encode_pair16");
        return PQHIP_OK;
    }
    if (cb->variant == 7) return PQHIP_EUNSUPPORTED;
    // K <= 32, sub-vectors of 4 / 8 / 12 / 16 / 20 / 24 / 32 floats and 16-byte aligned rows: the 16x16x4 kernel with the
    // transposed codebook image in LDS (kernels_small16.hip.h).  Variant 10 forces it.
    const bool s16_fits = cb->KP != 0 && small16_has(cb->KP, (int)cb->dsub) && code_bytes == 1 && cb->norms_ok && bad_flag == nullptr &&
                          x_rs % 4 == 0 && ((uintptr_t)d_x & 15) == 0 && small16_lds_bytes((int)cb->M, (int)cb->dsub, cb->KP) <= 96 * 1024;
    // Auto wherever it fits: it is the fastest kernel for every such shape measured (tools/small16_sweep.sh, vectors/s against the
    // best of the others): d=128 M=16 K=16 7.4e9 / 6.2e9, d=128 M=32 (dsub 4) 5.6e9 / 5.0e9, d=256 M=32 4.1e9 / 3.2e9, d=768 M=96
    // 1.29e9 / 0.85e9, d=64 M=8 1.41e10 / 1.10e10, K=32: d=128 M=16 5.2e9 / 4.5e9, d=300 M=75 1.49e9 / 1.25e9 -- except 4-float
    // sub-vectors from 48 subquantizers on at K <= 16, which stay with the pair kernel above (d=300 M=75: 1.95e9 / 2.07e9).
    // 16- and 32-float sub-vectors (tools/small16_sweep16.sh): d=768 M=48 K=16 1.48e9 / 1.11e9, d=128 M=8 8.0e9 / 6.3e9,
    // d=1024 M=64 1.09e9 / 0.85e9, d=1024 M=32 1.25e9 / 1.05e9, d=768 M=24 1.44e9 / 1.37e9, K=32: d=128 M=8 5.8e9 / 5.5e9;
    // 12 / 20 / 24 floats: d=300 M=25 3.0e9 / 2.25e9, d=300 M=15 K=16 (the headline shape with 4-bit codes) 3.2e9 / 2.7e9,
    // d=768 M=32 1.47e9 / 1.15e9, d=300 M=15 K=32 2.57e9 / 2.57e9.
    if ((cb->variant == 10 || cb->variant == 0) && s16_fits) {
        SmallKArgs a;
        a.x = d_x; a.n = n; a.x_rs = x_rs; a.out = (uint8_t*)d_codes; a.o_rs = o_rs;
        a.cbt = cd.cbt; a.cc = cd.cc; a.cb = cd.cb;
        a.M = (int)cb->M; a.K = (int)cb->K; a.k_pad = cb->k_pad;
        // consecutive 64-row tiles per wave: the codebook image is staged once per workgroup, so as many as leave about
        // eight rounds of workgroups for the launch
        const int64_t tile_rows = small16_tile_rows((int)cb->dsub);
        const int64_t n_tiles = (n + tile_rows - 1) / tile_rows;
        const int64_t wg_slots = (int64_t)cb->ctx->devs[slot]->n_cus * 4 * 8;
        a.word_stores = (o_rs % 4 == 0 && ((uintptr_t)d_codes & 3) == 0) ? 1 : 0;
        a.tiles_per_wave = (int)std::max<int64_t>(1, std::min<int64_t>(kSmall16TilesMax, n_tiles / (4 * wg_slots)));
        const size_t lds = small16_lds_bytes(a.M, (int)cb->dsub, cb->KP);
        const dim3 grid((unsigned)((n_tiles + 4 * a.tiles_per_wave - 1) / (4 * a.tiles_per_wave)));
        if (!launch_small16(cb->KP, (int)cb->dsub, a, grid, lds, st)) return PQHIP_EUNSUPPORTED;
        HIPCHK(hipGetLastError());
        cb->last_kernel = "k_encode_small16";
        note_kernel("k_encode_small16");
        return PQHIP_OK;
    }
    if (cb->variant == 10) return PQHIP_EUNSUPPORTED;
    // Small codebooks: the VALU kernel reads x once, in whole row segments, and keeps the centroids on the scalar
    // path (kernels_smallk.hip.h).  Auto choice for K <= 16 with sub-vectors of <= 8 floats -- the reference's
    // own bench shape, d = 128, M = 16, K = 16: 6.3e9 vectors/s against 4.4e9 for the MFMA kernel; for wider
    // sub-vectors or K = 32 / 64 the MFMA kernels are still the faster ones (tools/smallk_sweep.sh) -- when the
    // host knows the norms are finite; variant 6 forces it for any K <= 64.
    if (((cb->variant == 0 && cb->KP == 16 && cb->dsub <= 8) || cb->variant == 6) && cb->KP != 0 && code_bytes == 1 && cb->norms_ok &&
        bad_flag == nullptr) {
        SmallKArgs a;
        a.x = d_x; a.n = n; a.x_rs = x_rs; a.out = (uint8_t*)d_codes; a.o_rs = o_rs;
        a.cbt = cd.cbt; a.cc = cd.cc; a.cb = cd.cb;
        a.M = (int)cb->M; a.K = (int)cb->K; a.k_pad = cb->k_pad;
        const dim3 grid((unsigned)((n + 255) / 256));
        if (!launch_smallk(cb->KP, (int)cb->dsub, a, grid, st)) return PQHIP_EUNSUPPORTED;
        HIPCHK(hipGetLastError());
        cb->last_kernel = "k_encode_smallk";
        note_kernel("k_encode_smallk");
        return PQHIP_OK;
    }
    if (cb->variant == 6) return PQHIP_EUNSUPPORTED;
    // MFMA kernels: u8 codes from every variant, u32 codes (k-means assignments, wide index types)
    // from the default variant; K <= 256 here (larger K: encode_grouped_dev above, or the anchor)
    const bool mfma_possible = !cb->wide && cb->groups == 1 && cb->T != 0 && (cb->norms_ok || bad_flag != nullptr) &&
                               (code_bytes == 1 || (code_bytes == 4 && (cb->variant == 0 || cb->variant == 4 || cb->variant == 9)));
    bool use_mfma = mfma_possible;
    if (cb->variant == 1) use_mfma = false;
    if (cb->variant >= 2 && !mfma_possible) return PQHIP_EUNSUPPORTED;

    if (use_mfma) {
        EncodeArgs a;
        a.x = d_x; a.n = n; a.x_rs = x_rs; a.out = d_codes; a.o_rs = o_rs;
        a.frags = cd.frags; a.cc = cd.cc; a.cb = cd.cb;
        a.M = (int)cb->M; a.K = (int)cb->K; a.dsub = (int)cb->dsub; a.k_pad = cb->k_pad;
        a.groups = 1;
        a.bad_flag = bad_flag;
        // kernel kind: 0 VALU argmin, 2 LDS argmin + LDS A fragments, 3 the same epilogue on 16x16x4
        // auto: for sub-vectors of <= 2 floats the per-distance work outweighs the MFMA chain and the
        // LDS pipe (one atomic per 64 distances) becomes the bound: the VALU-argmin kernel is 4-20 % faster
        // (round 3: with the hybrid lane-local + LDS argmin of the default kernel, 4-float sub-vectors moved to the default:
        // d=300 M=75 2.98e8 vs 2.80e8 vectors/s; 2-float ones stay here: M=150 1.62e8 vs 1.69e8, d=20 M=10 K=128 4.3e9 vs 5.1e9)
        const bool tiny = cb->variant == 0 && cb->DP <= 2 && code_bytes == 1;
        // the VALU-argmin kernel keeps all T * DP/2 fragments in registers: small codebooks only
        const bool kind0_fits = cb->DP <= 32 && cb->T * (cb->DP / 2) <= 128 && code_bytes == 1;
        if (cb->variant == 2 && !kind0_fits) return PQHIP_EUNSUPPORTED;
        // kind 3 (k_encode_mfma16: the same epilogue on v_mfma_f32_16x16x4_f32, four waves per SIMD) is instantiated for
        // >= 64 centroids and sub-vectors of 4, 8, .., 32 real floats; auto takes it where it wins on one box
        // (tools/mfma16_shapes.sh, profiles/r3_encode_experiments.md): K > 128 and 12..24 floats -- +2 % at 12 / 24, +2.5 % at 20,
        // +5 % at 16; shorter chains lose to the hybrid argmin of kind 2 (-15 % at 4 floats), 32 floats leave only 3 waves per
        // SIMD (-2.4 %), and with 64 / 128 centroids the per-tile work (norms, row loads, code bytes) weighs more (-1 .. -18 %)
        const bool no_mfma16 = diag().no_mfma16;
        const bool kind3_fits = cb->T >= 2 && cb->DP <= 32 && cb->DP % 4 == 0 && cb->DP == cb->dsub && (code_bytes == 1 || code_bytes == 4);
        // (beside_update: the k-means assignment step, whose update kernels run beside it on a second stream: with four encode
        // waves per SIMD the iteration was 2 % slower -- 20.4 vs 19.95 ms per 10 M rows -- so that caller stays on kind 2)
        const bool kind3_auto = kind3_fits && cb->T == 8 && cb->DP >= 12 && cb->DP <= 24 && !beside_update && !no_mfma16;
        if (cb->variant == 9 && !kind3_fits) return PQHIP_EUNSUPPORTED;
        const int kind = (cb->variant == 2 || tiny) ? 0 : (cb->variant == 9 || (cb->variant == 0 && kind3_auto)) ? 3 : 2;
        dim3 grid;
        if (kind >= 2) {
            // one workgroup = one subquantizer x 4 row streams (one per wave)
            const int64_t rpi_max = diag().rpi_max, rpi_min = diag().rpi_min;
            int64_t rpi = round_up((n * cb->M + 4 * 4096 - 1) / (4 * 4096), 32);
            rpi = std::max<int64_t>(rpi_min, std::min<int64_t>(rpi_max, rpi));
            if (kind == 3) rpi = std::min<int64_t>(rpi, 32 * kMfma16MaxTiles);   // one bit per row tile in the wave's exact-path mask
            a.rows_per_item = (int)rpi;
            a.n_chunks = (n + 4 * rpi - 1) / (4 * rpi);       // row groups
            a.chunks_per_xcd = (a.n_chunks + 7) / 8;
            grid = dim3((unsigned)(a.chunks_per_xcd * cb->M * 8));
        } else {
            // ~2 items per wave slot (256 CUs x 8 waves), 32..1024 rows each
            int64_t rpi = round_up((n * cb->M + 4095) / 4096, 32);
            rpi = std::max<int64_t>(32, std::min<int64_t>(1024, rpi));
            a.rows_per_item = (int)rpi;
            a.n_chunks = (n + rpi - 1) / rpi;
            a.chunks_per_xcd = (a.n_chunks + 7) / 8;
            const int64_t items_per_xcd = a.chunks_per_xcd * cb->M;
            const int64_t wgs_per_xcd = (items_per_xcd + 3) / 4;
            grid = dim3((unsigned)(wgs_per_xcd * 8));
        }
        // template flag: every one of the DP floats of a sub-vector is real (dsub == DP), or the last
        // one is padding (odd dsub).  Row alignment does not matter: the loads are dword-aligned wide loads.
        const bool vec = cb->DP == cb->dsub;
        const int grp = (cb->DP % 4 == 0) ? 4 : 2;
        StampRun stamps;      // (diagnostic builds: in-kernel s_memtime summary of the launch)
        PQCHK(stamps.begin(diag().enc_stamp && kind >= 2, (size_t)grid.x * 4 * 5, st));
        a.stamps = stamps.ptr();
        if (!launch_encode_mfma(kind, cb->T, cb->DP, vec, code_bytes, a, grid, st, diag().lds_pad)) return PQHIP_EUNSUPPORTED;
        PQCHK(stamps.report5(st, "encode", "steps", "seam"));
        static const char* const names[3][3] = {{"k_encode_mfma<odd>", "k_encode_mfma<vec2>", "k_encode_mfma<vec4>"},
                                                {"", "", ""},
                                                {"k_encode_mfma_lds3<odd>", "k_encode_mfma_lds3<vec2>", "k_encode_mfma_lds3<vec4>"}};
        cb->last_kernel = kind == 3 ? "k_encode_mfma16" : (!vec && cb->DP > 32) ? "k_encode_mfma_lds3<padded>" : names[kind][vec ? grp / 2 : 0];
        note_kernel(cb->last_kernel.load());
    } else {
        const int64_t total = n * cb->M;
        const int block = 256;
        const unsigned grid = (unsigned)std::min<int64_t>((total + block - 1) / block, 256 * 32);
        if (code_bytes == 1)
            hipLaunchKernelGGL((k_encode_scalar<uint8_t>), dim3(grid), dim3(block), 0, st, d_x, n,
                               x_rs, (uint8_t*)d_codes, o_rs, cd.cb, cd.cc, (int)cb->M, (int)cb->K,
                               (int)cb->dsub, cb->k_pad);
        else if (code_bytes == 4)
            hipLaunchKernelGGL((k_encode_scalar<uint32_t>), dim3(grid), dim3(block), 0, st, d_x, n,
                               x_rs, (uint32_t*)d_codes, o_rs, cd.cb, cd.cc, (int)cb->M, (int)cb->K,
                               (int)cb->dsub, cb->k_pad);
        else
            return PQHIP_EUNSUPPORTED;
        cb->last_kernel = "k_encode_scalar";
        note_kernel("k_encode_scalar");
    }
    HIPCHK(hipGetLastError());
    return PQHIP_OK;
}

}  // namespace pqh
