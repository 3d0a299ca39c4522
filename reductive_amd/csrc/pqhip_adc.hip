// pqhip_adc.hip -- asymmetric distance computation over a resident code matrix ("next" row):
// per-query lookup tables (linalg.rs:118-148 applied to the sub-vectors of the query) and the table-sum scans.
#include "pqhip_internal.h"

#include "kernels_adc.hip.h"

using namespace pqhip;

namespace pqh {

template <int NV>
int32_t launch_adc_nv(int nv, const uint8_t* codes, int64_t n, int64_t c_rs, const float* lut, int M, int K, float* out,
                      int n_cus, size_t lds, int* err, hipStream_t st)
{
    if constexpr (NV > kAdcMaxValueWords) {
        return PQHIP_EUNSUPPORTED;
    } else {
        if (nv != NV) return launch_adc_nv<NV + 1>(nv, codes, n, c_rs, lut, M, K, out, n_cus, lds, err, st);
        HIPCHK(hipFuncSetAttribute((const void*)k_adc_scan_u8<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        // all workgroups resident at once (occupancy API: registers and the LDS table both count), one contiguous
        // row range each -- the table is loaded once per workgroup
        const int adc_wgs = diag().adc_wgs;
        const int64_t max_wgs = (int64_t)n_cus * (adc_wgs ? adc_wgs : resident_wgs((const void*)k_adc_scan_u8<NV>, lds));
        int64_t rows_per_wg = round_up((n + max_wgs - 1) / max_wgs, 256);
        rows_per_wg = std::max<int64_t>(rows_per_wg, 1024);
        const unsigned grid = (unsigned)((n + rows_per_wg - 1) / rows_per_wg);
        hipLaunchKernelGGL((k_adc_scan_u8<NV>), dim3(grid), dim3(256), lds, st, codes, n, c_rs, lut, M, K, out, rows_per_wg, err);
        note_kernel("k_adc_scan_u8");
        return PQHIP_OK;
    }
}

template <int NV, int NQ>
int32_t launch_adc_mq(int nv, const uint8_t* codes, int64_t n, int64_t c_rs, const float* lut, int M, int K, float* out,
                      int64_t o_rs, int n_cus, size_t lds, int* err, hipStream_t st)
{
    if constexpr (NV > kAdcMaxValueWords) {
        return PQHIP_EUNSUPPORTED;
    } else {
        if (nv != NV) return launch_adc_mq<NV + 1, NQ>(nv, codes, n, c_rs, lut, M, K, out, o_rs, n_cus, lds, err, st);
        HIPCHK(hipFuncSetAttribute((const void*)k_adc_scan_u8_mq<NV, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        // one 1,024-thread workgroup per CU (the table image fills most of the LDS), one contiguous row range each
        int64_t rows_per_wg = round_up((n + n_cus - 1) / n_cus, 1024);
        rows_per_wg = std::max<int64_t>(rows_per_wg, 4096);
        const unsigned grid = (unsigned)((n + rows_per_wg - 1) / rows_per_wg);
        hipLaunchKernelGGL((k_adc_scan_u8_mq<NV, NQ>), dim3(grid), dim3(1024), lds, st, codes, n, c_rs, lut, M, K, out, o_rs, rows_per_wg, err);
        note_kernel(NQ == 8 ? "k_adc_scan_u8_mq<8 queries>" : "k_adc_scan_u8_mq<4 queries>");
        return PQHIP_OK;
    }
}

}  // namespace pqh

using namespace pqh;

extern "C" {

// ---- "next" row: asymmetric distance computation over a resident code matrix ----------------------
int32_t pqhip_adc_tables_f32_dev(pqhip_codebook* cb, int32_t slot, const float* d_q, int64_t nq, int64_t q_rs,
                                 float* d_tables, void* stream)
{
    if (!cb || nq < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (nq == 0) return PQHIP_OK;
    if (!d_q || !d_tables) return PQHIP_EINVAL;
    if (q_rs < cb->d) return PQHIP_ESHAPE;
    if (nq > (1 << 20)) return PQHIP_EUNSUPPORTED;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    CodebookDev& cd = cb->dev[slot];
    const float* y = d_q;
    int64_t y_rs = q_rs;
    ScratchLease rot(cb, slot, st);
    if (cb->has_proj) {       // pq.rs:293: the query is rotated like a vector to be quantized
        PQCHK(rot.acquire((size_t)nq * cb->d * sizeof(float)));
        const int64_t total = nq * cb->d;
        hipLaunchKernelGGL(k_adc_rotate_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_q, q_rs,
                           (int)nq, cd.P, (int)cb->d, (float*)rot.ptr());
        note_kernel("k_adc_rotate_queries");
        y = (const float*)rot.ptr();
        y_rs = cb->d;
    }
    const int64_t total = nq * cb->M * cb->K;
    hipLaunchKernelGGL(k_adc_tables, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, y_rs, (int)nq, cd.cb,
                       cd.cc, (int)cb->M, (int)cb->K, (int)cb->dsub, cb->k_pad, d_tables);
    HIPCHK(hipGetLastError());
    note_kernel("k_adc_tables");
    return PQHIP_OK;
}


int32_t pqhip_adc_scan_f32_dev(pqhip_codebook* cb, int32_t slot, const float* d_tables, int64_t nq, const void* d_codes,
                               int32_t code_bytes, int64_t n, int64_t c_rs, float* d_out, int64_t o_rs, void* stream)
{
    if (!cb || nq < 0 || n < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (code_bytes != 1 && code_bytes != 4) return PQHIP_EUNSUPPORTED;
    if (nq == 0 || n == 0) return PQHIP_OK;
    if (!d_tables || !d_codes || !d_out) return PQHIP_EINVAL;
    if (c_rs < cb->M || o_rs < n) return PQHIP_ESHAPE;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    ErrFlag ef(cb, slot, st);
    int* err = ef.flag;
    const int M = (int)cb->M, K = (int)cb->K;
    const size_t lds = (size_t)M * K * sizeof(float);
    const int nv = (M + 3) / 4;
    const bool fast = code_bytes == 1 && lds <= 160 * 1024 && nv <= kAdcMaxValueWords;
    // several queries per pass over the code matrix: 8 (or 4) tables interleaved in LDS when they fit
    // (context option "adc_single_query" = 1: one pass per query, the round-2 form, for A/B)
    const bool mq_on = cb->ctx->opt.adc_single_query.load(std::memory_order_relaxed) == 0;   // option "adc_single_query"
    const bool adc_any = diag().adc_any;   // the generic kernel for 32-bit codes (A/B, diagnostic builds)
    int64_t q = 0;
    if (fast && mq_on) {
        const int n_cus = cb->ctx->devs[slot]->n_cus;
        for (int nqp : {8, 4}) {
            const size_t lds_q = lds * nqp;
            if (lds_q > 160 * 1024) continue;
            for (; q + nqp <= nq; q += nqp) {
                const float* lut = d_tables + q * (int64_t)M * K;
                float* out = d_out + q * o_rs;
                if (nqp == 8) PQCHK((launch_adc_mq<1, 8>(nv, (const uint8_t*)d_codes, n, c_rs, lut, M, K, out, o_rs, n_cus, lds_q, err, st)));
                else PQCHK((launch_adc_mq<1, 4>(nv, (const uint8_t*)d_codes, n, c_rs, lut, M, K, out, o_rs, n_cus, lds_q, err, st)));
                HIPCHK(hipGetLastError());
            }
        }
    }
    for (; q < nq; ++q) {
        const float* lut = d_tables + q * (int64_t)M * K;
        float* out = d_out + q * o_rs;
        if (fast) {
            PQCHK(launch_adc_nv<1>(nv, (const uint8_t*)d_codes, n, c_rs, lut, M, K, out, cb->ctx->devs[slot]->n_cus, lds, err, st));
        } else if (code_bytes == 4 && lds <= 160 * 1024 && !adc_any) {
            // 32-bit codes, table within LDS (K <= 2,048 at M = 15): one 1,024-thread workgroup per CU, contiguous row ranges
            const int n_cus = cb->ctx->devs[slot]->n_cus;
            const int64_t rows_per_wg = round_up((n + n_cus - 1) / n_cus, 1024);
            const unsigned grid = (unsigned)((n + rows_per_wg - 1) / rows_per_wg);
            HIPCHK(hipFuncSetAttribute((const void*)k_adc_scan_wide, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL(k_adc_scan_wide, dim3(grid), dim3(1024), lds, st, (const uint32_t*)d_codes, n, c_rs, lut, M, K, out, rows_per_wg, err);
            note_kernel("k_adc_scan_wide");
        } else {
            const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32);
            if (code_bytes == 1)
                hipLaunchKernelGGL((k_adc_scan_any<uint8_t>), dim3(grid), dim3(256), 0, st, (const uint8_t*)d_codes, n, c_rs, lut, M, K, out, err);
            else
                hipLaunchKernelGGL((k_adc_scan_any<uint32_t>), dim3(grid), dim3(256), 0, st, (const uint32_t*)d_codes, n, c_rs, lut, M, K, out, err);
            note_kernel("k_adc_scan_any");
        }
        HIPCHK(hipGetLastError());
    }
    return PQHIP_OK;
}

}  // extern "C"
