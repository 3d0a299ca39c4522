// pqhip_opq.hip -- quantize / reconstruct / lookup on device-resident rows: the fused OPQ encode, the two-kernel
// OPQ paths through a leased scratch buffer, the codebook gather and its rotation-fused form, and the device entry
// points of include/pqhip.h (pq.rs:268-283, pq.rs:309-327, primitives.rs:110-173).
#include "pqhip_internal.h"

#include "kernels_gather.hip.h"
#include "kernels_gather_cg.hip.h"
#include "opq_fused2_launch.h"

using namespace pqhip;

namespace pqh {

// Workgroups of 256 threads of `kernel` that one CU holds at once with `lds` bytes of dynamic LDS (occupancy
// API; cached per thread for the last few (kernel, lds) pairs -- the query is a host-side table walk, but the
// small-batch path should not pay it per call).
int resident_wgs(const void* kernel, size_t lds)
{
    struct Ent { const void* k; size_t lds; int dev; int n; };
    thread_local Ent cache[8] = {};
    thread_local int next = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (const Ent& e : cache)
        if (e.k == kernel && e.lds == lds && e.dev == dev) return e.n;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, lds) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 4;
    }
    cache[next] = Ent{kernel, lds, dev, nb};
    next = (next + 1) % 8;
    return nb;
}

static int32_t gather_compact(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes, int64_t n, int64_t c_rs, float* d_out,
                              int64_t o_rs, hipStream_t st, int* err, const float* sel_scales, const int64_t* sel_rows = nullptr,
                              int64_t n_codes = 0, int64_t s_rs = 1);

int32_t gather_dev(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes, int64_t n,
                   int64_t c_rs, float* d_out, int64_t o_rs, hipStream_t st, int* err,
                   const int64_t* sel_rows, int64_t n_codes, const float* sel_scales, int64_t s_rs)
{
    if (n == 0) return PQHIP_OK;
    // Lookups into a resident matrix beyond the Infinity Cache (256 MB): two passes -- k_select_code_rows copies the selected
    // code rows and scales into a leased compact staging area, then this function runs again over it as a plain batch (with
    // per-row scales).  The random 15-byte reads miss the vector L1's TLB at that size, and a miss stalls the CU's whole
    // memory pipeline, the 1.2 KB-per-row store stream included (counters in kernels_gather.hip.h / profiles/r4_lookup_counters.json):
    // 100 M resident rows, 10 M lookups: 4.2 -> 3.2 ms (0.37 -> 0.48 of HBM); +2.5 % bytes (the staging round trip).  Option "lookup_two_pass":
    // 0 never, 1 always, default by size.
    if (sel_rows) {
        const int64_t opt = cb->ctx->opt.lookup_two_pass.load(std::memory_order_relaxed);
        const bool big = (double)n_codes * (double)c_rs * code_bytes > 256.0 * 1024 * 1024;
        if ((opt == 1 || (opt != 0 && big)) && (code_bytes == 1 || code_bytes == 4)) {
            const int64_t M = cb->M;
            // 1-byte codes with M <= 16: one thread per lookup writes an aligned 16-byte record (compact row stride 16)
            const bool rec16 = code_bytes == 1 && M <= 16 && c_rs >= M;
            const int64_t crs = rec16 ? 16 : M;
            const int64_t per_row = crs * code_bytes + 4;
            const int64_t chunk = std::max<int64_t>(4, std::min<int64_t>(n, (1ll << 30) / per_row)) & ~(int64_t)3;   // (scales first: codes stay 16-byte aligned)
            ScratchLease st_buf(cb, slot, st);
            PQCHK(st_buf.acquire((size_t)chunk * per_row + 64));
            float* sc = (float*)st_buf.ptr();                                  // [chunk] scales, then [chunk][crs] codes
            void* cc = (void*)(sc + chunk);
            const int64_t matrix_bytes = ((n_codes - 1) * c_rs + M) * code_bytes;
            for (int64_t r0 = 0; r0 < n; r0 += chunk) {
                const int64_t rows = std::min<int64_t>(chunk, n - r0);
                if (rec16) {
                    const unsigned g = (unsigned)std::min<int64_t>((rows + 255) / 256, (int64_t)cus_of(cb, slot) * 64);
                    hipLaunchKernelGGL(k_select_code_rows16, dim3(g), dim3(256), 0, st, (const uint8_t*)d_codes, c_rs, n_codes, matrix_bytes,
                                       sel_rows + r0, rows, (int)M, (uint8_t*)cc, sel_scales, s_rs, sc, err);
                } else {
                    const unsigned g = (unsigned)std::min<int64_t>((rows * M + 255) / 256, (int64_t)cus_of(cb, slot) * 64);
                    if (code_bytes == 1)
                        hipLaunchKernelGGL((k_select_code_rows<uint8_t>), dim3(g), dim3(256), 0, st, (const uint8_t*)d_codes, c_rs, n_codes,
                                           sel_rows + r0, rows, (int)M, (uint8_t*)cc, sel_scales, s_rs, sc, err);
                    else
                        hipLaunchKernelGGL((k_select_code_rows<uint32_t>), dim3(g), dim3(256), 0, st, (const uint32_t*)d_codes, c_rs, n_codes,
                                           sel_rows + r0, rows, (int)M, (uint32_t*)cc, sel_scales, s_rs, sc, err);
                }
                HIPCHK(hipGetLastError());
                note_kernel(rec16 ? "k_select_code_rows16" : "k_select_code_rows");
                PQCHK(gather_compact(cb, slot, cc, code_bytes, rows, crs, d_out + r0 * o_rs, o_rs, st, err, sel_scales ? sc : nullptr));
            }
            return PQHIP_OK;
        }
    }
    return gather_compact(cb, slot, d_codes, code_bytes, n, c_rs, d_out, o_rs, st, err, sel_scales, sel_rows, n_codes, s_rs);
}

// the reconstruct launch proper.  sel_rows == nullptr with sel_scales != nullptr: rows already selected (compact staging of
// the two-pass lookup), scale i belongs to row i.
static int32_t gather_compact(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes, int64_t n, int64_t c_rs, float* d_out,
                                   int64_t o_rs, hipStream_t st, int* err, const float* sel_scales, const int64_t* sel_rows, int64_t n_codes,
                                   int64_t s_rs)
{
    CodebookDev& cd = cb->dev[slot];
    const int d = (int)cb->d;
    const bool sel = sel_rows != nullptr || sel_scales != nullptr;            // the SEL form of the kernel
    // short sub-vectors (1 or 2 floats), K <= 256, 1-byte codes: centroids from LDS, a group of subquantizers per workgroup
    // (kernels_gather_cg.hip.h; four-float sub-vectors measured no faster there and stay below)
    if (code_bytes == 1 && cb->K <= 256 && (cb->dsub == 1 || cb->dsub == 2) && d % 4 == 0 && o_rs % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 && n < (1ll << 40)) {
        const int dsub = (int)cb->dsub, cpc = 4 / dsub;
        const size_t per_m = (size_t)cb->K * dsub * sizeof(float);
        // subquantizers per group: 64 KB of centroids (two workgroups per CU), whole 128-byte pieces of the output row where possible,
        // at most 64 chunks per row and group
        int64_t mg = (int64_t)((64 * 1024) / per_m);
        const int unit = 32 / dsub;
        mg = mg >= unit ? mg / unit * unit : mg / cpc * cpc;
        mg = std::min<int64_t>(std::min<int64_t>(mg, 256 / dsub), cb->M);
        const int n_groups = (int)((cb->M + mg - 1) / mg);
        const size_t lds = (size_t)mg * per_m;
        const int cpr = (int)(mg / cpc);
        RecCgArgs a;
        a.codes = (const uint8_t*)d_codes; a.n = n; a.c_rs = c_rs; a.out = d_out; a.o_rs = o_rs; a.cb = cd.cb;
        a.M = (int)cb->M; a.K = (int)cb->K; a.mg = (int)mg; a.n_groups = n_groups; a.err = err;
        a.sel_rows = sel_rows; a.n_codes = n_codes; a.sel_scales = sel_scales; a.s_rs = s_rs;
        // rows per workgroup: the staged centroids are paid once per workgroup -- about eight times their bytes in output, and no
        // fewer workgroups than fill the device twice
        int64_t rpw = round_up((int64_t)(8 * lds) / (cpr * 16), 64);
        rpw = std::max<int64_t>(256, std::min<int64_t>(4096, rpw));
        const int64_t fill = round_up(std::max<int64_t>(64, n * n_groups / ((int64_t)cus_of(cb, slot) * 4)), 64);
        rpw = std::min<int64_t>(rpw, fill);
        a.rows_per_wg = (int)rpw;
        a.sb_rows = std::max(1, 32768 / cpr);
        a.n_row_blocks = (n + rpw - 1) / rpw;
        const int64_t n_wg = ((a.n_row_blocks + 7) / 8) * 8 * n_groups;
        if (n_wg <= 0x7fffffffll) {
#define LAUNCH_CG(DS, SELT)                                                                                              \
    do {                                                                                                                 \
        if (lds > 48 * 1024)                                                                                             \
            HIPCHK(hipFuncSetAttribute((const void*)k_reconstruct_cg<DS, SELT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
        hipLaunchKernelGGL((k_reconstruct_cg<DS, SELT>), dim3((unsigned)n_wg), dim3(512), lds, st, a);                   \
    } while (0)
            if (dsub == 1) { if (sel) LAUNCH_CG(1, true); else LAUNCH_CG(1, false); }
            else { if (sel) LAUNCH_CG(2, true); else LAUNCH_CG(2, false); }
#undef LAUNCH_CG
            HIPCHK(hipGetLastError());
            note_kernel(sel_rows ? "k_reconstruct_cg<lookup>" : sel ? "k_reconstruct_cg<scaled>" : "k_reconstruct_cg");
            return PQHIP_OK;
        }
    }
    // 16-byte output chunks whenever a row is a whole number of them (the stores are dword-aligned
    // wide stores, so neither the row stride nor the base address matters); a chunk is filled with
    // one, two or four codebook accesses depending on how sub-vectors line up with it
    const bool vec = d % 4 == 0;
    // (gsz 0, odd sub-vectors of >= 5 floats: one unaligned 16-byte access per chunk that lies inside a sub-vector,
    // element-wise across a boundary -- 10 M x 300: dsub 15 4.10 -> 3.50 ms, dsub 5 5.30 -> 4.59 ms; even sub-vectors keep
    // two aligned 8-byte accesses per chunk, which is faster there: dsub 30 2.15 vs 3.09 ms.
    // diagnostic builds, PQHIP_DEBUG_REC_ELEMWISE=1: the per-element form, for A/B)
    const bool rec_elemwise = diag().rec_elemwise;
    const int gsz = !vec ? 1 : (cb->dsub % 4 == 0) ? 4 : (cb->dsub % 2 == 0) ? 2 : (cb->dsub > 4 && !rec_elemwise) ? 0 : 1;
    const int cpr = vec ? d / 4 : d;
    // rows per block: as many as keep rows*cpr < 2^16 (so that L / cpr == umulhi(L, ceil(2^32 / cpr))
    // exactly: the error term L * (inv * cpr - 2^32) stays below 2^32) and the block's codes within
    // 16 elements per thread (<= 64)
    int rows_per_block = 64;
    while (rows_per_block > 1 &&
           ((int64_t)rows_per_block * cpr >= 65536 || (int64_t)rows_per_block * cb->M > 256 * 16))
        rows_per_block /= 2;
    const unsigned inv_cpr = (cpr == 1) ? 0u : (unsigned)(((1ull << 32) + cpr - 1) / cpr);
    if (cb->M > 256 * 16 || (int64_t)cpr >= 65536) {
        const unsigned g = (unsigned)std::min<int64_t>((n * d + 255) / 256, 256 * 32);
        if (code_bytes == 1)
            hipLaunchKernelGGL((k_reconstruct_any<uint8_t>), dim3(g), dim3(256), 0, st, (const uint8_t*)d_codes, n, c_rs,
                               d_out, o_rs, cd.cb, (int)cb->M, (int)cb->K, (int)cb->dsub, err, sel_rows, n_codes, sel_scales, s_rs);
        else if (code_bytes == 4)
            hipLaunchKernelGGL((k_reconstruct_any<uint32_t>), dim3(g), dim3(256), 0, st, (const uint32_t*)d_codes, n, c_rs,
                               d_out, o_rs, cd.cb, (int)cb->M, (int)cb->K, (int)cb->dsub, err, sel_rows, n_codes, sel_scales, s_rs);
        else
            return PQHIP_EUNSUPPORTED;
        HIPCHK(hipGetLastError());
        note_kernel("k_reconstruct_any");
        return PQHIP_OK;
    }
    // grid = the workgroups that are RESIDENT at once (occupancy API x CUs), each owning one contiguous range of
    // row blocks.  Round 1 launched 8 per CU although the kernel's registers allow 4: the second half of the
    // ranges then ran as a second round behind the first, and the 100 M-row launch took 18.7 or 21.0 ms depending
    // on which allocation the output was (tools/rec_variance*.py).  Never more than 4 per CU for the
    // plain form, though: with the 4-element code prefetch 7 workgroups fit, and 1792 concurrent store streams
    // write slower than 1024 (100 M rows: 22.3 vs 19.4 ms on one box; a store-only kernel shows the same trend).
    // The lookup form (random source rows: workgroup times vary, reads matter) takes three rounds of its resident
    // count in shorter ranges (2.55 ms per 10 M rows against 2.8-2.9 with 4 or 8 per CU).
    // (Diagnostic builds: PQHIP_DEBUG_REC_WGS overrides the per-CU count.)
    const int rec_wgs_per_cu = diag().rec_wgs;
    const int64_t nblocks = (n + rows_per_block - 1) / rows_per_block;
    // small codebooks (plain form): the centroids are gathered from an LDS copy
    const size_t cb_bytes = (size_t)(cb->M * cb->K * cb->dsub) * sizeof(float);
    const bool cbl = !sel && cb_bytes <= 48 * 1024;
    const size_t lds = (((size_t)cpr * ((vec && gsz) ? 4 / gsz : 1) * sizeof(int) + 15) & ~(size_t)15) +
                       (((size_t)2 * rows_per_block * cb->M * code_bytes + 15) & ~(size_t)15) +
                       (sel ? (((size_t)2 * rows_per_block * sizeof(float) + 15) & ~(size_t)15) : 0) + (cbl ? cb_bytes : 0);
#define LAUNCH_REC3(IDX, V, GG, NEE)                                                              \
    do {                                                                                          \
        const int per_cu = rec_wgs_per_cu ? rec_wgs_per_cu                                        \
            : sel ? 3 * resident_wgs((const void*)k_reconstruct<IDX, V, true, GG, NEE>, lds) \
                       : std::min(4, resident_wgs((const void*)k_reconstruct<IDX, V, false, GG, NEE>, lds)); \
        const unsigned grid = (unsigned)std::min<int64_t>(nblocks, (int64_t)cus_of(cb, slot) * per_cu); \
        if (sel)                                                                                  \
            hipLaunchKernelGGL((k_reconstruct<IDX, V, true, GG, NEE>), dim3(grid), dim3(256), lds, st, \
                               (const IDX*)d_codes, n, c_rs, d_out, o_rs, cd.cb, (int)cb->M,      \
                               (int)cb->K, (int)cb->dsub, rows_per_block, inv_cpr, err,        \
                               sel_rows, n_codes, sel_scales, s_rs);                              \
        else if (cbl) {                                                                           \
            if (lds > 48 * 1024)                                                                  \
                HIPCHK(hipFuncSetAttribute((const void*)k_reconstruct<IDX, V, false, GG, NEE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
            hipLaunchKernelGGL((k_reconstruct<IDX, V, false, GG, NEE, true>), dim3(grid), dim3(256), lds, st, \
                               (const IDX*)d_codes, n, c_rs, d_out, o_rs, cd.cb, (int)cb->M,      \
                               (int)cb->K, (int)cb->dsub, rows_per_block, inv_cpr, err,        \
                               (const int64_t*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)1); \
        } else                                                                                    \
            hipLaunchKernelGGL((k_reconstruct<IDX, V, false, GG, NEE>), dim3(grid), dim3(256), lds, st, \
                               (const IDX*)d_codes, n, c_rs, d_out, o_rs, cd.cb, (int)cb->M,      \
                               (int)cb->K, (int)cb->dsub, rows_per_block, inv_cpr, err,        \
                               (const int64_t*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)1); \
    } while (0)
#define LAUNCH_REC2(IDX, V, GG)                                                                   \
    do {                                                                                          \
        if ((int64_t)rows_per_block * cb->M <= 256 * 4) LAUNCH_REC3(IDX, V, GG, 4);               \
        else LAUNCH_REC3(IDX, V, GG, 16);                                                         \
    } while (0)
#define LAUNCH_REC(IDX)                                                                           \
    do {                                                                                          \
        if (!vec) LAUNCH_REC2(IDX, 1, 1);                                                         \
        else if (gsz == 4) LAUNCH_REC2(IDX, 4, 4);                                                \
        else if (gsz == 0) LAUNCH_REC2(IDX, 4, 0);                                                \
        else if (gsz == 2) LAUNCH_REC2(IDX, 4, 2);                                                \
        else LAUNCH_REC2(IDX, 4, 1);                                                              \
    } while (0)
    if (code_bytes == 1) LAUNCH_REC(uint8_t);
    else if (code_bytes == 4) LAUNCH_REC(uint32_t);
    else return PQHIP_EUNSUPPORTED;
#undef LAUNCH_REC3
#undef LAUNCH_REC2
#undef LAUNCH_REC
    HIPCHK(hipGetLastError());
    note_kernel(sel_rows ? "k_reconstruct<lookup>" : sel ? "k_reconstruct<scaled>" : "k_reconstruct");
    return PQHIP_OK;
}

// Rows per chunk of the OPQ paths (rotation through a leased scratch buffer).  The rotation kernel runs one
// 12-wave workgroup per CU, (d / 64) column blocks x row groups of 4,608 rows, the column blocks of a row group on
// one XCD: a chunk whose workgroups fill every XCD's CUs a whole number of times leaves no partial last round.
// Measured on 10 M x 300 (rotate + encode, one box): 3.58 M rows (the 4 GiB cap: 15.3 rounds) 33.2 ms, 3.54 M
// (15 rounds) 32.5, 2.36 M (10) 32.2, 1.18 M (5) 32.0-32.2, 0.59 M (2.5 rounds) 34.8; one 12 GB chunk 33.1-33.5 ms.
static int64_t opq_chunk_rows(pqhip_codebook* cb, int slot, int64_t n)
{
    const int64_t dbg_rows = cb->ctx->opt.opq_scratch_rows.load(std::memory_order_relaxed);   // pqhip_ctx_set_option("opq_scratch_rows")
    const int64_t cap_rows = std::max<int64_t>(1, kScratchBytesMax / (cb->d * (int64_t)sizeof(float)));
    if (dbg_rows) return std::min<int64_t>(n, std::min<int64_t>(dbg_rows, cap_rows));
    const int ncb = (int)((cb->d + 63) / 64);
    const int slots_per_xcd = std::max(1, cb->ctx->devs[slot]->n_cus / 8);
    int g = slots_per_xcd, b = ncb;                 // gcd
    while (b) { const int t = g % b; g = b; b = t; }
    const int64_t unit_rg = 8ll * (slots_per_xcd / g);          // row groups per balanced unit (all 8 XCDs)
    const int64_t cap_rg = cap_rows / rot_rows_per_wg();
    const int64_t want_rg = 8ll * slots_per_xcd;                // 256 row groups = 1.18 M rows on a 256-CU device
    const int64_t chunk_rg = std::max<int64_t>(unit_rg, std::min<int64_t>(cap_rg, want_rg) / unit_rg * unit_rg);
    const int64_t rows = chunk_rg * rot_rows_per_wg();
    return std::min<int64_t>(n, std::min<int64_t>(rows, cap_rows));
}

int32_t quantize_dev_impl(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs,
                          void* d_codes, int code_bytes, int64_t o_rs, hipStream_t st)
{
    if (!cb->has_proj) return encode_plain_dev(cb, slot, d_x, n, x_rs, d_codes, code_bytes, o_rs, st);
    if (n == 0) return PQHIP_OK;
    CodebookDev& cd = cb->dev[slot];
    // OPQ (pq.rs:276) in ONE kernel where kernels_opq_fused2.hip.h is instantiated (P block AND codebook fragments in LDS,
    // x straight from global memory, the rotated rows never leave the register file: no scratch buffer, no chunk loop).
    // Needs u8 codes from a codebook with finite norms and 16-byte aligned rows.  Encode variant 8 forces it; the context
    // option "opq_fused" = 0 (or PQHIP_FUSED2_OPQ=0) keeps the two-kernel path.
    {
        const int DP = (int)cb->dsub;
        // (same-box A/B, 10 M x 300: 29.95 vs 30.57 ms in steady state, and the HBM traffic of a step drops from 3.5x to
        // ~1x the algorithmic bytes)
        const bool fused2_off = cb->ctx->opt.opq_fused.load(std::memory_order_relaxed) == 0;   // option "opq_fused" / PQHIP_FUSED2_OPQ=0
        const bool want2 = cb->variant == 8 || (cb->variant == 0 && !fused2_off);
        const bool vec = (cb->d % 4 == 0) && (x_rs % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_x) & 15) == 0);
        if (want2 && code_bytes == 1 && cb->groups == 1 && cb->T != 0 && cb->norms_ok && cb->dsub % 2 == 0 && cb->dsub <= 32 && vec &&
            opq_fused2_has(DP, cb->T, (int)cb->d)) {
            OpqFusedArgs a;
            a.x = d_x; a.n = n; a.x_rs = x_rs; a.P = cd.P; a.d = (int)cb->d;
            a.frags = cd.frags; a.cc = cd.cc; a.cb = cd.cb;
            a.out = (uint8_t*)d_codes; a.o_rs = o_rs;
            a.M = (int)cb->M; a.K = (int)cb->K; a.k_pad = cb->k_pad;
            const int nm = opq_fused2_slots(DP, cb->T, (int)cb->d) / DP;      // sub-vectors per column block (64 or 32 slots)
            a.ncb = (int)((cb->M + nm - 1) / nm);
            // tiles of 32 rows per wave: as many as leave ~8 rounds of workgroups (one per CU) for the whole launch, 4 .. 96
            // (10 M x 300, one box: 12 tiles 29.84 ms, 24: 29.57, 48: 29.40, 96: 29.24, 160: 30.6, 192 (4 rounds): 38.8 --
            // the P block and three fragment sets, 138 KB, are staged once per workgroup)
            const int f2_tiles_env = diag().fused2_tiles;
            const int64_t want_rg = std::max<int64_t>(1, 8ll * cb->ctx->devs[slot]->n_cus / a.ncb);
            const int f2_tiles = f2_tiles_env ? f2_tiles_env : (int)std::max<int64_t>(4, std::min<int64_t>(96, (n / want_rg + 255) / 256));
            a.rows_per_wg = 8 * 32 * f2_tiles;              // 8 waves x f2_tiles tiles of 32 rows
            const int64_t n_rg = (n + a.rows_per_wg - 1) / a.rows_per_wg;
            a.rg_per_xcd = (n_rg + 7) / 8;
            const dim3 grid((unsigned)(a.rg_per_xcd * a.ncb * 8));
            StampRun stamps;      // (diagnostic builds: in-kernel s_memtime summary of the launch)
            PQCHK(stamps.begin(diag().fused_stamp, (size_t)grid.x * 8 * 5, st));
            a.stamps = stamps.ptr();
            const int e = launch_opq_fused2(DP, cb->T, a, grid, st);
            if (e != 0) { g_hip_err = std::string("k_opq_encode_fused2: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "no instantiation"); return PQHIP_EHIP; }
            PQCHK(stamps.report5(st, "fused2", "rotation", "encode"));
            note_kernel("k_opq_encode_fused2");
            cb->last_kernel = "k_opq_encode_fused2";
            return PQHIP_OK;
        }
        if (cb->variant == 8) return PQHIP_EUNSUPPORTED;
    }
    // otherwise: rx = x.dot(P) into a leased scratch buffer, chunked, then PQ encode of rx
    const int64_t chunk = opq_chunk_rows(cb, slot, n);
    ScratchLease rx(cb, slot, st);
    PQCHK(rx.acquire((size_t)chunk * cb->d * sizeof(float)));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min<int64_t>(chunk, n - r0);
        PQCHK(rotate_dev(d_x + r0 * x_rs, rows, x_rs, cd.P, (int)cb->d, (float*)rx.ptr(), cb->d, st));
        PQCHK(encode_plain_dev(cb, slot, (const float*)rx.ptr(), rows, cb->d,
                               (char*)d_codes + r0 * o_rs * code_bytes, code_bytes, o_rs, st));
    }
    return PQHIP_OK;
}

// sel_rows != nullptr: lookup form -- output row i reconstructs code row sel_rows[i] (and is scaled
// by sel_scales[sel_rows[i]] when given); d_codes is then the whole resident [n_codes][M] matrix.
int32_t reconstruct_dev_impl(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes,
                             int64_t n, int64_t c_rs, float* d_out, int64_t o_rs, hipStream_t st,
                             const int64_t* sel_rows, int64_t n_codes, const float* sel_scales, int64_t s_rs)
{
    if (n == 0) return PQHIP_OK;
    ErrFlag ef(cb, slot, st);     // (its destructor marks the end of this call's flag-raising launches on `st`)
    int* err = ef.flag;
    if (!cb->has_proj)
        return gather_dev(cb, slot, d_codes, code_bytes, n, c_rs, d_out, o_rs, st, err, sel_rows, n_codes, sel_scales, s_rs);
    CodebookDev& cd = cb->dev[slot];
    // OPQ (pq.rs:323-326) in ONE kernel when the rotation kernel can gather (sub-vectors of whole 16-byte pieces, P block
    // within LDS): the reconstructed rows never exist in memory, no scratch buffer.  Context option "opq_gather_rotation" = 0: the
    // round-2 form below (gather -> scratch -> rotate), for A/B and for the chunk-loop tests.
    const bool fused_off = cb->ctx->opt.opq_gather_rotation.load(std::memory_order_relaxed) == 0;   // option "opq_gather_rotation"
    const int64_t code_rows = sel_rows ? n_codes : n;
    if (!fused_off && code_bytes == 1 && cb->dsub % 4 == 0 && cb->d < 65536 && cb->M * cb->K * cb->dsub < (1 << 24) &&
        code_rows * c_rs < (1ll << 32)) {
        RotGather ga;
        ga.codes = (const uint8_t*)d_codes; ga.c_rs = c_rs; ga.cb = cd.cb; ga.K = (int)cb->K; ga.dsub = (int)cb->dsub;
        ga.inv_dsub = (unsigned)(((1ull << 32) + cb->dsub - 1) / cb->dsub);
        ga.sel_rows = sel_rows; ga.n_codes = n_codes; ga.err = err;
        const int32_t rc = rotate_dev(nullptr, n, 0, cd.PT, (int)cb->d, d_out, o_rs, st, &ga);
        if (rc == PQHIP_OK) {
            if (sel_rows && sel_scales) {
                const unsigned g = (unsigned)std::min<int64_t>((n * cb->d + 255) / 256, 256 * 32);
                hipLaunchKernelGGL(k_scale_rows, dim3(g), dim3(256), 0, st, d_out, n, (int)cb->d, o_rs, sel_rows, n_codes, sel_scales, s_rs);
                HIPCHK(hipGetLastError());
                note_kernel("k_scale_rows");
            }
            return PQHIP_OK;
        }
        if (rc != PQHIP_EUNSUPPORTED) return rc;
    }
    // otherwise: gather into a leased scratch buffer, then out = r.dot(P^T); a lookup's scale comes last
    const int64_t chunk = opq_chunk_rows(cb, slot, n);
    ScratchLease rec(cb, slot, st);
    PQCHK(rec.acquire((size_t)chunk * cb->d * sizeof(float)));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min<int64_t>(chunk, n - r0);
        if (sel_rows)
            PQCHK(gather_dev(cb, slot, d_codes, code_bytes, rows, c_rs, (float*)rec.ptr(), cb->d, st, err, sel_rows + r0, n_codes, nullptr));
        else
            PQCHK(gather_dev(cb, slot, (const char*)d_codes + r0 * c_rs * code_bytes, code_bytes, rows,
                             c_rs, (float*)rec.ptr(), cb->d, st, err));
        PQCHK(rotate_dev((const float*)rec.ptr(), rows, cb->d, cd.PT, (int)cb->d, d_out + r0 * o_rs, o_rs, st));
        if (sel_rows && sel_scales) {
            const unsigned g = (unsigned)std::min<int64_t>((rows * cb->d + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(k_scale_rows, dim3(g), dim3(256), 0, st, d_out + r0 * o_rs, rows, (int)cb->d, o_rs,
                               sel_rows + r0, n_codes, sel_scales, s_rs);
            HIPCHK(hipGetLastError());
            note_kernel("k_scale_rows");
        }
    }
    return PQHIP_OK;
}

// 2- and 8-byte codes (u16 / u64 / usize index types, traits.rs:77-88): the kernels work on u8 (K <= 256) or u32 codes in a
// leased scratch matrix, k_convert_codes widens / narrows between it and the caller's matrix, chunk by chunk.
template <typename Fn>
static int32_t with_converted_codes(pqhip_codebook* cb, int slot, int64_t n, hipStream_t st, int dev_bytes, Fn fn)
{
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (1ll << 30) / (cb->M * dev_bytes)));
    ScratchLease tmp(cb, slot, st);
    PQCHK(tmp.acquire((size_t)chunk * cb->M * dev_bytes));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) PQCHK(fn(tmp.ptr(), r0, std::min<int64_t>(chunk, n - r0)));
    return PQHIP_OK;
}

template <typename Src, typename Dst>
static int32_t convert_codes(const void* src, int64_t s_rs, void* dst, int64_t d_rs, int64_t rows, int M, uint64_t K, int* err, hipStream_t st)
{
    const unsigned g = (unsigned)std::min<int64_t>((rows * M + 255) / 256, 256 * 32);
    hipLaunchKernelGGL((k_convert_codes<Src, Dst>), dim3(g), dim3(256), 0, st, (const Src*)src, s_rs, (Dst*)dst, d_rs, rows, M,
                       (unsigned long long)K, err);
    HIPCHK(hipGetLastError());
    note_kernel("k_convert_codes");
    return PQHIP_OK;
}

}  // namespace pqh

using namespace pqh;

extern "C" {

// ---- device-resident entry points -------------------------------------------------------------
int32_t pqhip_quantize_batch_f32_dev(pqhip_codebook* cb, int32_t slot, const float* d_x, int64_t n,
                                     int64_t x_rs, void* d_codes, int32_t code_bytes, int64_t o_rs,
                                     void* stream)
{
    if (!cb || n < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_x || !d_codes)) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4 && code_bytes != 8) return PQHIP_EUNSUPPORTED;
    // primitives.rs:31-34 "Cannot store centroids in quantizer index type"
    if (code_bytes < 8 && (uint64_t)(cb->K - 1) > ((1ull << (8 * code_bytes)) - 1)) return PQHIP_EINDEX_WIDTH;
    if (n > 0 && (x_rs < cb->d || o_rs < cb->M)) return PQHIP_ESHAPE;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1 || code_bytes == 4) return quantize_dev_impl(cb, slot, d_x, n, x_rs, d_codes, code_bytes, o_rs, st);
    if (n == 0) return PQHIP_OK;
    const int dev_bytes = cb->K <= 256 ? 1 : 4;
    const int M = (int)cb->M;
    return with_converted_codes(cb, slot, n, st, dev_bytes, [&](void* tmp, int64_t r0, int64_t rows) -> int32_t {
        PQCHK(quantize_dev_impl(cb, slot, d_x + r0 * x_rs, rows, x_rs, tmp, dev_bytes, M, st));
        char* dst = (char*)d_codes + r0 * o_rs * code_bytes;
        if (code_bytes == 2) return dev_bytes == 1 ? convert_codes<uint8_t, uint16_t>(tmp, M, dst, o_rs, rows, M, 0, nullptr, st)
                                                   : convert_codes<uint32_t, uint16_t>(tmp, M, dst, o_rs, rows, M, 0, nullptr, st);
        return dev_bytes == 1 ? convert_codes<uint8_t, uint64_t>(tmp, M, dst, o_rs, rows, M, 0, nullptr, st)
                              : convert_codes<uint32_t, uint64_t>(tmp, M, dst, o_rs, rows, M, 0, nullptr, st);
    });
}

int32_t pqhip_reconstruct_batch_f32_dev(pqhip_codebook* cb, int32_t slot, const void* d_codes,
                                        int32_t code_bytes, int64_t n, int64_t c_rs, float* d_out,
                                        int64_t o_rs, void* stream)
{
    if (!cb || n < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_codes || !d_out)) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4 && code_bytes != 8) return PQHIP_EUNSUPPORTED;
    if (n > 0 && (c_rs < cb->M || o_rs < cb->d)) return PQHIP_ESHAPE;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1 || code_bytes == 4)
        return reconstruct_dev_impl(cb, slot, d_codes, code_bytes, n, c_rs, d_out, o_rs, st);
    if (n == 0) return PQHIP_OK;
    const int M = (int)cb->M;
    // (u32 scratch whatever K is: a 2- or 8-byte value >= K must raise the range flag, so it is clamped to K, which the
    // gather then reports -- K itself does not fit a byte when K = 256)
    return with_converted_codes(cb, slot, n, st, 4, [&](void* tmp, int64_t r0, int64_t rows) -> int32_t {
        const char* src = (const char*)d_codes + r0 * c_rs * code_bytes;
        {
            ErrFlag ef(cb, slot, st);
            if (code_bytes == 2) PQCHK((convert_codes<uint16_t, uint32_t>(src, c_rs, tmp, M, rows, M, (uint64_t)cb->K, ef.flag, st)));
            else PQCHK((convert_codes<uint64_t, uint32_t>(src, c_rs, tmp, M, rows, M, (uint64_t)cb->K, ef.flag, st)));
        }
        return reconstruct_dev_impl(cb, slot, tmp, 4, rows, M, d_out + r0 * o_rs, o_rs, st);
    });
}

int32_t pqhip_reconstruct_rows_f32_dev(pqhip_codebook* cb, int32_t slot, const void* d_codes,
                                       int32_t code_bytes, int64_t n_codes, int64_t c_rs,
                                       const int64_t* d_rows, int64_t n, const float* d_scales,
                                       float* d_out, int64_t o_rs, void* stream)
{
    if (!cb || n < 0 || n_codes < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_codes || !d_out || !d_rows)) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 4) return PQHIP_EUNSUPPORTED;
    if (n > 0 && (c_rs < cb->M || o_rs < cb->d)) return PQHIP_ESHAPE;
    if (n == 0) return PQHIP_OK;
    if (n_codes == 0) return PQHIP_ECODE_RANGE;  // every index is out of bounds
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    return reconstruct_dev_impl(cb, slot, d_codes, code_bytes, n, c_rs, d_out, o_rs, (hipStream_t)stream,
                                d_rows, n_codes, d_scales);
}

int32_t pqhip_reconstruct_rows_records_f32_dev(pqhip_codebook* cb, int32_t slot, const void* d_records, int32_t code_bytes,
                                               int64_t n_codes, int64_t record_bytes, int64_t scale_offset_bytes,
                                               const int64_t* d_rows, int64_t n, float* d_out, int64_t o_rs, void* stream)
{
    if (!cb || n < 0 || n_codes < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_records || !d_out || !d_rows)) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 4) return PQHIP_EUNSUPPORTED;
    // a record = M codes, padding, one f32 scale, padding: whole code elements and whole floats per record
    if (record_bytes <= 0 || record_bytes % 4 != 0 || record_bytes % code_bytes != 0 || scale_offset_bytes % 4 != 0 ||
        scale_offset_bytes < cb->M * code_bytes || scale_offset_bytes + 4 > record_bytes || (reinterpret_cast<uintptr_t>(d_records) & 3))
        return PQHIP_ESHAPE;
    if (n > 0 && o_rs < cb->d) return PQHIP_ESHAPE;
    if (n == 0) return PQHIP_OK;
    if (n_codes == 0) return PQHIP_ECODE_RANGE;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    const float* scales = reinterpret_cast<const float*>(static_cast<const char*>(d_records) + scale_offset_bytes);
    return reconstruct_dev_impl(cb, slot, d_records, code_bytes, n, record_bytes / code_bytes, d_out, o_rs, (hipStream_t)stream,
                                d_rows, n_codes, scales, record_bytes / 4);
}

}  // extern "C"
