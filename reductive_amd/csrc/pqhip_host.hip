// pqhip_host.hip -- the host-resident entry points of libpqhip.so: one host batch is cut into contiguous row shards,
// one per device of the context (codebook replicated, no collective, results land in disjoint row ranges
// of the caller's arrays), and every shard streams through a leased staging set of its device -- pinned double-buffered
// H2D / D2H with the packing of strided caller rows spread over the device's host threads.
#include "pqhip_internal.h"

#include <cstdlib>

namespace pqh {

// run fn(slot, row_begin, row_end) for the contiguous row shard of every device
template <typename F>
int32_t for_each_shard(pqhip_ctx* ctx, int64_t n, F fn)
{
    const int nd = (int)ctx->devs.size();
    const int used = (int)std::max<int64_t>(1, std::min<int64_t>(nd, (n + 4095) / 4096));
    const int64_t per = (n + used - 1) / used;
    if (used == 1) return fn(0, (int64_t)0, n);
    std::vector<int32_t> rc(used, PQHIP_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < used; ++i) {
        const int64_t b = std::min<int64_t>(n, i * per), e = std::min<int64_t>(n, b + per);
        th.emplace_back([&, i, b, e] { rc[i] = fn(i, b, e); });
    }
    for (auto& t : th) t.join();
    for (int32_t r : rc)
        if (r != PQHIP_OK) return r;
    return PQHIP_OK;
}

// rows per pinned staging buffer for rows of `row_bytes` input bytes: kStageBytes worth of rows, at least
// kStageRowsMin of them only while that floor stays within 4 x kStageBytes (very wide rows: the byte cap wins,
// down to one row per buffer)
int64_t stage_rows(int64_t shard_rows, int64_t row_bytes)
{
    row_bytes = std::max<int64_t>(1, row_bytes);
    int64_t r = kStageBytes / row_bytes;
    if (r < kStageRowsMin) r = std::min<int64_t>(kStageRowsMin, std::max<int64_t>(1, 4 * kStageBytes / row_bytes));
    return std::max<int64_t>(1, std::min<int64_t>(r, shard_rows));
}

// the pinned staging buffers of a device slot are reused from call to call: an earlier call that returned on an error
// may have left copies in flight on the staging streams
int32_t quiesce_staging(StageSet& ss)
{
    HIPCHK(hipStreamSynchronize(ss.stream[0]));
    HIPCHK(hipStreamSynchronize(ss.stream[1]));
    return PQHIP_OK;
}

void store_code(void* base, int bytes, int64_t off, uint32_t v)
{
    switch (bytes) {
    case 1: ((uint8_t*)base)[off] = (uint8_t)v; break;
    case 2: ((uint16_t*)base)[off] = (uint16_t)v; break;
    case 4: ((uint32_t*)base)[off] = v; break;
    default: ((uint64_t*)base)[off] = v; break;
    }
}

uint64_t load_code(const void* base, int bytes, int64_t off)
{
    switch (bytes) {
    case 1: return ((const uint8_t*)base)[off];
    case 2: return ((const uint16_t*)base)[off];
    case 4: return ((const uint32_t*)base)[off];
    default: return ((const uint64_t*)base)[off];
    }
}

}  // namespace pqh

using namespace pqh;

extern "C" {

// ---- host-resident entry points ---------------------------------------------------------------
int32_t pqhip_quantize_batch_f32(pqhip_codebook* cb, const float* x, int64_t n, int64_t x_rs,
                                 int64_t x_cs, void* codes, int32_t code_bytes, int64_t o_rs,
                                 int64_t o_cs)
{
    if (!cb || n < 0) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4 && code_bytes != 8) return PQHIP_EINVAL;
    if (n == 0) return PQHIP_OK;
    if (!x || !codes) return PQHIP_EINVAL;
    // primitives.rs:31-34 "Cannot store centroids in quantizer index type"
    if (code_bytes < 8 && (uint64_t)(cb->K - 1) > ((1ull << (8 * code_bytes)) - 1)) return PQHIP_EINDEX_WIDTH;
    const int dev_bytes = cb->K <= 256 ? 1 : 4;
    const int64_t d = cb->d, M = cb->M;

    return for_each_shard(cb->ctx, n, [&](int slot, int64_t rb, int64_t re) -> int32_t {
        DeviceSlot& ds = *cb->ctx->devs[slot];
        StageLease lease(ds);
        StageSet& ss = *lease.s;
        SET_DEVICE(ds.ordinal);
        PQCHK(quiesce_staging(ss));
        const int64_t cap = stage_rows(re - rb, d * (int64_t)sizeof(float));
        // Zero-copy leg (opt-in: PQHIP_HOST_ZERO_COPY=1; unit column stride): the caller's rows are page-locked in place,
        // chunk by chunk, and the DMA engine reads them with a 2-D copy (row pitch = the caller's row stride) -- no
        // pageable -> pinned memcpy, which doubles the host-memory traffic of the call.  Registration of chunk k + 1 runs
        // on this host thread while chunk k is copied and encoded.  Measured (tools/mb_hostreg.hip, bench.py
        // --in-process 1, one box): hipHostRegister of fresh pages runs at 21-50 GB/s from one thread and does not
        // scale with threads, the copy from registered pages at the full 57.5 GB/s; end to end 3.77e7 vectors/s
        // (45.8 GB/s) against 4.25e7 (51.6 GB/s) for 16 packing threads + pinned staging -- so packing stays the
        // default on a host that has 16 cores per GPU to spend, and this leg is for hosts where memory bandwidth or
        // cores are the scarce resource (8 GPUs x 2 x 57 GB/s of packing traffic).  Any failure to register falls back
        // to the packing path for that chunk.
        // Registrations never overlap: a chunk page-locks only the WHOLE pages inside its own byte range --
        // consecutive chunks and neighbouring shards share their boundary pages, and a page registered twice fails -- and the
        // few rows that touch a boundary page go through the pinned staging buffer like any packed row.  A registration is
        // released only after its stream has been synchronised, on the error paths too.
        static const bool zero_copy_on = [] { const char* e = getenv("PQHIP_HOST_ZERO_COPY"); return e && e[0] == '1'; }();
        const bool zero_copy = zero_copy_on && x_cs == 1 && x_rs >= d;
        for (int b = 0; b < 2; ++b)
            PQCHK(ensure_staging(ss.st[b], (size_t)cap * d * sizeof(float), (size_t)cap * M * dev_bytes));
        void* reg_ptr[2] = {nullptr, nullptr};
        struct Unreg {
            void** p; StageSet* ss;
            ~Unreg()
            {
                for (int i = 0; i < 2; ++i)
                    if (p[i]) { (void)hipStreamSynchronize(ss->stream[i]); (void)hipHostUnregister(p[i]); (void)hipGetLastError(); }
            }
        } unreg{reg_ptr, &ss};
        auto drain = [&](int b, int64_t r0, int64_t rows) -> int32_t {
            HIPCHK(hipStreamSynchronize(ss.stream[b]));
            if (reg_ptr[b]) { (void)hipHostUnregister(reg_ptr[b]); reg_ptr[b] = nullptr; }
            const uint8_t* h8 = (const uint8_t*)ss.st[b].h_out;
            const uint32_t* h32 = (const uint32_t*)ss.st[b].h_out;
            lease.pool->run(rows, [&, r0, h8, h32](int64_t ib, int64_t ie) {
                if (code_bytes == dev_bytes && o_cs == 1) {           // same width, unit column stride: row copies
                    char* dst = (char*)codes;
                    const char* src = (const char*)ss.st[b].h_out;
                    const size_t rb_ = (size_t)M * dev_bytes;
                    if (o_rs == M) std::memcpy(dst + (size_t)(r0 + ib) * rb_, src + (size_t)ib * rb_, (size_t)(ie - ib) * rb_);
                    else
                        for (int64_t i = ib; i < ie; ++i)
                            std::memcpy(dst + (size_t)(r0 + i) * o_rs * dev_bytes, src + (size_t)i * rb_, rb_);
                } else {
                    for (int64_t i = ib; i < ie; ++i)
                        for (int64_t m = 0; m < M; ++m) {
                            const uint32_t v = dev_bytes == 1 ? h8[i * M + m] : h32[i * M + m];
                            store_code(codes, code_bytes, (r0 + i) * o_rs + m * o_cs, v);
                        }
                }
            });
            return PQHIP_OK;
        };
        int64_t pend_r0[2] = {0, 0}, pend_rows[2] = {0, 0};
        int b = 0;
        for (int64_t r0 = rb; r0 < re; r0 += cap, b ^= 1) {
            const int64_t rows = std::min<int64_t>(cap, re - r0);
            if (pend_rows[b]) { PQCHK(drain(b, pend_r0[b], pend_rows[b])); pend_rows[b] = 0; }
            bool direct = false;
            if (zero_copy) {
                constexpr uintptr_t kPage = 4096;
                const float* src = x + r0 * x_rs;
                const uintptr_t lo = reinterpret_cast<uintptr_t>(src);
                const uintptr_t hi = lo + ((size_t)(rows - 1) * x_rs + d) * sizeof(float);
                const uintptr_t in_lo = (lo + kPage - 1) & ~(kPage - 1), in_hi = hi & ~(kPage - 1);   // whole pages of this chunk only
                const size_t row_b = (size_t)x_rs * sizeof(float), data_b = (size_t)d * sizeof(float);
                // rows [i0, i1) lie entirely inside the registered pages; the edge rows are packed
                const int64_t i0 = in_lo > lo ? (int64_t)((in_lo - lo + row_b - 1) / row_b) : 0;
                const int64_t i1 = in_hi >= lo + data_b ? std::min<int64_t>(rows, (int64_t)((in_hi - lo - data_b) / row_b) + 1) : 0;
                if (in_hi > in_lo && i1 - i0 >= 64 &&
                    hipHostRegister(reinterpret_cast<void*>(in_lo), in_hi - in_lo, hipHostRegisterDefault) == hipSuccess) {
                    reg_ptr[b] = reinterpret_cast<void*>(in_lo);
                    float* hin = (float*)ss.st[b].h_in;
                    float* din = (float*)ss.st[b].d_in;
                    hipError_t e = hipMemcpy2DAsync(din + i0 * d, data_b, src + i0 * x_rs, row_b, data_b, (size_t)(i1 - i0),
                                                    hipMemcpyHostToDevice, ss.stream[b]);
                    for (int64_t i = 0; i < i0; ++i) std::memcpy(hin + i * d, src + i * x_rs, data_b);
                    for (int64_t i = i1; i < rows; ++i) std::memcpy(hin + i * d, src + i * x_rs, data_b);
                    if (e == hipSuccess && i0 > 0)
                        e = hipMemcpyAsync(din, hin, (size_t)i0 * data_b, hipMemcpyHostToDevice, ss.stream[b]);
                    if (e == hipSuccess && i1 < rows)
                        e = hipMemcpyAsync(din + i1 * d, hin + i1 * d, (size_t)(rows - i1) * data_b, hipMemcpyHostToDevice, ss.stream[b]);
                    if (e == hipSuccess) direct = true;
                    else {   // nothing may still read the pages when they are released; then the packing path redoes the chunk
                        (void)hipGetLastError();
                        (void)hipStreamSynchronize(ss.stream[b]);
                        (void)hipHostUnregister(reg_ptr[b]);
                        (void)hipGetLastError();
                        reg_ptr[b] = nullptr;
                    }
                } else {
                    (void)hipGetLastError();
                }
            }
            if (!direct) {
                float* hin = (float*)ss.st[b].h_in;
                lease.pool->run(rows, [&, r0, hin](int64_t ib, int64_t ie) {
                    if (x_cs == 1 && x_rs == d) {
                        std::memcpy(hin + ib * d, x + (r0 + ib) * d, (size_t)(ie - ib) * d * sizeof(float));
                    } else if (x_cs == 1) {
                        for (int64_t i = ib; i < ie; ++i)
                            std::memcpy(hin + i * d, x + (r0 + i) * x_rs, (size_t)d * sizeof(float));
                    } else {
                        for (int64_t i = ib; i < ie; ++i)
                            for (int64_t k = 0; k < d; ++k) hin[i * d + k] = x[(r0 + i) * x_rs + k * x_cs];
                    }
                });
                HIPCHK(hipMemcpyAsync(ss.st[b].d_in, hin, (size_t)rows * d * sizeof(float),
                                      hipMemcpyHostToDevice, ss.stream[b]));
            }
            PQCHK(quantize_dev_impl(cb, slot, (const float*)ss.st[b].d_in, rows, d, ss.st[b].d_out,
                                    dev_bytes, M, ss.stream[b]));
            HIPCHK(hipMemcpyAsync(ss.st[b].h_out, ss.st[b].d_out, (size_t)rows * M * dev_bytes,
                                  hipMemcpyDeviceToHost, ss.stream[b]));
            pend_r0[b] = r0; pend_rows[b] = rows;
        }
        for (int k = 0; k < 2; ++k)
            if (pend_rows[k]) PQCHK(drain(k, pend_r0[k], pend_rows[k]));
        return PQHIP_OK;
    });
}

int32_t pqhip_reconstruct_batch_f32(pqhip_codebook* cb, const void* codes, int32_t code_bytes,
                                    int64_t n, int64_t c_rs, int64_t c_cs, float* out,
                                    int64_t o_rs, int64_t o_cs)
{
    if (!cb || n < 0) return PQHIP_EINVAL;
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4 && code_bytes != 8) return PQHIP_EINVAL;
    if (n == 0) return PQHIP_OK;
    if (!codes || !out) return PQHIP_EINVAL;
    const int dev_bytes = code_bytes == 1 ? 1 : 4;
    const int64_t d = cb->d, M = cb->M;

    return for_each_shard(cb->ctx, n, [&](int slot, int64_t rb, int64_t re) -> int32_t {
        DeviceSlot& ds = *cb->ctx->devs[slot];
        StageLease lease(ds);
        StageSet& ss = *lease.s;
        SET_DEVICE(ds.ordinal);
        PQCHK(quiesce_staging(ss));
        const int64_t cap = stage_rows(re - rb, d * (int64_t)sizeof(float));   // sized by the OUTPUT rows here
        for (int b = 0; b < 2; ++b)
            PQCHK(ensure_staging(ss.st[b], (size_t)cap * M * dev_bytes, (size_t)cap * d * sizeof(float)));
        auto drain = [&](int b, int64_t r0, int64_t rows) -> int32_t {
            HIPCHK(hipStreamSynchronize(ss.stream[b]));
            const float* h = (const float*)ss.st[b].h_out;
            lease.pool->run(rows, [&, r0, h](int64_t ib, int64_t ie) {
                if (o_cs == 1 && o_rs == d) {
                    std::memcpy(out + (r0 + ib) * d, h + ib * d, (size_t)(ie - ib) * d * sizeof(float));
                } else if (o_cs == 1) {
                    for (int64_t i = ib; i < ie; ++i)
                        std::memcpy(out + (r0 + i) * o_rs, h + i * d, (size_t)d * sizeof(float));
                } else {
                    for (int64_t i = ib; i < ie; ++i)
                        for (int64_t k = 0; k < d; ++k) out[(r0 + i) * o_rs + k * o_cs] = h[i * d + k];
                }
            });
            return PQHIP_OK;
        };
        int64_t pend_r0[2] = {0, 0}, pend_rows[2] = {0, 0};
        int b = 0;
        std::atomic<bool> range_err{false};
        for (int64_t r0 = rb; r0 < re; r0 += cap, b ^= 1) {
            const int64_t rows = std::min<int64_t>(cap, re - r0);
            if (pend_rows[b]) { PQCHK(drain(b, pend_r0[b], pend_rows[b])); pend_rows[b] = 0; }
            void* hin = ss.st[b].h_in;
            lease.pool->run(rows, [&, r0, hin](int64_t ib, int64_t ie) {
                bool bad = false;
                for (int64_t i = ib; i < ie; ++i)
                    for (int64_t m = 0; m < M; ++m) {
                        const uint64_t c = load_code(codes, code_bytes, (r0 + i) * c_rs + m * c_cs);
                        if (c >= (uint64_t)cb->K) bad = true;  // primitives.rs:146 index_axis panic
                        if (dev_bytes == 1) ((uint8_t*)hin)[i * M + m] = (uint8_t)c;
                        else ((uint32_t*)hin)[i * M + m] = (uint32_t)std::min<uint64_t>(c, 0xffffffffull);
                    }
                if (bad) range_err = true;
            });
            if (range_err) break;
            HIPCHK(hipMemcpyAsync(ss.st[b].d_in, hin, (size_t)rows * M * dev_bytes,
                                  hipMemcpyHostToDevice, ss.stream[b]));
            PQCHK(reconstruct_dev_impl(cb, slot, ss.st[b].d_in, dev_bytes, rows, M,
                                       (float*)ss.st[b].d_out, d, ss.stream[b]));
            HIPCHK(hipMemcpyAsync(ss.st[b].h_out, ss.st[b].d_out, (size_t)rows * d * sizeof(float),
                                  hipMemcpyDeviceToHost, ss.stream[b]));
            pend_r0[b] = r0; pend_rows[b] = rows;
        }
        for (int k = 0; k < 2; ++k)
            if (pend_rows[k]) PQCHK(drain(k, pend_r0[k], pend_rows[k]));
        return range_err ? PQHIP_ECODE_RANGE : PQHIP_OK;
    });
}

int32_t pqhip_cluster_assignments_f32(pqhip_ctx* ctx, const float* centroids, int64_t n_centroids,
                                      int64_t dim, const float* x, int64_t n_rows, int64_t x_rs,
                                      int64_t x_cs, void* out, int32_t out_bytes)
{
    if (!ctx || !centroids) return PQHIP_EINVAL;
    pqhip_codebook* cb = nullptr;
    PQCHK(pqhip_codebook_create(ctx, centroids, 1, n_centroids, dim, nullptr, &cb));
    const int32_t rc = pqhip_quantize_batch_f32(cb, x, n_rows, x_rs, x_cs, out, out_bytes, 1, 1);
    pqhip_codebook_destroy(cb);
    return rc;
}

int32_t pqhip_kmeans_iterations_f32(pqhip_ctx* ctx, float* quantizers, int64_t M, int64_t K, int64_t dsub,
                                    const float* x, int64_t n, int64_t x_rs, int64_t x_cs,
                                    int32_t n_iterations, float* loss)
{
    if (!ctx || !quantizers || n < 0 || n_iterations < 0) return PQHIP_EINVAL;
    if (ctx->devs.empty()) return PQHIP_ENODEV;
    if (M <= 0 || K <= 0 || dsub <= 0) return PQHIP_ESHAPE;
    if (n > 0 && (!x || x_rs < 0 || x_cs < 0)) return PQHIP_EINVAL;
    // the instances stay resident on the first device of the context for all iterations
    const int slot = 0;
    const int64_t d = M * dsub;
    pqhip_matrix* mx = nullptr;
    PQCHK(pqhip_matrix_upload_f32(ctx, slot, x, n, d, x_rs, x_cs, &mx));
    struct MG { pqhip_matrix* p; ~MG() { pqhip_matrix_destroy(p); } } mg{mx};
    struct { const float* p; } dx{mx->d};
    DeviceSlot& ds = *ctx->devs[slot];
    return pqhip_kmeans_iterations_f32_dev(ctx, slot, quantizers, M, K, dsub, (const float*)dx.p, n, d,
                                           n_iterations, loss, (void*)ds.stream[0]);
}

// ---- resident instance matrices (training entry points iterate over the same rows many times) ----
int32_t pqhip_matrix_upload_f32(pqhip_ctx* ctx, int32_t slot, const float* x, int64_t n, int64_t d, int64_t x_rs,
                                int64_t x_cs, pqhip_matrix** out)
{
    if (!ctx || !out || n < 0 || d <= 0) return PQHIP_EINVAL;
    *out = nullptr;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    if (n > 0 && (!x || x_rs < 0 || x_cs < 0)) return PQHIP_EINVAL;
    DeviceSlot& ds = *ctx->devs[slot];
    std::unique_ptr<pqhip_matrix> m(new pqhip_matrix());
    m->ctx = ctx; m->slot = slot; m->rows = n; m->cols = d;
    StageLease lease(ds);
    StageSet& ss = *lease.s;
    SET_DEVICE(ds.ordinal);
    PQCHK(quiesce_staging(ss));
    HIPCHK(hipMalloc((void**)&m->d, (size_t)std::max<int64_t>(n, 1) * d * sizeof(float)));
    struct Free { float* p; ~Free() { if (p) (void)hipFree(p); } } guard{m->d};
    const int64_t cap = stage_rows(std::max<int64_t>(n, 1), d * (int64_t)sizeof(float));
    for (int b = 0; b < 2; ++b) PQCHK(ensure_staging(ss.st[b], (size_t)cap * d * sizeof(float), 16));
    int b = 0;
    for (int64_t r0 = 0; r0 < n; r0 += cap, b ^= 1) {
        const int64_t rows = std::min<int64_t>(cap, n - r0);
        HIPCHK(hipStreamSynchronize(ss.stream[b]));
        float* hin = (float*)ss.st[b].h_in;
        lease.pool->run(rows, [&, r0, hin](int64_t ib, int64_t ie) {
            if (x_cs == 1) {
                for (int64_t i = ib; i < ie; ++i)
                    std::memcpy(hin + i * d, x + (r0 + i) * x_rs, (size_t)d * sizeof(float));
            } else {
                for (int64_t i = ib; i < ie; ++i)
                    for (int64_t k = 0; k < d; ++k) hin[i * d + k] = x[(r0 + i) * x_rs + k * x_cs];
            }
        });
        HIPCHK(hipMemcpyAsync(m->d + r0 * d, hin, (size_t)rows * d * sizeof(float), hipMemcpyHostToDevice, ss.stream[b]));
    }
    HIPCHK(hipStreamSynchronize(ss.stream[0]));
    HIPCHK(hipStreamSynchronize(ss.stream[1]));
    guard.p = nullptr;
    *out = m.release();
    return PQHIP_OK;
}

}  // extern "C"
