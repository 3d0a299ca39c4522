// pqhip_train.hip -- the training rows of the path ("next" row): k-means iterations on all
// subquantizers (kmeans.rs:308-360), X^T.R (opq.rs:191), the device part of Opq::train_iteration (opq.rs:156-195)
// and the resident instance matrices they iterate over.
#include "pqhip_internal.h"

#include "kernels_kmeans.hip.h"
#include "kernels_atb.hip.h"

using namespace pqhip;

namespace pqh {

// `n_iterations` x kmeans_iteration (kmeans.rs:308-327) on every subquantizer of `cb`, whose
// device copy on `slot` is updated in place.  Work on one stream; returns synchronised.
static int32_t kmeans_run_dev(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs,
                       int n_iterations, float* h_loss, hipStream_t st)
{
    CodebookDev& cd = cb->dev[slot];
    const int64_t M = cb->M, K = cb->K, dsub = cb->dsub;
    if (n > (1ll << 31) || K > 16384) return PQHIP_EUNSUPPORTED;  // 32-bit row ids; K counters in LDS
    if (x_rs >= (1ll << 30)) return PQHIP_EUNSUPPORTED;           // 32-bit byte stride in the update walk
    if (n == 0) {
        // no instances: every centroid is "empty" -> zero (kmeans.rs:180), loss 0/0
        HIPCHK(hipMemsetAsync(cd.cb, 0, (size_t)(M * K * dsub) * sizeof(float), st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_loss) for (int64_t m = 0; m < M; ++m) h_loss[m] = std::numeric_limits<float>::quiet_NaN();
        return PQHIP_OK;
    }
    const int code_bytes = K <= 256 ? 1 : 4;
    // Row windows: window w is assigned on `st`, and partitioned + summed on a second stream while
    // the MFMA-bound assignment of window w+1 runs (the update walk is memory-latency bound, the
    // two overlap well).  The chains carry over from window to window, so the order of the adds
    // is still the row order.
    const int64_t win_opt = cb->ctx->opt.kmeans_window_rows.load(std::memory_order_relaxed);   // option "kmeans_window_rows" (tests shrink it to exercise many windows)
    const int64_t win_rows_target = win_opt > 0 ? win_opt : (int64_t)(512 << 10);
    const int64_t wrows = std::min<int64_t>(n, round_up(std::max<int64_t>(win_rows_target, (n + 31) / 32), 64));
    const int nwin = (int)((n + wrows - 1) / wrows);
    const int64_t rpb = std::max<int64_t>(4096, round_up((wrows + 255) / 256, 64));
    const int nb_max = (int)((wrows + rpb - 1) / rpb);
    const int64_t w_pad = round_up(wrows, 4);  // 16-byte aligned row-id groups for every subquantizer
    // work buffers: the device's grow-only training workspaces 3..9 (the caller holds ds.train_mu);
    // seven hipMalloc/hipFree per call used to cost more than a small training set's iterations
    DeviceSlot& ds = *cb->ctx->devs[slot];
    struct { void* p = nullptr; } codes, counts, seg, perm, loss, acc, tot;
    {
        const size_t need[7] = {(size_t)n * M * code_bytes, (size_t)M * nb_max * K * sizeof(unsigned),
                                (size_t)M * (K + 1) * sizeof(unsigned), (size_t)M * w_pad * sizeof(unsigned),
                                (size_t)M * sizeof(float), (size_t)(M * K * dsub) * sizeof(float),
                                (size_t)2 * M * K * sizeof(unsigned)};
        void** dst[7] = {&codes.p, &counts.p, &seg.p, &perm.p, &loss.p, &acc.p, &tot.p};
        for (int i = 0; i < 7; ++i) {
            PQCHK(ensure_ws(ds, 3 + i, std::max<size_t>(need[i], 16)));
            *dst[i] = ds.ws[3 + i];
        }
    }
    const size_t lds_k = (size_t)K * sizeof(unsigned), lds_scan = (size_t)(K + 256) * sizeof(unsigned);
    HIPCHK(hipFuncSetAttribute((const void*)k_km_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const float len_f = (float)(uint64_t)(n * dsub);  // `instances.len().as_()` (kmeans.rs:359)
    const int64_t lanes = M * K * dsub;

    const float* gx = d_x;  // what the update walk reads: the row-major instances themselves
    const int64_t g_rs = x_rs, g_ms = dsub;
    const bool vec4 = (dsub % 4 == 0) && (g_rs % 4 == 0) && ((reinterpret_cast<uintptr_t>(gx) & 15) == 0);

    struct Aux {
        hipStream_t s = nullptr;
        std::vector<hipEvent_t> ev;
        hipEvent_t done = nullptr;
        ~Aux()
        {
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            if (done) (void)hipEventDestroy(done);
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        }
    } aux;
    {
        // highest priority: its short, latency-bound kernels must not queue behind the long
        // assignment workgroups of the other stream
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&aux.s, hipStreamNonBlocking, greatest));
    }
    aux.ev.assign((size_t)nwin, nullptr);
    for (int w = 0; w < nwin; ++w) HIPCHK(hipEventCreateWithFlags(&aux.ev[w], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&aux.done, hipEventDisableTiming));
    hipStream_t su = aux.s;

    // One kmeans_iteration.  s_enc / s_upd: streams of the assignment and of the update (equal inside a
    // captured graph); bad_flag: device flag consulted by the encode kernel instead of the host;
    // with_loss: launch the exact loss fold; host_prep: rebuild the encode tables with the host reading
    // the finite-norm flag (the per-iteration sync) rather than leaving it on the device.
    auto iteration = [&](hipStream_t s_enc, hipStream_t s_upd, const int* bad_flag, bool with_loss, bool host_prep) -> int32_t {
        for (int w = 0; w < nwin; ++w) {
            const int64_t r0 = (int64_t)w * wrows, rows = std::min<int64_t>(wrows, n - r0);
            const float* xw = d_x + r0 * x_rs;
            char* cw = (char*)codes.p + r0 * M * code_bytes;
            PQCHK(encode_plain_dev(cb, slot, xw, rows, x_rs, cw, code_bytes, M, s_enc, bad_flag, /*beside_update=*/true));
            if (s_enc != s_upd) {
                HIPCHK(hipEventRecord(aux.ev[w], s_enc));
                HIPCHK(hipStreamWaitEvent(s_upd, aux.ev[w], 0));
            }
            const int nb = (int)((rows + rpb - 1) / rpb);
            const dim3 gbm((unsigned)nb, (unsigned)M);
#define KM_LAUNCH(IDX)                                                                                   \
            hipLaunchKernelGGL((k_km_hist<IDX>), gbm, dim3(256), lds_k, s_upd, (const IDX*)cw, rows, M, (int)K, \
                               (int)rpb, nb, (unsigned*)counts.p);                                       \
            hipLaunchKernelGGL(k_km_scan, dim3((unsigned)M), dim3(256), lds_scan, s_upd, (unsigned*)counts.p, \
                               (int)K, nb, (unsigned*)seg.p);                                            \
            hipLaunchKernelGGL((k_km_scatter<IDX>), gbm, dim3(64), lds_k, s_upd, (const IDX*)cw, rows, M, \
                               (int)K, (int)rpb, nb, (const unsigned*)counts.p, (const unsigned*)seg.p,  \
                               (unsigned*)perm.p, w_pad)
            if (code_bytes == 1) { KM_LAUNCH(uint8_t); } else { KM_LAUNCH(uint32_t); }
#undef KM_LAUNCH
            note_kernel("k_km_hist");
            note_kernel("k_km_scan");
            note_kernel("k_km_scatter");
            const unsigned* tin = (const unsigned*)tot.p + (size_t)(w & 1) * M * K;
            unsigned* tout = (unsigned*)tot.p + (size_t)((w + 1) & 1) * M * K;
            const int first = w == 0, last = w == nwin - 1;
            const float* gw = gx + r0 * g_rs;
            // one wave per cluster (rows of a sub-vector on q lanes); lane-per-chain form for very wide sub-vectors
            const bool wave_form = cb->ctx->opt.kmeans_lane_form.load(std::memory_order_relaxed) == 0 && dsub <= 64;  // one lane per dimension adds
            if (wave_form) {
                const int qq = vec4 ? (int)dsub / 4 : (int)dsub;
                const size_t slab = (size_t)4 * 8 * (64 / qq) * dsub * sizeof(float);
                const dim3 g((unsigned)((M * K + 3) / 4));
                if (vec4)
                    hipLaunchKernelGGL((k_km_segsum_w<true>), g, dim3(256), slab, s_upd, gw, g_rs, g_ms, w_pad,
                                       (const unsigned*)perm.p, (const unsigned*)seg.p, (int)M, (int)K, (int)dsub,
                                       (float*)acc.p, tin, tout, first, last);
                else
                    hipLaunchKernelGGL((k_km_segsum_w<false>), g, dim3(256), slab, s_upd, gw, g_rs, g_ms, w_pad,
                                       (const unsigned*)perm.p, (const unsigned*)seg.p, (int)M, (int)K, (int)dsub,
                                       (float*)acc.p, tin, tout, first, last);
                note_kernel("k_km_segsum_w");
            } else {
                hipLaunchKernelGGL(k_km_segsum, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s_upd, gw, g_rs, g_ms, w_pad,
                                   (const unsigned*)perm.p, (const unsigned*)seg.p, (int)M, (int)K, (int)dsub,
                                   (float*)acc.p, tin, tout, first, last);
                note_kernel("k_km_segsum");
            }
            HIPCHK(hipGetLastError());
        }
        // the new centroids replace the old ones only after every window has been assigned
        if (s_enc != s_upd) {
            HIPCHK(hipEventRecord(aux.done, s_upd));
            HIPCHK(hipStreamWaitEvent(s_enc, aux.done, 0));
        }
        HIPCHK(hipMemcpyAsync(cd.cb, acc.p, (size_t)lanes * sizeof(float), hipMemcpyDeviceToDevice, s_enc));
        if (with_loss) {
            if (code_bytes == 1)
                hipLaunchKernelGGL((k_km_loss<uint8_t>), dim3((unsigned)M), dim3(256), 0, s_enc, d_x, x_rs, n,
                                   (const uint8_t*)codes.p, M, cd.cb, (int)K, (int)dsub, len_f, (float*)loss.p);
            else
                hipLaunchKernelGGL((k_km_loss<uint32_t>), dim3((unsigned)M), dim3(256), 0, s_enc, d_x, x_rs, n,
                                   (const uint32_t*)codes.p, M, cd.cb, (int)K, (int)dsub, len_f, (float*)loss.p);
            note_kernel("k_km_loss");
        }
        HIPCHK(hipGetLastError());
        if (host_prep) {
            bool ok = true;
            PQCHK(prepare_codebook_dev(cb, slot, s_enc, &ok));  // also the per-iteration synchronisation point
            cb->norms_ok = ok;
        } else {
            PQCHK(prepare_codebook_async(cb, slot, s_enc));
        }
        return PQHIP_OK;
    };

    // Small training sets are launch-bound (a dozen short kernels and a host sync per iteration):
    // all iterations but the last are one captured hipGraph replayed on the internal stream.  The
    // finite-norm decision stays on the device inside the graph (bad_flag).  Any failure to capture
    // or instantiate falls back to the eager loop below, which has not run anything yet.
    int it0 = 0;
    const bool try_graph = nwin == 1 && cb->groups == 1 && !cb->wide && cb->T != 0 && cb->variant != 1 && n_iterations >= 3 &&
                           n <= (1 << 20) && cb->ctx->opt.kmeans_no_graph.load(std::memory_order_relaxed) == 0;
    if (try_graph) {
        HIPCHK(hipEventRecord(aux.done, st));            // the instances and the codebook are ready on st
        HIPCHK(hipStreamWaitEvent(su, aux.done, 0));
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool ok = hipStreamBeginCapture(su, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int32_t rc = iteration(su, su, cd.err + 1, false, false);
            const hipError_t e = hipStreamEndCapture(su, &graph);
            ok = rc == PQHIP_OK && e == hipSuccess && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        int32_t status = PQHIP_OK;
        if (ok) {
            const int ng = n_iterations - 1;
            for (int i = 0; i < ng; ++i)
                if (hipGraphLaunch(exec, su) != hipSuccess) { status = PQHIP_EHIP; g_hip_err = "hipGraphLaunch (k-means iteration)"; break; }
            if (status == PQHIP_OK) {
                int bad = 0;
                if (hipMemcpyAsync(&bad, cd.err + 1, sizeof(int), hipMemcpyDeviceToHost, su) != hipSuccess ||
                    hipStreamSynchronize(su) != hipSuccess) { status = PQHIP_EHIP; g_hip_err = "k-means graph: flag readback"; }
                cb->norms_ok = bad == 0;
                it0 = ng;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (status != PQHIP_OK) return status;
        // su is synchronised (or untouched): st continues in order
    }
    for (int it = it0; it < n_iterations; ++it)
        PQCHK(iteration(st, su, nullptr, h_loss && it == n_iterations - 1, true));
    if (h_loss && n_iterations > 0) {
        HIPCHK(hipMemcpyAsync(h_loss, loss.p, (size_t)M * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return PQHIP_OK;
}

// C[da][db] (device, row stride pb = round_up(db, 16) floats, pa = round_up(da, 16) rows) = A^T . B over n rows with
// rule-2 arithmetic (kernels_atb.hip.h: one workgroup per row block computes all of its output, k_atb_fold adds the
// partial matrices in block order).  ga != nullptr: the rows of B are gathered from the codebook inside the kernel.
// Parts are processed in groups whose partial matrices fit 4 GiB of workspace; the fold carries C from group to group,
// so the order of the adds is the row order.  Context option "cross_product_exact" = 0: a part is 1/512 of the rows
// instead of one 256-row block (float-tolerance mode; see the kernel's header).
struct AtbGather { const void* codes; int64_t c_rs; int code_bytes; const float* cb; int K, dsub; };
static int32_t atb_dev(pqhip_ctx* ctx, DeviceSlot& ds, const float* dA, int64_t a_rs, int da, const float* dB, int64_t b_rs, int db,
                       const AtbGather* ga, int64_t n, float* dC, int pa, int pb, hipStream_t st)
{
    if (n == 0) {
        HIPCHK(hipMemsetAsync(dC, 0, (size_t)pa * pb * sizeof(float), st));
        return PQHIP_OK;
    }
    const bool exact = ctx->opt.cross_product_exact.load(std::memory_order_relaxed) != 0;
    const int64_t total_blocks = (n + kKC - 1) / kKC;
    // float-tolerance mode: ~2 parts per CU, whole 256-row blocks each
    const int64_t blocks_per_part = exact ? 1 : std::max<int64_t>(1, (total_blocks + 2 * ds.n_cus - 1) / (2 * ds.n_cus));
    const int64_t total_parts = (total_blocks + blocks_per_part - 1) / blocks_per_part;
    const int64_t per = (int64_t)pa * pb * sizeof(float);
    const int64_t group_opt = ctx->opt.cross_product_group_bytes.load(std::memory_order_relaxed);
    int64_t G = std::max<int64_t>(1, (group_opt > 0 ? group_opt : (4ll << 30)) / per);
    G = std::min<int64_t>(G, total_parts);
    PQCHK(ensure_ws(ds, 2, (size_t)G * per));
    float* part = (float*)ds.ws[2];
    AtbArgs a;
    a.A = dA; a.a_rs = a_rs; a.da = da; a.B = dB; a.b_rs = b_rs; a.db = db;
    a.codes = nullptr; a.c_rs = 0; a.cb = nullptr; a.K = 0; a.dsub = 1; a.inv_dsub = 0;
    const bool gather = ga != nullptr;
    if (gather) {
        a.codes = ga->codes; a.c_rs = ga->c_rs; a.cb = ga->cb; a.K = ga->K; a.dsub = ga->dsub;
        a.inv_dsub = (unsigned)(((1ull << 32) + ga->dsub - 1) / ga->dsub);
    }
    a.n = n; a.rows_per_part = blocks_per_part * kKC;
    a.nba = (da + kAtbW - 1) / kAtbW; a.nbb = (db + kAtbW - 1) / kAtbW;
    a.pa = pa; a.pb = pb; a.part = part;
    if (pa * (int64_t)pb > (1ll << 31) || n >= (1ll << 40)) return PQHIP_EUNSUPPORTED;
    // the padded rows / columns of C are never written by the kernel (only real 16 x 16 tiles are): keep them defined
    for (int64_t g0 = 0; g0 < total_parts; g0 += G) {
        const int np = (int)std::min<int64_t>(G, total_parts - g0);
        a.row0 = g0 * a.rows_per_part; a.nparts = np;
        const dim3 grid((unsigned)((int64_t)np * a.nba * a.nbb));
        if (!gather) hipLaunchKernelGGL((k_atb_rowblock<false, uint8_t>), grid, dim3(512), 0, st, a);
        else if (ga->code_bytes == 1) hipLaunchKernelGGL((k_atb_rowblock<true, uint8_t>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_atb_rowblock<true, uint32_t>), grid, dim3(512), 0, st, a);
        hipLaunchKernelGGL(k_atb_fold, dim3((unsigned)(((int64_t)pa * pb + 255) / 256)), dim3(256), 0, st,
                           (const float*)part, np, (int64_t)pa * pb, g0 == 0 ? 1 : 0, dC);
        HIPCHK(hipGetLastError());
        note_kernel(gather ? "k_atb_rowblock<gather>" : "k_atb_rowblock");
        note_kernel("k_atb_fold");
    }
    return PQHIP_OK;
}

}  // namespace pqh

using namespace pqh;

extern "C" {

// ---- "next" row: the k-means step of training ---------------------------------------------------
int32_t pqhip_kmeans_iterations_f32_dev(pqhip_ctx* ctx, int32_t slot, float* quantizers, int64_t M,
                                        int64_t K, int64_t dsub, const float* d_x, int64_t n,
                                        int64_t x_rs, int32_t n_iterations, float* loss, void* stream)
{
    if (!ctx || !quantizers || n < 0 || n_iterations < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_x || x_rs < M * dsub)) return PQHIP_EINVAL;
    SET_DEVICE(ctx->devs[slot]->ordinal);
    pqhip_codebook* cb = nullptr;
    PQCHK(codebook_create_impl(ctx, quantizers, M, K, dsub, nullptr, slot, &cb));
    struct G { pqhip_codebook* p; ~G() { pqhip_codebook_destroy(p); } } g{cb};
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> tg(ctx->devs[slot]->train_mu);   // the device's training workspaces
    if (n_iterations > 0) PQCHK(kmeans_run_dev(cb, slot, d_x, n, x_rs, n_iterations, loss, st));
    HIPCHK(hipMemcpy(quantizers, cb->dev[slot].cb, (size_t)(M * K * dsub) * sizeof(float), hipMemcpyDeviceToHost));
    return PQHIP_OK;
}

// ---- resident instance matrices (uploaded by pqhip_matrix_upload_f32, pqhip_host.hip) ----
const float* pqhip_matrix_device_ptr(const pqhip_matrix* m) { return m ? m->d : nullptr; }
int64_t pqhip_matrix_rows(const pqhip_matrix* m) { return m ? m->rows : 0; }

void pqhip_matrix_destroy(pqhip_matrix* m)
{
    if (!m) return;
    DeviceGuard dg(m->ctx->devs[m->slot]->ordinal);
    if (m->d) (void)hipFree(m->d);
    delete m;
}

// ---- "next" row: the device part of Opq::train_iteration (opq.rs:156-195) ------------------------
int32_t pqhip_opq_train_step_f32_dev(pqhip_ctx* ctx, int32_t slot, float* quantizers, int64_t M, int64_t K,
                                     int64_t dsub, const float* projection, const float* d_x, int64_t n,
                                     int64_t x_rs, float* cross, void* stream)
{
    if (!ctx || !quantizers || !projection || !cross || n < 0) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    if (M <= 0 || K <= 0 || dsub <= 0) return PQHIP_ESHAPE;
    const int64_t d = M * dsub;
    if (n > 0 && (!d_x || x_rs < d)) return PQHIP_EINVAL;
    SET_DEVICE(ctx->devs[slot]->ordinal);
    pqhip_codebook* cb = nullptr;
    PQCHK(codebook_create_impl(ctx, quantizers, M, K, dsub, projection, slot, &cb));
    struct G { pqhip_codebook* p; ~G() { pqhip_codebook_destroy(p); } } g{cb};
    CodebookDev& cd = cb->dev[slot];
    hipStream_t st = (hipStream_t)stream;
    const int code_bytes = K <= 256 ? 1 : 4;
    const int pa = (int)round_up(d, 16);
    DeviceSlot& ds = *ctx->devs[slot];
    std::lock_guard<std::mutex> tg(ds.train_mu);
    PQCHK(ensure_ws(ds, 0, (size_t)std::max<int64_t>(n, 1) * d * sizeof(float)));
    PQCHK(ensure_ws(ds, 1, (size_t)std::max<int64_t>(n, 1) * M * code_bytes));
    struct { void* p; } rx{ds.ws[0]}, codes{ds.ws[1]};
    DevBuf dcross;
    PQCHK(dcross.alloc((size_t)pa * pa * sizeof(float)));
    // opq.rs:167  rx = instances.dot(&projection)
    PQCHK(rotate_dev(d_x, n, x_rs, cd.P, (int)d, (float*)rx.p, d, st));
    // opq.rs:168  update_subquantizers: one kmeans_iteration per subquantizer on rx, loss discarded
    PQCHK(kmeans_run_dev(cb, slot, (const float*)rx.p, n, d, 1, nullptr, st));
    // opq.rs:176-182  quantize -> reconstruct round trip with the new centroids; opq.rs:191 instances.t().dot(&reconstructed).
    // The reconstructed matrix is never written: the cross-product kernel gathers its rows from the codebook (sub-vectors
    // of whole 16-byte pieces; other shapes reconstruct into rx first, which is recycled as in the reference).
    PQCHK(encode_plain_dev(cb, slot, (const float*)rx.p, n, d, codes.p, code_bytes, M, st));
    HIPCHK(hipMemsetAsync(dcross.p, 0, (size_t)pa * pa * sizeof(float), st));
    if (dsub % 4 == 0) {
        const AtbGather ga{codes.p, M, code_bytes, cd.cb, (int)K, (int)dsub};
        PQCHK(atb_dev(ctx, ds, d_x, x_rs, (int)d, nullptr, 0, (int)d, &ga, n, (float*)dcross.p, pa, pa, st));
    } else {
        {
            ErrFlag ef(cb, slot, st);
            PQCHK(gather_dev(cb, slot, codes.p, code_bytes, n, M, (float*)rx.p, d, st, ef.flag));
        }
        PQCHK(atb_dev(ctx, ds, d_x, x_rs, (int)d, (const float*)rx.p, d, (int)d, nullptr, n, (float*)dcross.p, pa, pa, st));
    }
    HIPCHK(hipMemcpy2DAsync(cross, (size_t)d * sizeof(float), dcross.p, (size_t)pa * sizeof(float),
                            (size_t)d * sizeof(float), (size_t)d, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(quantizers, cd.cb, (size_t)(M * K * dsub) * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PQHIP_OK;
}

int32_t pqhip_at_dot_b_f32_dev(pqhip_ctx* ctx, int32_t slot, const float* d_a, int64_t a_rs, int64_t da,
                               const float* d_b, int64_t b_rs, int64_t db, int64_t n, float* out, void* stream)
{
    if (!ctx || !out || n < 0 || da <= 0 || db <= 0 || da > 65536 || db > 65536) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;
    if (n > 0 && (!d_a || !d_b || a_rs < da || b_rs < db)) return PQHIP_EINVAL;
    SET_DEVICE(ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    const int pa = (int)round_up(da, 16), pb = (int)round_up(db, 16);
    DeviceSlot& ds = *ctx->devs[slot];
    std::lock_guard<std::mutex> tg(ds.train_mu);
    DevBuf dc;
    PQCHK(dc.alloc((size_t)pa * pb * sizeof(float)));
    HIPCHK(hipMemsetAsync(dc.p, 0, (size_t)pa * pb * sizeof(float), st));
    PQCHK(atb_dev(ctx, ds, d_a, a_rs, (int)da, d_b, b_rs, (int)db, nullptr, n, (float*)dc.p, pa, pb, st));
    HIPCHK(hipMemcpy2DAsync(out, (size_t)db * sizeof(float), dc.p, (size_t)pb * sizeof(float),
                            (size_t)db * sizeof(float), (size_t)da, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PQHIP_OK;
}

}  // extern "C"
