// pqhip_codebook.hip -- codebook handles of libpqhip.so: upload and replication over the context's devices, the
// preparation kernels (norms, finite-norm flag, MFMA fragment images), the leased scratch buffers and the per-stream
// error-flag slots.
#include "pqhip_internal.h"

#include "kernels_mfma.hip.h"     // kBigNorm
#include "kernels_prep.hip.h"
#include "smallk_launch.h"        // smallk_has / smallk_kp
#include "vor2_prep.h"

using namespace pqhip;

namespace pqh {

// Lease a scratch buffer of at least `bytes` for one call on stream `st` (see ScratchBuf).  Preference:
// an idle buffer that is large enough; an idle buffer that has to grow (or a new one while the pool is
// below kScratchPoolMax); otherwise the call queues behind a buffer whose work is still in flight
// (stream order through its event) or, when every buffer is leased to another host thread, waits for a
// release.  On return the buffer is exclusively this call's until release_scratch(), and *out_p is its
// device pointer (copied under the mutex: the pool vector may be touched by other threads afterwards).
// Nothing that can block for long -- hipEventSynchronize on the old buffer's work, hipFree, a <= 4 GiB
// hipMalloc -- runs under cb->mu: the buffer is first marked leased (nobody else can pick it), then resized
// with the mutex released, so other callers of the codebook (and every release_scratch) keep moving.
// Nested leases (a call that holds a buffer and needs another: OPQ rotation scratch -> K > 256 keys; converted codes ->
// rotation scratch -> keys; lookup staging) take their buffers from SEPARATE partitions of the pool, one per nesting depth of
// the calling thread: a holder of a depth-0 buffer only ever waits for depth-1 buffers, whose holders wait for nothing of
// depth <= 1 -- so three callers that each hold one buffer and want a second cannot wait for each other in a circle (with one
// shared partition of three they could).
thread_local int g_lease_depth = 0;

int32_t lease_scratch(pqhip_codebook* cb, int slot, size_t bytes, hipStream_t st, int* out_idx, void** out_p)
{
    CodebookDev& cd = cb->dev[slot];
    const int level = std::min(g_lease_depth, kScratchLevels - 1);
    const int lo = level * kScratchPoolMax, hi = lo + kScratchPoolMax;
    std::unique_lock<std::mutex> lk(cb->mu);
    for (;;) {
        int idle_fit = -1, idle_any = -1, busy_fit = -1, busy_any = -1, fresh = -1;
        for (int i = lo; i < hi; ++i) {
            ScratchBuf& b = cd.pool[i];
            if (b.leased) continue;
            if (!b.done) { if (fresh < 0) fresh = i; continue; }      // slot never used: its event is created on first lease
            const bool idle = hipEventQuery(b.done) == hipSuccess;
            (void)hipGetLastError();
            const bool fit = b.bytes >= bytes;
            if (idle && fit && idle_fit < 0) idle_fit = i;
            if (idle && idle_any < 0) idle_any = i;
            if (!idle && fit && busy_fit < 0) busy_fit = i;
            if (!idle && busy_any < 0) busy_any = i;
        }
        int pick = idle_fit;
        if (pick < 0 && fresh >= 0) {
            HIPCHK(hipEventCreateWithFlags(&cd.pool[fresh].done, hipEventDisableTiming));   // (a never-recorded event counts as complete)
            pick = fresh;
        }
        if (pick < 0) pick = idle_any >= 0 ? idle_any : busy_fit >= 0 ? busy_fit : busy_any;
        if (pick < 0) {              // every buffer of this depth is leased to another host thread
            cb->cv.wait(lk);
            continue;
        }
        cd.pool[pick].leased = true;
        if (cd.pool[pick].bytes < bytes) {
            // leased to us, so no other host thread holds or can take this buffer: wait for the device work that
            // still uses the old allocation and replace it, with the mutex released
            hipEvent_t done = cd.pool[pick].done;
            void* old = cd.pool[pick].p;
            lk.unlock();
            hipError_t e = hipEventSynchronize(done);
            bool freed = false;
            if (e == hipSuccess && old) { e = hipFree(old); freed = e == hipSuccess; }
            void* np = nullptr;
            if (e == hipSuccess) e = hipMalloc(&np, bytes);
            lk.lock();
            ScratchBuf& b = cd.pool[pick];
            if (e != hipSuccess) {
                if (freed) { b.p = nullptr; b.bytes = 0; }   // (otherwise the old allocation stays on record)
                b.leased = false;
                lk.unlock();
                cb->cv.notify_all();
                g_hip_err = std::string("lease_scratch: ") + hipGetErrorString(e);
                (void)hipGetLastError();
                return (e == hipErrorOutOfMemory) ? PQHIP_ENOMEM : PQHIP_EHIP;
            }
            b.p = np;
            b.bytes = bytes;
        }
        ScratchBuf& b = cd.pool[pick];
        const hipError_t e = hipStreamWaitEvent(st, b.done, 0);
        if (e != hipSuccess) {
            b.leased = false;
            lk.unlock();
            cb->cv.notify_all();
            g_hip_err = std::string("hipStreamWaitEvent(scratch): ") + hipGetErrorString(e);
            (void)hipGetLastError();
            return PQHIP_EHIP;
        }
        *out_idx = pick;
        *out_p = b.p;
        ++g_lease_depth;
        return PQHIP_OK;
    }
}

void release_scratch(pqhip_codebook* cb, int slot, int idx, hipStream_t st)
{
    CodebookDev& cd = cb->dev[slot];
    {
        std::lock_guard<std::mutex> g(cb->mu);
        (void)hipEventRecord(cd.pool[idx].done, st);
        cd.pool[idx].leased = false;
    }
    --g_lease_depth;
    cb->cv.notify_all();             // (waiters of different depths share the condition variable)
}

// ---- error-flag slots --------------------------------------------------------------------------------------------
// One slot per caller stream; when all kErrSlots are taken the least recently used one is handed to the new stream:
// the new stream first WAITS for the event that the old stream recorded behind its last flag-raising launches, then
// clears the flag (so a late atomicOr of the old stream cannot land after the clear and be reported to the wrong
// caller -- ADVICE r3); a pending error of a stream that has not been seen for kErrSlots other streams is dropped
// rather than delivered to the wrong caller.  A destroyed and re-created stream with the same handle value keeps
// its slot -- callers that check after every call, the default of the Python / C++ / Rust mirrors, never leave one pending.
ErrFlag::ErrFlag(pqhip_codebook* c, int s, hipStream_t t) : cb(c), slot(s), idx(0), st(t), flag(nullptr)
{
    CodebookDev& cd = cb->dev[slot];
    std::lock_guard<std::mutex> g(cb->mu);
    const uint64_t now = ++cd.err_clock;
    int i = 0;
    for (; i < (int)cd.err_streams.size(); ++i)
        if (cd.err_streams[i] == st) break;
    if (i == (int)cd.err_streams.size()) {
        if (i < kErrSlots) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; (void)hipGetLastError(); }
            cd.err_streams.push_back(st);
            cd.err_used.push_back(now);
            cd.err_done.push_back(ev);
        } else {
            i = 0;
            for (int k = 1; k < kErrSlots; ++k)
                if (cd.err_used[k] < cd.err_used[i]) i = k;
            cd.err_streams[i] = st;
            if (cd.err_done[i]) (void)hipStreamWaitEvent(st, cd.err_done[i], 0);   // the evicted stream's launches first
            (void)hipMemsetAsync(cd.err + 2 + i, 0, sizeof(int), st);
            (void)hipGetLastError();
        }
    }
    cd.err_used[i] = now;
    idx = i;
    flag = cd.err + 2 + i;
}

ErrFlag::~ErrFlag()
{
    CodebookDev& cd = cb->dev[slot];
    std::lock_guard<std::mutex> g(cb->mu);
    // (the slot may have been recycled to another stream by a concurrent caller meanwhile: then it is theirs to mark)
    if (idx < (int)cd.err_streams.size() && cd.err_streams[idx] == st && cd.err_done[idx]) {
        (void)hipEventRecord(cd.err_done[idx], st);
        (void)hipGetLastError();
    }
}

// ---- preparation ---------------------------------------------------------------------------------------------------
static int32_t launch_prepare(pqhip_codebook* cb, int slot, hipStream_t st)
{
    CodebookDev& cd = cb->dev[slot];
    const int64_t M = cb->M, K = cb->K, dsub = cb->dsub;
    HIPCHK(hipMemsetAsync(cd.err + 1, 0, sizeof(int), st));
    const int total = (int)(M * cb->k_pad);
    hipLaunchKernelGGL(k_centroid_norms, dim3((total + 255) / 256), dim3(256), 0, st, cd.cb,
                       (int)M, (int)K, (int)dsub, cb->k_pad, cd.cc);
    hipLaunchKernelGGL(k_check_norms, dim3((total + 255) / 256), dim3(256), 0, st, cd.cc,
                       (int)M, (int)K, cb->k_pad, kBigNorm, cd.err + 1);
    note_kernel("k_centroid_norms");
    note_kernel("k_check_norms");
    if (cb->T) {
        const int S = cb->DP / 2;
        const int tiles = cb->T * cb->groups;  // grouped codebooks: [M][groups * 8][S][64]
        const int64_t tot = M * tiles * S * 64;
        hipLaunchKernelGGL(k_build_frags, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                           cd.cb, (int)M, (int)K, (int)dsub, tiles, S, cd.frags);
        note_kernel("k_build_frags");
    }
    if (cb->KP) {
        const int64_t tot = M * dsub * cb->KP;
        hipLaunchKernelGGL(k_build_cbt, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, cd.cb, (int)M, (int)K,
                           (int)dsub, cb->KP, cd.cbt);
        note_kernel("k_build_cbt");
    }
    if (cb->pair16) {
        const int NP = (int)((M + 1) / 2);
        const int64_t tot = (int64_t)NP * dsub * 64 + NP * 32;
        hipLaunchKernelGGL(k_build_pair_frags, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, cd.cb, cd.cc, (int)M, (int)K,
                           (int)dsub, cb->k_pad, cd.fragp, cd.fragp + (int64_t)NP * dsub * 64);
        note_kernel("k_build_pair_frags");
    }
    HIPCHK(hipGetLastError());
    return PQHIP_OK;
}

// (Re)derive everything the encode kernels need from the centroids in cd.cb: ||c||^2, the
// finite/small-norm flag and the MFMA A-fragment image.  Synchronises `st` (4-byte flag readback).
int32_t prepare_codebook_dev(pqhip_codebook* cb, int slot, hipStream_t st, bool* norms_ok)
{
    PQCHK(launch_prepare(cb, slot, st));
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, cb->dev[slot].err + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *norms_ok = bad == 0;
    return PQHIP_OK;
}

// the same launches without looking at the flag (it stays on the device for the kernels to read)
int32_t prepare_codebook_async(pqhip_codebook* cb, int slot, hipStream_t st) { return launch_prepare(cb, slot, st); }

// only_slot < 0: replicate on every device of the context (Pq handles); otherwise build the
// device copy on that slot alone (internal k-means handles).
int32_t codebook_create_impl(pqhip_ctx* ctx, const float* quantizers, int64_t M, int64_t K,
                             int64_t dsub, const float* projection, int only_slot,
                             pqhip_codebook** out)
{
    if (!ctx || !out) return PQHIP_EINVAL;
    *out = nullptr;
    if (!quantizers) return PQHIP_EINVAL;
    if (M <= 0 || K <= 0 || dsub <= 0) return PQHIP_ESHAPE;  // pq.rs:39-42 "without quantizers"
    if (M > 65535 || dsub > 65535 || K > (1ll << 31) - 1 || M * dsub > (1 << 24)) return PQHIP_EUNSUPPORTED;
    if (only_slot >= (int)ctx->devs.size()) return PQHIP_ENODEV;

    // on any failure below the partially built handle (and its device memory) is destroyed
    struct Guard { pqhip_codebook* p; ~Guard() { if (p) pqhip_codebook_destroy(p); } } guard{new pqhip_codebook()};
    pqhip_codebook* cb = guard.p;
    cb->ctx = ctx;
    cb->M = M; cb->K = K; cb->dsub = dsub; cb->d = M * dsub;
    cb->has_proj = projection != nullptr;
    // MFMA geometry: K <= 256 padded to {1,2,4,8} tiles of 32; dsub <= 32 padded to an even
    // number of k (one MFMA consumes two); A fragments must fit (T * DP/2 <= 128).
    int T = 0, DP = 0, groups = 1;
    if (K <= 65536 && dsub <= 128) {
        // sub-dimension: even padding up to 32, multiples of 8 for wide sub-vectors (33..64), of 16 for 65..128 (one chain
        // of up to 128 k: still a single rule-2 block; the scalar kernel those shapes ran on reached 5e5 vectors/s)
        DP = dsub <= 32 ? (int)round_up(dsub, 2) : dsub <= 64 ? (int)round_up(dsub, 8) : (int)round_up(dsub, 16);
        if (K <= 256) {
            const int tiles = (int)((K + 31) / 32);
            T = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : 8;
        } else {
            // grouped: ceil(K / 256) virtual subquantizers of 8 tiles each per real one
            T = 8;
            groups = (int)((K + 255) / 256);
        }
    }
    if (K <= 65536 && dsub > 128 && dsub <= 256) {
        // one chain of up to 256 k is still a single rule-2 block; groups of <= 128 centroids keep the fragments within LDS
        DP = (int)round_up(dsub, 16);
        const int tiles = (int)((std::min<int64_t>(K, 128) + 31) / 32);
        T = tiles <= 1 ? 1 : tiles <= 2 ? 2 : 4;
        groups = (int)((K + 32 * T - 1) / (32 * T));
        cb->wide = true;
    }
    if (K <= 65536 && dsub > 256 && dsub <= 1024) {
        // several rule-2 blocks per dot product (k_encode_mfma_wide2): the fragments of all blocks of a group stay in LDS,
        // T DP / 8 KB <= 128 KB -> groups of 64 centroids up to 512 floats, of 32 beyond
        DP = (int)round_up(dsub, 64);
        T = (DP <= 512 && K > 32) ? 2 : 1;
        groups = (int)((K + 32 * T - 1) / (32 * T));
        cb->wide = true;
    }
    cb->T = T; cb->DP = DP; cb->groups = groups;
    cb->KP = (T != 0 && smallk_has((int)dsub)) ? smallk_kp(K) : 0;
    {   // pair kernel: K <= 16, power-of-two sub-vectors up to 16 floats, fragment image + slabs within 160 KB of LDS
        const int64_t NP = (M + 1) / 2;
        const size_t lds = ((size_t)NP * dsub * 64 + (size_t)NP * 32 + 4 * 2 * 32 * 36) * sizeof(float);
        cb->pair16 = K <= 16 && (dsub == 2 || dsub == 4 || dsub == 8 || dsub == 16) && lds <= 160 * 1024;
    }
    cb->k_pad = T ? T * 32 * groups : (int)round_up(K, 32);
    const int S = DP / 2;

    std::vector<float> PT;
    if (projection) {
        PT.resize((size_t)cb->d * cb->d);
        for (int64_t k = 0; k < cb->d; ++k)
            for (int64_t c = 0; c < cb->d; ++c) PT[c * cb->d + k] = projection[k * cb->d + c];
    }

    // 2-float sub-vectors: candidate tables (Pq handles only: the centroids of a k-means handle move)
    Vor2Tables vor2;
    // (context option "candidate_tables" = 0 skips them: the host build takes 2.4 s for M = 150, K = 256 on 8 cores)
    if (dsub <= 2 && K <= 256 && only_slot < 0 && T != 0 && ctx->opt.candidate_tables.load(std::memory_order_relaxed) != 0 &&
        vor2_build(quantizers, M, K, dsub, vor2)) {
        cb->vor2 = true;
        cb->vor2_max_region_words = vor2.max_region_words;
    }

    cb->dev.resize(ctx->devs.size());
    for (CodebookDev& cd : cb->dev) cd.pool.resize((size_t)kScratchPoolMax * kScratchLevels);   // fixed size: elements never move
    bool norms_ok = true;
    for (size_t i = 0; i < ctx->devs.size(); ++i) {
        if (only_slot >= 0 && (int)i != only_slot) continue;
        CodebookDev& cd = cb->dev[i];
        SET_DEVICE(ctx->devs[i]->ordinal);
        hipStream_t st = ctx->devs[i]->stream[0];
        const size_t cb_bytes = (size_t)(M * K * dsub) * sizeof(float);
        HIPCHK(hipMalloc((void**)&cd.cb, cb_bytes));
        HIPCHK(hipMemcpyAsync(cd.cb, quantizers, cb_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMalloc((void**)&cd.cc, (size_t)M * cb->k_pad * sizeof(float)));
        HIPCHK(hipMalloc((void**)&cd.err, (2 + kErrSlots) * sizeof(int)));
        HIPCHK(hipMemsetAsync(cd.err, 0, (2 + kErrSlots) * sizeof(int), st));
        if (T) HIPCHK(hipMalloc((void**)&cd.frags, (size_t)(M * groups * T * S * 64) * sizeof(float)));
        if (cb->KP) HIPCHK(hipMalloc((void**)&cd.cbt, (size_t)(M * dsub * cb->KP) * sizeof(float)));
        if (cb->pair16) HIPCHK(hipMalloc((void**)&cd.fragp, (size_t)(((M + 1) / 2) * (dsub * 64 + 32)) * sizeof(float)));
        if (cb->vor2) {
            HIPCHK(hipMalloc((void**)&cd.vor2_tab, vor2.words.size() * sizeof(uint32_t)));
            HIPCHK(hipMalloc((void**)&cd.vor2_off, vor2.region_off.size() * sizeof(uint32_t)));
            HIPCHK(hipMemcpyAsync(cd.vor2_tab, vor2.words.data(), vor2.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(cd.vor2_off, vor2.region_off.data(), vor2.region_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
        if (projection) {
            const size_t pb = (size_t)cb->d * cb->d * sizeof(float);
            HIPCHK(hipMalloc((void**)&cd.P, pb));
            HIPCHK(hipMalloc((void**)&cd.PT, pb));
            HIPCHK(hipMemcpyAsync(cd.P, projection, pb, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(cd.PT, PT.data(), pb, hipMemcpyHostToDevice, st));
        }
        bool ok = true;
        PQCHK(prepare_codebook_dev(cb, (int)i, st, &ok));
        if (!ok) norms_ok = false;   // (prepare_codebook_dev returns synchronised: the device copy is ready for any stream)
    }
    cb->norms_ok = norms_ok;
    *out = cb;
    guard.p = nullptr;
    return PQHIP_OK;
}

}  // namespace pqh

using namespace pqh;

extern "C" {

int32_t pqhip_codebook_create(pqhip_ctx* ctx, const float* quantizers, int64_t M, int64_t K,
                              int64_t dsub, const float* projection, pqhip_codebook** out)
{
    return codebook_create_impl(ctx, quantizers, M, K, dsub, projection, -1, out);
}

void pqhip_codebook_destroy(pqhip_codebook* cb)
{
    if (!cb) return;
    for (size_t i = 0; i < cb->dev.size(); ++i) {
        CodebookDev& cd = cb->dev[i];
        if (!cd.cb && !cd.err) continue;   // never built on this device (single-slot training handles)
        DeviceGuard dg(cb->ctx->devs[i]->ordinal);
        (void)hipDeviceSynchronize();
        if (cd.cb) (void)hipFree(cd.cb);
        if (cd.frags) (void)hipFree(cd.frags);
        if (cd.cc) (void)hipFree(cd.cc);
        if (cd.cbt) (void)hipFree(cd.cbt);
        if (cd.fragp) (void)hipFree(cd.fragp);
        if (cd.vor2_tab) (void)hipFree(cd.vor2_tab);
        if (cd.vor2_off) (void)hipFree(cd.vor2_off);
        if (cd.P) (void)hipFree(cd.P);
        if (cd.PT) (void)hipFree(cd.PT);
        if (cd.err) (void)hipFree(cd.err);
        for (hipEvent_t e : cd.err_done)
            if (e) (void)hipEventDestroy(e);
        for (ScratchBuf& b : cd.pool) {
            if (b.p) (void)hipFree(b.p);
            if (b.done) (void)hipEventDestroy(b.done);
        }
    }
    delete cb;
}

int64_t pqhip_codebook_quantized_len(const pqhip_codebook* cb) { return cb ? cb->M : 0; }
int64_t pqhip_codebook_reconstructed_len(const pqhip_codebook* cb) { return cb ? cb->d : 0; }
int64_t pqhip_codebook_n_centroids(const pqhip_codebook* cb) { return cb ? cb->K : 0; }
int32_t pqhip_codebook_has_projection(const pqhip_codebook* cb) { return cb && cb->has_proj; }

// variants: 0 auto; 1 scalar anchor; 2 VALU-argmin MFMA kernel; 4 k_encode_mfma_lds3; 6 small-codebook VALU kernel;
// 7 pair kernel; 8 fused OPQ kernel; 9 k_encode_mfma16.  (3 and 5 were the retired register-resident LDS-argmin kernel
// and the first-generation fused OPQ kernel: refused since round 4.)
int32_t pqhip_set_encode_variant(pqhip_codebook* cb, int32_t variant)
{
    if (!cb || variant < 0 || variant > 11 || variant == 3 || variant == 5) return PQHIP_EINVAL;
    cb->variant = variant;
    return PQHIP_OK;
}

int32_t pqhip_vor2_tables_host(const float* quantizers, int64_t M, int64_t K, int64_t dsub, uint32_t* words_out, int64_t words_cap,
                               uint32_t* region_off_out, int64_t* n_words)
{
    if (!quantizers || !n_words || M <= 0 || K <= 0 || dsub < 1 || dsub > 2) return PQHIP_EINVAL;
    Vor2Tables t;
    if (!vor2_build(quantizers, M, K, dsub, t)) return PQHIP_EUNSUPPORTED;
    *n_words = (int64_t)t.words.size();
    if (words_out) {
        if (words_cap < (int64_t)t.words.size()) return PQHIP_EINVAL;
        std::copy(t.words.begin(), t.words.end(), words_out);
    }
    if (region_off_out) std::copy(t.region_off.begin(), t.region_off.end(), region_off_out);
    return PQHIP_OK;
}

const char* pqhip_last_encode_kernel(const pqhip_codebook* cb)
{
    return cb ? cb->last_kernel.load() : "";
}

int32_t pqhip_check_codes_dev(pqhip_codebook* cb, int32_t slot, void* stream)
{
    if (!cb) return PQHIP_EINVAL;
    if (slot < 0 || slot >= (int)cb->dev.size()) return PQHIP_ENODEV;
    SET_DEVICE(cb->ctx->devs[slot]->ordinal);
    hipStream_t st = (hipStream_t)stream;
    ErrFlag ef(cb, slot, st);   // the flag of THIS stream's calls: concurrent callers on other streams keep theirs
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, ef.flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(ef.flag, 0, sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    return flag ? PQHIP_ECODE_RANGE : PQHIP_OK;
}

}  // extern "C"
