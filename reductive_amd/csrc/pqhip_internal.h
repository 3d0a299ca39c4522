// pqhip_internal.h -- what the host-side translation units of libpqhip.so share: handle layouts, error
// macros, the leases (staging sets, scratch buffers, error-flag slots), per-context options, the launch log
// and the prototypes of the device-side building blocks.  Round 4 split the former 2,471-line pqhip.hip by
// concern:
//     pqhip_ctx.hip       contexts, device slots, staging sets, options, launch log, status strings
//     pqhip_codebook.hip  codebook handles: upload, preparation kernels, scratch leases, error-flag slots
//     pqhip_encode.hip    PQ encode dispatch (every encode kernel family)
//     pqhip_rotate.hip    x.P / r.P^T dispatch (rotation kernels v8 / v9, slab fallback)
//     pqhip_opq.hip       quantize / reconstruct / lookup on device-resident rows (fused OPQ, gather)
//     pqhip_adc.hip       asymmetric-distance tables and scans
//     pqhip_train.hip     k-means iterations, X^T.R, the OPQ training step, resident matrices
//     pqhip_host.hip      host-resident entry points: row sharding over devices, pinned double-buffered staging
// There is deliberately NO CPU compute fallback anywhere: without HIP or a gfx950 device every compute entry
// point returns an error status.
#pragma once
#include "../../include/pqhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pqh {

extern thread_local std::string g_hip_err;

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            ::pqh::g_hip_err = std::string(#call) + ": " + hipGetErrorString(e__);           \
            (void)hipGetLastError();                                                         \
            return (e__ == hipErrorOutOfMemory) ? PQHIP_ENOMEM : PQHIP_EHIP;                  \
        }                                                                                    \
    } while (0)

#define PQCHK(call)                      \
    do {                                 \
        int32_t s__ = (call);            \
        if (s__ != PQHIP_OK) return s__; \
    } while (0)

constexpr int64_t kStageBytes = 256ll << 20;   // input bytes per pinned staging buffer of a host-resident call (two per staging set)
constexpr int64_t kStageRowsMin = 4096;
constexpr int64_t kScratchBytesMax = 4ll << 30;   // upper bound of one leased scratch buffer; the OPQ paths take far less
                                                  // (opq_chunk_rows: ~1.2 M rows, whole rounds of the rotation grid)
constexpr int kRotRowsPerWg = 12 * 32 * 12;       // P-block rotation kernels: 12 waves x 12 tiles of 32 rows per workgroup
constexpr int kScratchPoolMax = 3;                // leased scratch buffers per (codebook, device) and nesting depth: <= 12 GiB of the
                                                  // 288 GB HBM per depth, and only while that many callers are inside such calls at once
constexpr int kScratchLevels = 3;                 // nesting depths of leases (converted codes -> rotation scratch -> K > 256 keys)
constexpr int kErrSlots = 64;                     // per-stream "code >= K" flags per (codebook, device)
constexpr int kTrainWs = 10;                      // grow-only training workspaces per device
constexpr int kStageSets = 4;                     // staging sets per device slot (concurrent host callers on one device)

// ---- knobs ---------------------------------------------------------------------------------------------------
// (1) Environment variables of the DEFAULT build -- three, all about the host-resident path:
//       PQHIP_PACK_THREADS=<n>    host threads per device that pack / drain rows (default 16, capped by the cores)
//       PQHIP_HOST_ZERO_COPY=1    page-lock the caller's rows in place instead of packing them (pqhip_host.hip)
//       PQHIP_FUSED2_OPQ=0        OPQ encode through rotation + scratch + encode instead of the fused kernel
// (2) Per-context options, set through the C ABI (pqhip_ctx_set_option): what the test-suite and the A/B tools switch.
// (3) Everything else that rounds 1-3 read from PQHIP_DEBUG_* variables lives in `Diag`, which is a constant in the
//     default build (the compiler folds every use) and only reads the environment when the library is compiled with
//     -DPQHIP_DIAG (make DIAG=1 -> libpqhip_diag.so; make TIMING=1 implies it).  The shipped library's dispatch
//     depends on its arguments, the three variables above and the context's options -- nothing else.
struct Options {
    std::atomic<int64_t> kmeans_window_rows{0};     // rows per k-means window (0: 512 K)
    std::atomic<int64_t> kmeans_lane_form{0};       // 1: lane-per-chain update walk for every shape
    std::atomic<int64_t> kmeans_no_graph{0};        // 1: never replay small training sets as a captured hipGraph
    std::atomic<int64_t> opq_scratch_rows{0};       // rows per chunk of the two-kernel OPQ paths (0: whole rounds of the rotation grid)
    std::atomic<int64_t> opq_fused{1};              // 0: OPQ encode as rotation -> scratch -> encode
    std::atomic<int64_t> opq_gather_rotation{1};    // 0: OPQ reconstruct as gather -> scratch -> rotation
    std::atomic<int64_t> adc_single_query{0};       // 1: one scan pass per query
    std::atomic<int64_t> cross_product_exact{1};    // 0: X^T.R as a plain split-K product (float tolerance, no per-block partials)
    std::atomic<int64_t> lookup_two_pass{2};        // row lookups: 0 one pass, 1 select-then-reconstruct, 2 two passes when the matrix exceeds 256 MB
    std::atomic<int64_t> cross_product_group_bytes{0};   // workspace of partial matrices per launch group (0: 4 GiB; tests shrink it)
    std::atomic<int64_t> candidate_tables{1};       // 0: Pq handles are created without the candidate tables of the 1- / 2-float encode kernel (vor2_prep.h)
};

struct Diag {
    bool enc_stamp = false, rot_stamp = false, fused_stamp = false, occ = false, rec_elemwise = false, adc_any = false,
         no_mfma16 = false;
    const char* rot_stamp_file = nullptr;
    int64_t rpi_min = 32, rpi_max = 1024;
    int lds_pad = 0, rec_wgs = 0, adc_wgs = 0, rot_rpw = kRotRowsPerWg, fused2_tiles = 0;
};
#ifdef PQHIP_DIAG
const Diag& diag();                                // reads PQHIP_DEBUG_* once (pqhip_ctx.hip)
#else
inline constexpr Diag kNoDiag{};
inline const Diag& diag() { return kNoDiag; }
#endif
inline int rot_rows_per_wg() { return diag().rot_rpw; }

// ---- launch log ------------------------------------------------------------------------------------------------
// Every kernel launch of the library notes its name in a thread-local list (distinct names, with counts);
// pqhip_launch_log() renders it, pqhip_launch_log_reset() clears it.  bench.py takes `roofline.kernel` from it, so
// the line cannot name a kernel that did not run.  A pointer compare per launch.
void note_kernel(const char* name);

// RAII: every entry point runs on the device it was asked for and leaves the caller's thread on the
// device it came with (torch callers in the same process keep their current device).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int ordinal)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        err = hipSetDevice(ordinal);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define SET_DEVICE(ordinal)                  \
    ::pqh::DeviceGuard dev_guard__(ordinal); \
    HIPCHK(dev_guard__.err)

struct Staging {
    void* h_in = nullptr;   // pinned
    void* h_out = nullptr;  // pinned
    void* d_in = nullptr;
    void* d_out = nullptr;
    size_t in_bytes = 0, out_bytes = 0;
};

// A few persistent host threads per DEVICE: packing strided caller rows into the pinned staging buffers and
// draining results back is memory-bound work that one core cannot do at PCIe Gen5 speed (~10 GB/s per core
// against ~55 GB/s).  run() splits a row range into contiguous parts, executes one on the calling thread and
// returns when all are done.  One pool serves all staging sets of its device (round 3 created one pool of
// n_pack_threads per SET, under the slot mutex: up to 64 threads per GPU -- ADVICE r3); run() may be called
// from several host threads at once: parts go through one FIFO, every call waits on its own counter.
class RowPool {
public:
    explicit RowPool(int n_threads);
    ~RowPool();
    template <typename F>
    void run(int64_t rows, F fn)
    {
        const int nt = (int)std::min<int64_t>(n_, (rows + 1023) / 1024);
        if (nt <= 1) { fn((int64_t)0, rows); return; }
        const std::function<void(int64_t, int64_t)> f = fn;
        Call call;
        call.fn = &f;
        call.pending = nt - 1;
        const int64_t per = (rows + nt - 1) / nt;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (int i = 1; i < nt; ++i) {
                const int64_t b = std::min<int64_t>(rows, i * per), e = std::min<int64_t>(rows, b + per);
                q_.push_back(Task{&call, b, e});
            }
        }
        cv_.notify_all();
        f((int64_t)0, std::min<int64_t>(rows, per));
        std::unique_lock<std::mutex> lk(mu_);
        call.done.wait(lk, [&] { return call.pending == 0; });
    }

private:
    struct Call {
        const std::function<void(int64_t, int64_t)>* fn = nullptr;
        int pending = 0;                       // under mu_
        std::condition_variable done;
    };
    struct Task { Call* call; int64_t b, e; };
    void worker();
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    bool stop_ = false;
};

// Staging of one host-resident call: two streams and two pinned/device buffer pairs (double buffering).  A device
// slot owns kStageSets of them and LEASES one per call, so host callers on one device overlap one's packing and
// PCIe copies with the others' kernels (four sets: the cache-concurrency test's four host threads ran at 0.56-0.68
// of the serial time with two, 0.42-0.45 with four); the buffers of a set are created with its first lease.
struct StageSet {
    hipStream_t stream[2] = {nullptr, nullptr};
    Staging st[2];
    bool leased = false;
};

struct DeviceSlot {
    int ordinal = -1;
    int n_cus = 256;                 // compute units of the device (grid sizing of the persistent kernels)
    int n_pack_threads = 1;
    std::mutex mu;                   // guards sets[*].leased only (never held across a copy, a launch or a thread start)
    std::condition_variable cv;      // a staging set was released
    StageSet sets[kStageSets];
    std::once_flag pool_once;        // the packing threads start with the device's first host-resident call
    std::unique_ptr<RowPool> pool;
    hipStream_t stream[2] = {nullptr, nullptr};   // internal work: codebook preparation, training entry points
    // grow-only device workspaces of the training entry points (a 12 GB hipMalloc + hipFree per
    // call costs ~0.4 s); used under `train_mu`, released with the context
    std::mutex train_mu;
    void* ws[kTrainWs] = {};
    size_t ws_bytes[kTrainWs] = {};
    RowPool& row_pool()
    {
        std::call_once(pool_once, [this] { pool.reset(new RowPool(n_pack_threads)); });
        return *pool;
    }
};

// RAII lease of one staging set of a device slot (waits while every set is in another host thread's call)
struct StageLease {
    DeviceSlot& ds;
    StageSet* s = nullptr;
    RowPool* pool = nullptr;
    explicit StageLease(DeviceSlot& d) : ds(d)
    {
        pool = &ds.row_pool();       // (thread creation happens here, outside ds.mu)
        std::unique_lock<std::mutex> lk(ds.mu);
        for (;;) {
            for (StageSet& c : ds.sets)
                if (!c.leased) { s = &c; break; }
            if (s) break;
            ds.cv.wait(lk);
        }
        s->leased = true;
    }
    ~StageLease()
    {
        { std::lock_guard<std::mutex> g(ds.mu); s->leased = false; }
        ds.cv.notify_one();
    }
    StageLease(const StageLease&) = delete;
    StageLease& operator=(const StageLease&) = delete;
};

// One leasable scratch buffer (rotated rows of the OPQ paths, partial-minimum keys of K > 256).
// A buffer is handed to exactly one call at a time (`leased`, under the codebook mutex); `done` is
// recorded on the call's stream when its last launch has been enqueued, and the next lessee's stream
// waits for it.  A buffer is only ever freed while it is not leased AND its event has completed, so
// no caller can launch on (or be about to launch on) freed memory.
struct ScratchBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool leased = false;
};

struct CodebookDev {
    float* cb = nullptr;     // [M][K][dsub]
    float* frags = nullptr;  // [M][T][S][64]
    float* cc = nullptr;     // [M][k_pad]
    float* cbt = nullptr;    // [M][dsub][KP] transposed image for the small-codebook kernel (K <= 64)
    float* fragp = nullptr;  // [NP][dsub][64] block-diagonal pair fragments + [NP][2][16] norms (K <= 16: kernels_pair16.hip.h)
    uint32_t* vor2_tab = nullptr;   // dsub == 2, K <= 256: candidate tables of kernels_vor2.hip.h (vor2_prep.h) ...
    uint32_t* vor2_off = nullptr;   // ... and the [M + 1] word offsets of the regions
    float* P = nullptr;      // [d][d]   x.dot(P)
    float* PT = nullptr;     // [d][d]   r.dot(P^T)
    int* err = nullptr;      // [0] unused, [1] "some ||c||^2 not finite" (k_check_norms), [2 .. 2 + kErrSlots):
                             // "code >= K / row index out of range seen by reconstruct", one flag per caller stream
    std::vector<hipStream_t> err_streams;  // stream of flag slot i (under cb->mu); least recently used slot is recycled
    std::vector<uint64_t> err_used;        // last use of slot i (err_clock ticks)
    std::vector<hipEvent_t> err_done;      // recorded on the slot's stream behind the last kernels that may raise the flag
    uint64_t err_clock = 0;
    std::vector<ScratchBuf> pool;          // under cb->mu; kScratchPoolMax x kScratchLevels slots, sized at creation (elements never move)
};

}  // namespace pqh

struct pqhip_ctx {
    std::vector<std::unique_ptr<pqh::DeviceSlot>> devs;
    pqh::Options opt;
};

struct pqhip_matrix {
    pqhip_ctx* ctx = nullptr;
    int slot = 0;
    float* d = nullptr;
    int64_t rows = 0, cols = 0;
};

struct pqhip_codebook {
    pqhip_ctx* ctx = nullptr;
    int64_t M = 0, K = 0, dsub = 0, d = 0;
    bool has_proj = false;
    // MFMA encode geometry (0 = shape not covered, anchor kernel is used)
    int T = 0, DP = 0, k_pad = 0;
    bool wide = false;      // 128 < dsub <= 1,024: groups of 32 T <= 128 centroids through k_encode_mfma_wide / _wide2 (kernels_mfma_wide.hip.h)
    int KP = 0;             // small codebooks (K <= 64, instantiated dsub): padded centroid count of the VALU kernel
    bool pair16 = false;    // K <= 16 and dsub in {2, 4, 8, 16}: the two-subquantizers-per-tile kernel applies
    bool vor2 = false;      // dsub == 2, K <= 256, immutable centroids within range: the candidate-list kernel applies
    uint32_t vor2_max_region_words = 0;
    int groups = 1;         // K > 256: groups of 256 centroids (8 tiles each) merged through 64-bit keys
    bool norms_ok = false;  // all ||c||^2 finite and < 2^100
    int variant = 0;        // pqhip_set_encode_variant
    std::vector<pqh::CodebookDev> dev;
    std::atomic<const char*> last_kernel{""};
    std::mutex mu;  // guards the scratch pools and the stream -> flag-slot tables
    std::condition_variable cv;  // a scratch buffer was released
};

namespace pqh {

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
inline int cus_of(pqhip_codebook* cb, int slot) { return cb->ctx->devs[slot]->n_cus; }

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int32_t alloc(size_t bytes)
    {
        HIPCHK(hipMalloc(&p, bytes ? bytes : 1));
        return PQHIP_OK;
    }
};

// In-kernel s_memtime stamps (diagnostic builds only): a zeroed device buffer handed to the kernel, read back and
// summarised on stderr after the launch.  In the default build `want` is false by construction and nothing is allocated.
struct StampRun {
    DevBuf buf;
    size_t n = 0;
    int32_t begin(bool want, size_t n_words, hipStream_t st)
    {
        if (!want) return PQHIP_OK;
        n = n_words;
        PQCHK(buf.alloc(n * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(buf.p, 0, n * sizeof(unsigned long long), st));
        return PQHIP_OK;
    }
    unsigned long long* ptr() const { return (unsigned long long*)buf.p; }
    // synchronous: copies the stamps back; `h` receives them
    int32_t fetch(hipStream_t st, std::vector<unsigned long long>& h)
    {
        h.assign(n, 0);
        if (!buf.p) return PQHIP_OK;
        HIPCHK(hipMemcpyAsync(h.data(), buf.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return PQHIP_OK;
    }
    // the five-word-per-wave layout of the encode / fused kernels: {tiles, phase A cycles, phase B cycles, wave life, real time}
    int32_t report5(hipStream_t st, const char* what, const char* a_name, const char* b_name);
};

// ---- pqhip_ctx.hip -------------------------------------------------------------------------------------------
int32_t ensure_staging(Staging& s, size_t in_bytes, size_t out_bytes);
void free_staging(Staging& s);
int32_t ensure_ws(DeviceSlot& ds, int i, size_t bytes);
int pack_threads(size_t n_devs);

// ---- pqhip_codebook.hip --------------------------------------------------------------------------------------
int32_t codebook_create_impl(pqhip_ctx* ctx, const float* quantizers, int64_t M, int64_t K, int64_t dsub,
                             const float* projection, int only_slot, pqhip_codebook** out);
int32_t prepare_codebook_dev(pqhip_codebook* cb, int slot, hipStream_t st, bool* norms_ok);
int32_t prepare_codebook_async(pqhip_codebook* cb, int slot, hipStream_t st);
int32_t lease_scratch(pqhip_codebook* cb, int slot, size_t bytes, hipStream_t st, int* out_idx, void** out_p);
void release_scratch(pqhip_codebook* cb, int slot, int idx, hipStream_t st);
struct ScratchLease {
    pqhip_codebook* cb;
    int slot, idx = -1;
    hipStream_t st;
    void* p = nullptr;       // the leased buffer (copied under cb->mu by lease_scratch; never read from the pool again)
    ScratchLease(pqhip_codebook* c, int s, hipStream_t t) : cb(c), slot(s), st(t) {}
    ~ScratchLease() { if (idx >= 0) release_scratch(cb, slot, idx, st); }
    int32_t acquire(size_t bytes) { return lease_scratch(cb, slot, bytes, st, &idx, &p); }
    void* ptr() const { return p; }
};
// Device flag of "code >= K / row index out of range" for the calls of stream `st` (one slot per caller stream).
// The object marks the END of the launches that may raise it: its destructor records the slot's event on `st`, and a
// stream that later inherits a recycled slot waits for that event before clearing the flag (the clear used
// to be ordered on the NEW stream only, so kernels still in flight on the old one could raise it afterwards).
struct ErrFlag {
    pqhip_codebook* cb;
    int slot, idx;
    hipStream_t st;
    int* flag;
    ErrFlag(pqhip_codebook* cb, int slot, hipStream_t st);
    ~ErrFlag();
    ErrFlag(const ErrFlag&) = delete;
    ErrFlag& operator=(const ErrFlag&) = delete;
};

// ---- pqhip_encode.hip ----------------------------------------------------------------------------------------
// PQ encode of device-resident, already rotated rows.  bad_flag != nullptr: the matrix-core kernel is launched whatever
// the host last knew about the centroid norms and consults the device flag itself (captured k-means iterations).
int32_t encode_plain_dev(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs, void* d_codes, int code_bytes,
                         int64_t o_rs, hipStream_t st, const int* bad_flag = nullptr, bool beside_update = false);

// ---- pqhip_rotate.hip ----------------------------------------------------------------------------------------
struct RotGather {           // rows gathered from the codebook inside the rotation kernel (OPQ reconstruct / lookup)
    const uint8_t* codes = nullptr;
    int64_t c_rs = 0;
    const float* cb = nullptr;
    int K = 0, dsub = 0;
    unsigned inv_dsub = 0;
    const int64_t* sel_rows = nullptr;
    int64_t n_codes = 0;
    int* err = nullptr;
};
int32_t rotate_dev(const float* d_x, int64_t n, int64_t x_rs, const float* Pm, int d, float* d_out, int64_t o_rs, hipStream_t st,
                   const RotGather* ga = nullptr);

// ---- pqhip_opq.hip -------------------------------------------------------------------------------------------
int32_t quantize_dev_impl(pqhip_codebook* cb, int slot, const float* d_x, int64_t n, int64_t x_rs, void* d_codes, int code_bytes,
                          int64_t o_rs, hipStream_t st);
int32_t reconstruct_dev_impl(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes, int64_t n, int64_t c_rs, float* d_out,
                             int64_t o_rs, hipStream_t st, const int64_t* sel_rows = nullptr, int64_t n_codes = 0,
                             const float* sel_scales = nullptr, int64_t s_rs = 1);
int32_t gather_dev(pqhip_codebook* cb, int slot, const void* d_codes, int code_bytes, int64_t n, int64_t c_rs, float* d_out, int64_t o_rs,
                   hipStream_t st, int* err, const int64_t* sel_rows = nullptr, int64_t n_codes = 0, const float* sel_scales = nullptr,
                   int64_t s_rs = 1);
int resident_wgs(const void* kernel, size_t lds);

}  // namespace pqh
