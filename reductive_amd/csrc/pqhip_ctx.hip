// pqhip_ctx.hip -- contexts and device slots of libpqhip.so: streams, staging sets, packing threads, training
// workspaces, per-context options, the launch log and the status strings (include/pqhip.h).
#include "pqhip_internal.h"

#include <cstdlib>
#include <sched.h>

namespace pqh {

thread_local std::string g_hip_err;

#ifdef PQHIP_DIAG
// Diagnostic builds only (make DIAG=1 / TIMING=1): the PQHIP_DEBUG_* variables of rounds 1-3, read once.
const Diag& diag()
{
    static const Diag d = [] {
        Diag v;
        auto flag = [](const char* n) { return getenv(n) != nullptr; };
        auto num = [](const char* n, int64_t dflt) { const char* e = getenv(n); return e ? (int64_t)atoll(e) : dflt; };
        v.enc_stamp = flag("PQHIP_DEBUG_ENC_STAMP");
        v.rot_stamp = flag("PQHIP_DEBUG_ROT_STAMP");
        v.fused_stamp = flag("PQHIP_DEBUG_FUSED_STAMP");
        v.occ = flag("PQHIP_DEBUG_OCC");
        v.rec_elemwise = flag("PQHIP_DEBUG_REC_ELEMWISE");
        v.adc_any = flag("PQHIP_DEBUG_ADC_ANY");
        v.no_mfma16 = flag("PQHIP_DEBUG_NO_MFMA16");
        v.rot_stamp_file = getenv("PQHIP_DEBUG_ROT_STAMP_FILE");
        v.rpi_min = num("PQHIP_DEBUG_RPI_MIN", 32);
        v.rpi_max = num("PQHIP_DEBUG_RPI_MAX", 1024);
        v.lds_pad = (int)num("PQHIP_DEBUG_LDS_PAD", 0);
        v.rec_wgs = (int)std::max<int64_t>(0, num("PQHIP_DEBUG_REC_WGS", 0));
        v.adc_wgs = (int)std::max<int64_t>(0, num("PQHIP_DEBUG_ADC_WGS", 0));
        v.fused2_tiles = (int)std::max<int64_t>(0, num("PQHIP_DEBUG_FUSED2_TILES", 0));
        const int64_t r = num("PQHIP_DEBUG_ROT_RPW", 0);
        v.rot_rpw = r >= 384 ? (int)((r / 384) * 384) : kRotRowsPerWg;
        return v;
    }();
    return d;
}
#endif

// ---- launch log (thread-local; see pqhip_internal.h) -----------------------------------------------------------
namespace {
struct LaunchLog {
    static constexpr int kMax = 24;
    const char* name[kMax];
    int64_t count[kMax];
    int n = 0;
    std::string text;
};
thread_local LaunchLog g_log;
}  // namespace

void note_kernel(const char* name)
{
    LaunchLog& l = g_log;
    for (int i = 0; i < l.n; ++i)
        if (l.name[i] == name || std::strcmp(l.name[i], name) == 0) { ++l.count[i]; return; }
    if (l.n < LaunchLog::kMax) { l.name[l.n] = name; l.count[l.n] = 1; ++l.n; }
}

int32_t StampRun::report5(hipStream_t st, const char* what, const char* a_name, const char* b_name)
{
    if (!buf.p) return PQHIP_OK;
    std::vector<unsigned long long> h;
    PQCHK(fetch(st, h));
    double tiles = 0, a = 0, b = 0, cyc = 0, rt = 0;
    size_t waves = 0;
    for (size_t i = 0; i + 5 <= n; i += 5)
        if (h[i]) { tiles += (double)h[i]; a += (double)h[i + 1]; b += (double)h[i + 2]; cyc += (double)h[i + 3]; rt += (double)h[i + 4]; ++waves; }
    if (tiles > 0)
        fprintf(stderr, "[pqhip] %s stamps: %zu waves, %.1f tiles/wave, %s %.0f cyc/tile, %s %.0f cyc/tile, wave life %.0f cyc, clock %.0f MHz\n",
                what, waves, tiles / waves, a_name, a / tiles, b_name, b / tiles, cyc / waves, rt > 0 ? cyc / rt * 100.0 : 0.0);
    return PQHIP_OK;
}

// ---- packing threads ---------------------------------------------------------------------------------------------
RowPool::RowPool(int n_threads) : n_(std::max(1, n_threads))
{
    for (int i = 1; i < n_; ++i) th_.emplace_back([this] { worker(); });
}

RowPool::~RowPool()
{
    {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
}

void RowPool::worker()
{
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;          // stop_ and nothing left to do
            t = q_.front();
            q_.pop_front();
        }
        if (t.b < t.e) (*t.call->fn)(t.b, t.e);
        // under mu_: the caller cannot wake up (and destroy `call`) before this thread has let go of it
        std::lock_guard<std::mutex> g(mu_);
        if (--t.call->pending == 0) t.call->done.notify_one();
    }
}

// CPU cores this process may actually use: the scheduler affinity and the cgroup CPU quota (v2 `cpu.max`, v1
// `cpu.cfs_quota_us`), whichever is smaller -- NOT std::thread::hardware_concurrency(), which reports the host's logical
// CPUs (256 on the GPU boxes, of which a one-GPU job is granted 16).
static unsigned usable_cores()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    double quota = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {};
        double per = 0;
        if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
        std::fclose(f);
    } else {
        double q = 0, per = 0;
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lf", &q) != 1) q = 0; std::fclose(g); }
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lf", &per) != 1) per = 0; std::fclose(g); }
        if (q > 0 && per > 0) quota = q / per;
    }
    if (quota >= 1.0) n = std::min<unsigned>(n, (unsigned)(quota + 0.5));
    return std::max(1u, n);
}

// host threads per device slot for packing / draining: the usable cores shared out over the context's device slots, at
// most 16 each (PQHIP_PACK_THREADS overrides the cap).  Round 3 divided the host's LOGICAL cpu count: two slots on a
// 16-core quota ran 2 x 16 packing threads and were throttled -- the two-slot rehearsal slid from 0.78x to 0.59x of one
// slot's rate from box to box while the host code did not change.
int pack_threads(size_t n_devs)
{
    static const unsigned cores = usable_cores();
    static const unsigned cap = [] { const char* e = getenv("PQHIP_PACK_THREADS"); return e ? (unsigned)std::max(1, atoi(e)) : 16u; }();
    return (int)std::max<unsigned>(1, std::min<unsigned>(cap, cores / (unsigned)std::max<size_t>(1, n_devs)));
}

// ---- staging buffers, training workspaces ------------------------------------------------------------------------
int32_t ensure_staging(Staging& s, size_t in_bytes, size_t out_bytes)
{
    if (s.in_bytes < in_bytes) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.d_in) (void)hipFree(s.d_in);
        s.h_in = s.d_in = nullptr;
        s.in_bytes = 0;
        HIPCHK(hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc(&s.d_in, in_bytes));
        s.in_bytes = in_bytes;
    }
    if (s.out_bytes < out_bytes) {
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.d_out) (void)hipFree(s.d_out);
        s.h_out = s.d_out = nullptr;
        s.out_bytes = 0;
        HIPCHK(hipHostMalloc(&s.h_out, out_bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc(&s.d_out, out_bytes));
        s.out_bytes = out_bytes;
    }
    return PQHIP_OK;
}

void free_staging(Staging& s)
{
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    s = Staging();
}

int32_t ensure_ws(DeviceSlot& ds, int i, size_t bytes)
{
    if (ds.ws_bytes[i] >= bytes) return PQHIP_OK;
    if (ds.ws[i]) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(ds.ws[i]); ds.ws[i] = nullptr; ds.ws_bytes[i] = 0; }
    HIPCHK(hipMalloc(&ds.ws[i], bytes));
    ds.ws_bytes[i] = bytes;
    return PQHIP_OK;
}

}  // namespace pqh

using namespace pqh;

extern "C" {

int32_t pqhip_version(void) { return PQHIP_VERSION; }

const char* pqhip_strerror(int32_t s)
{
    switch (s) {
    case PQHIP_OK: return "ok";
    case PQHIP_EINVAL: return "invalid argument";
    case PQHIP_ESHAPE: return "shape mismatch (quantizer / vector / output lengths)";
    case PQHIP_ECODE_RANGE: return "code out of range (>= number of centroids)";
    case PQHIP_EINDEX_WIDTH: return "cannot store centroids in quantizer index type";
    case PQHIP_ENODEV: return "no usable HIP device";
    case PQHIP_EHIP: return "HIP runtime error";
    case PQHIP_ENOMEM: return "out of memory";
    case PQHIP_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
    }
}

const char* pqhip_last_hip_error(void) { return g_hip_err.c_str(); }

void pqhip_launch_log_reset(void) { g_log.n = 0; }

const char* pqhip_launch_log(void)
{
    LaunchLog& l = g_log;
    l.text.clear();
    for (int i = 0; i < l.n; ++i) {
        if (i) l.text += " + ";
        l.text += l.name[i];
        if (l.count[i] > 1) l.text += " x" + std::to_string(l.count[i]);
    }
    return l.text.c_str();
}

int32_t pqhip_device_count(int32_t* out)
{
    if (!out) return PQHIP_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        *out = 0;
        return PQHIP_ENODEV;
    }
    *out = n;
    return PQHIP_OK;
}

int32_t pqhip_ctx_create(const int32_t* devices, int32_t n_devices, pqhip_ctx** out)
{
    if (!out || n_devices < 0) return PQHIP_EINVAL;
    *out = nullptr;
    int32_t avail = 0;
    PQCHK(pqhip_device_count(&avail));
    std::vector<int> ords;
    if (!devices || n_devices == 0) {
        for (int i = 0; i < avail; ++i) ords.push_back(i);
    } else {
        for (int i = 0; i < n_devices; ++i) {
            if (devices[i] < 0 || devices[i] >= avail) return PQHIP_ENODEV;
            ords.push_back(devices[i]);
        }
    }
    struct CtxGuard { pqhip_ctx* p; ~CtxGuard() { if (p) pqhip_ctx_destroy(p); } } ctx{new pqhip_ctx()};
    {   // PQHIP_FUSED2_OPQ=0: the deployment switch of the fused OPQ encode (also an option: "opq_fused")
        const char* e = getenv("PQHIP_FUSED2_OPQ");
        if (e && e[0] == '0') ctx.p->opt.opq_fused = 0;
    }
    for (int o : ords) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, o));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_hip_err = std::string("device is ") + prop.gcnArchName + ", library is built for gfx950";
            return PQHIP_ENODEV;
        }
        std::unique_ptr<DeviceSlot> ds(new DeviceSlot());
        ds->ordinal = o;
        if (prop.multiProcessorCount > 0) ds->n_cus = prop.multiProcessorCount;
        SET_DEVICE(o);
        ctx.p->devs.push_back(std::move(ds));   // (before the streams: a failure below destroys what exists through the context)
        DeviceSlot& d = *ctx.p->devs.back();
        HIPCHK(hipStreamCreateWithFlags(&d.stream[0], hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&d.stream[1], hipStreamNonBlocking));
        for (StageSet& c : d.sets)
            for (int i = 0; i < 2; ++i) HIPCHK(hipStreamCreateWithFlags(&c.stream[i], hipStreamNonBlocking));
        d.n_pack_threads = pack_threads(ords.size());
    }
    *out = ctx.p;
    ctx.p = nullptr;
    return PQHIP_OK;
}

void pqhip_ctx_destroy(pqhip_ctx* ctx)
{
    if (!ctx) return;
    for (auto& ds : ctx->devs) {
        DeviceGuard dg(ds->ordinal);
        for (int i = 0; i < 2; ++i)
            if (ds->stream[i]) { (void)hipStreamSynchronize(ds->stream[i]); (void)hipStreamDestroy(ds->stream[i]); }
        for (StageSet& c : ds->sets)
            for (int i = 0; i < 2; ++i) {
                if (c.stream[i]) { (void)hipStreamSynchronize(c.stream[i]); (void)hipStreamDestroy(c.stream[i]); }
                free_staging(c.st[i]);
            }
        for (int i = 0; i < kTrainWs; ++i)
            if (ds->ws[i]) (void)hipFree(ds->ws[i]);
    }
    delete ctx;
}

int32_t pqhip_ctx_n_devices(const pqhip_ctx* ctx) { return ctx ? (int32_t)ctx->devs.size() : 0; }

int32_t pqhip_ctx_set_option(pqhip_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name || value < 0) return PQHIP_EINVAL;
    Options& o = ctx->opt;
    struct { const char* n; std::atomic<int64_t>* v; } table[] = {
        {"kmeans_window_rows", &o.kmeans_window_rows}, {"kmeans_lane_form", &o.kmeans_lane_form},
        {"kmeans_no_graph", &o.kmeans_no_graph},       {"opq_scratch_rows", &o.opq_scratch_rows},
        {"opq_fused", &o.opq_fused},                   {"opq_gather_rotation", &o.opq_gather_rotation},
        {"adc_single_query", &o.adc_single_query},     {"cross_product_exact", &o.cross_product_exact},
        {"cross_product_group_bytes", &o.cross_product_group_bytes}, {"lookup_two_pass", &o.lookup_two_pass},
        {"candidate_tables", &o.candidate_tables}};
    for (auto& t : table)
        if (std::strcmp(t.n, name) == 0) { t.v->store(value, std::memory_order_relaxed); return PQHIP_OK; }
    return PQHIP_EINVAL;
}

}  // extern "C"
