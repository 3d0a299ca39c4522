// A host-memory stand-in for the HIP runtime, for ONE purpose: running the HOST logic of libpqhip.so
// (strided packing into pinned staging, double-buffered drains, row sharding over device slots, the
// scratch-lease pool, flag tables, handle lifetimes) under AddressSanitizer + UBSan / ThreadSanitizer on a
// machine without a GPU.  "Device" memory is malloc'd host memory, copies are memcpy, streams and events
// complete immediately, kernel launches do nothing (results are garbage; the sanitizers watch the memory
// traffic around them).  Test infrastructure only -- the product never links this.
//
// Round 4 ("multi-device code has only ever met one ordinal"): the mock reports
// MOCK_HIP_DEVICES devices (default 2, the multi-GPU driver sets 8) and TAGS every allocation, stream and
// event with the device that was current when it was created.  Using one under another current device --
// a launch or a copy on a stream of device A while device B is current, device memory of A copied or cleared
// through a stream of B, an event of A recorded on a stream of B, a synchronous copy of A's memory while B is
// current -- is a VIOLATION: it is printed, counted (mock_hip_violations()) and fails the driver.  Handles the
// mock did not create (the caller's own streams, host vectors standing in for caller-owned device memory) are
// not judged.  Destroying / freeing is allowed from any device, as in the real runtime.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {

std::mutex g_mu;
std::map<const char*, std::pair<size_t, int>> g_allocs;   // device allocations: base -> (bytes, device)
std::map<const void*, int> g_streams, g_events;           // handle -> device
std::atomic<int> g_violations{0};
thread_local int g_dev = 0;

int n_devices()
{
    static const int n = [] { const char* e = std::getenv("MOCK_HIP_DEVICES"); const int v = e ? std::atoi(e) : 2; return v >= 1 && v <= 64 ? v : 2; }();
    return n;
}

thread_local bool g_quiet = false;          // the self-test counts its deliberate misuses without printing them
void violation(const char* what, int have, int want)
{
    g_violations.fetch_add(1);
    if (g_quiet) return;
    std::fprintf(stderr, "MOCK-HIP DEVICE MISMATCH: %s: object of device %d used while device %d is current / on a stream of device %d\n",
                 what, have, g_dev, want);
}

// device of the allocation that contains p (-1: not device memory the mock handed out)
int dev_of_ptr(const void* p)
{
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_allocs.upper_bound((const char*)p);
    if (it == g_allocs.begin()) return -1;
    --it;
    return ((const char*)p < it->first + it->second.first) ? it->second.second : -1;
}
int dev_of(const std::map<const void*, int>& m, const void* h)
{
    std::lock_guard<std::mutex> g(g_mu);
    auto it = m.find(h);
    return it == m.end() ? -1 : it->second;
}
// the device a stream's work runs on: its own for streams the mock created, the current one for the null stream and
// for foreign handles
int stream_dev(hipStream_t s)
{
    const int d = s ? dev_of(g_streams, s) : -1;
    return d >= 0 ? d : g_dev;
}
void check_stream_current(const char* what, hipStream_t s)
{
    const int d = s ? dev_of(g_streams, s) : -1;
    if (d >= 0 && d != g_dev) violation(what, d, d);
}
void check_ptr_on(const char* what, const void* p, int want_dev)
{
    const int d = dev_of_ptr(p);
    if (d >= 0 && d != want_dev) violation(what, d, want_dev);
}

}  // namespace

extern "C" {

int mock_hip_violations(void) { return g_violations.load(); }
int mock_hip_registrations(void);     // successful hipHostRegister calls so far
void mock_hip_reset_violations(void) { g_violations.store(0); }

hipError_t hipGetDeviceCount(int* n) { *n = n_devices(); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_dev; return hipSuccess; }
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= n_devices()) return hipErrorInvalidDevice;
    g_dev = d;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->gcnArchName, "gfx950:mock");
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock"; }

static hipStream_t new_stream()
{
    hipStream_t s = (hipStream_t)std::malloc(8);
    std::lock_guard<std::mutex> g(g_mu);
    g_streams[s] = g_dev;
    return s;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s)
{
    { std::lock_guard<std::mutex> g(g_mu); g_streams.erase(s); }
    std::free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned)
{
    check_stream_current("hipStreamWaitEvent(stream)", s);     // (waiting for an event of another device is legal)
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }   // eager k-means loop
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned)
{
    *e = (hipEvent_t)std::malloc(8);
    std::lock_guard<std::mutex> g(g_mu);
    g_events[*e] = g_dev;
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e)
{
    { std::lock_guard<std::mutex> g(g_mu); g_events.erase(e); }
    std::free(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    check_stream_current("hipEventRecord(stream)", s);
    const int de = dev_of(g_events, e);
    if (de >= 0 && de != stream_dev(s)) violation("hipEventRecord(event)", de, stream_dev(s));
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }

// exact-size allocations: an overrun of a staging / scratch / code buffer by the host logic is an ASan report
hipError_t hipMalloc(void** p, size_t n)
{
    *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(g_mu);
    g_allocs[(const char*)*p] = {n ? n : 1, g_dev};
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    { std::lock_guard<std::mutex> g(g_mu); g_allocs.erase((const char*)p); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind)
{
    check_ptr_on("hipMemcpy(dst)", d, g_dev);
    check_ptr_on("hipMemcpy(src)", s, g_dev);
    std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st)
{
    check_stream_current("hipMemcpyAsync(stream)", st);
    check_ptr_on("hipMemcpyAsync(dst)", d, stream_dev(st));
    check_ptr_on("hipMemcpyAsync(src)", s, stream_dev(st));
    std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st)
{
    check_stream_current("hipMemcpy2DAsync(stream)", st);
    check_ptr_on("hipMemcpy2DAsync(dst)", d, stream_dev(st));
    check_ptr_on("hipMemcpy2DAsync(src)", s, stream_dev(st));
    for (size_t i = 0; i < h; ++i) std::memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st)
{
    check_stream_current("hipMemsetAsync(stream)", st);
    check_ptr_on("hipMemsetAsync(dst)", d, stream_dev(st));
    std::memset(d, v, n);
    return hipSuccess;
}
// page-locking of caller memory (the zero-copy input leg): every third request is refused, so that both the
// registered leg and its fall-back to the packing path run under the sanitizers; the byte range is touched so that a
// span that overruns the caller's buffer is an ASan report, and a page registered twice is refused as the real runtime does
static std::atomic<unsigned> g_reg_calls{0}, g_reg_ok{0};
static std::map<uintptr_t, uintptr_t> g_registered;   // [lo, hi) of live registrations (under g_mu)
hipError_t hipHostRegister(void* p, size_t n, unsigned)
{
    if (g_reg_calls.fetch_add(1) % 3 == 2) return hipErrorInvalidValue;
    volatile const char* c = (volatile const char*)p;
    if (n) { (void)c[0]; (void)c[n - 1]; }
    const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)4095, hi = ((uintptr_t)p + n + 4095) & ~(uintptr_t)4095;
    std::lock_guard<std::mutex> g(g_mu);
    for (auto& r : g_registered)
        if (lo < r.second && r.first < hi) return hipErrorHostMemoryAlreadyRegistered;
    g_registered[lo] = hi;
    g_reg_ok.fetch_add(1);
    return hipSuccess;
}
hipError_t hipHostUnregister(void* p)
{
    std::lock_guard<std::mutex> g(g_mu);
    g_registered.erase((uintptr_t)p & ~(uintptr_t)4095);
    return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
// nothing runs; the launch must target a stream of the CURRENT device (the library sets the device per entry point)
static thread_local hipStream_t g_cfg_stream = nullptr;
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t s)
{
    check_stream_current("hipLaunchKernel(stream)", s);
    return hipSuccess;
}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t s) { g_cfg_stream = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* m, hipStream_t* s)
{
    *g = dim3(1); *b = dim3(1); *m = 0; *s = g_cfg_stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h[1]; return h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}

int mock_hip_registrations(void) { return (int)g_reg_ok.load(); }

// Proves that the tagging works (called by the multi-device driver before it trusts a clean run): each misuse below
// must be counted, then the counter is cleared.  Returns the number of DETECTED misuses (expected: 5).
int mock_hip_selftest(void)
{
    if (n_devices() < 2) return -1;
    const int before = g_violations.load();
    g_quiet = true;
    int saved = 0;
    (void)hipGetDevice(&saved);
    void* mem0 = nullptr;
    hipStream_t s0 = nullptr;
    hipEvent_t e0 = nullptr;
    (void)hipSetDevice(0);
    (void)hipMalloc(&mem0, 64);
    (void)hipStreamCreateWithFlags(&s0, 0);
    (void)hipEventCreateWithFlags(&e0, 0);
    (void)hipSetDevice(1);
    hipStream_t s1 = nullptr;
    (void)hipStreamCreateWithFlags(&s1, 0);
    char host[64];
    (void)hipMemsetAsync(mem0, 0, 64, s1);                                   // 1: memory of device 0 through a stream of device 1
    (void)hipMemcpyAsync(host, mem0, 64, hipMemcpyDeviceToHost, s1);         // 2: the same for a copy
    (void)hipLaunchKernel(nullptr, dim3(1), dim3(1), nullptr, 0, s0);        // 3: a launch on device 0's stream while device 1 is current
    (void)hipEventRecord(e0, s1);                                            // 4: an event of device 0 on a stream of device 1
    (void)hipMemcpy(host, mem0, 64, hipMemcpyDeviceToHost);                  // 5: synchronous copy of device 0's memory under device 1
    const int seen = g_violations.load() - before;
    (void)hipStreamDestroy(s1);
    (void)hipEventDestroy(e0);
    (void)hipStreamDestroy(s0);
    (void)hipFree(mem0);
    (void)hipSetDevice(saved);
    g_violations.store(before);
    g_quiet = false;
    return seen;
}

}  // extern "C"
