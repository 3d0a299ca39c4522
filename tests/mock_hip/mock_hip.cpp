// A host-memory stand-in for the HIP runtime, for ONE purpose: running the HOST logic of libpqhip.so
// (strided packing into pinned staging, double-buffered drains, row sharding over device slots, the
// scratch-lease pool, flag tables, handle lifetimes) under AddressSanitizer + UBSan on a machine without
// a GPU.  "Device" memory is malloc'd host memory, copies are memcpy, streams and events complete
// immediately, kernel launches do nothing (results are garbage; the sanitizers watch the memory traffic
// around them).  Test infrastructure only -- the product never links this.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 2; return hipSuccess; }            // two "devices": the sharder runs
static thread_local int g_dev = 0;
hipError_t hipGetDevice(int* d) { *d = g_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { g_dev = d; return hipSuccess; }
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

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }   // eager k-means loop
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }

// exact-size allocations: an overrun of a staging / scratch / code buffer by the host logic is an ASan report
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t i = 0; i < h; ++i) std::memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
// page-locking of caller memory (the zero-copy input leg): every third request is refused, so that both the
// registered leg and its fall-back to the packing path run under the sanitizers; the byte range is touched so that a
// span that overruns the caller's buffer is an ASan report
static std::atomic<unsigned> g_reg_calls{0};
hipError_t hipHostRegister(void* p, size_t n, unsigned)
{
    if (g_reg_calls.fetch_add(1) % 3 == 2) return hipErrorInvalidValue;
    volatile const char* c = (volatile const char*)p;
    if (n) { (void)c[0]; (void)c[n - 1]; }
    return hipSuccess;
}
hipError_t hipHostUnregister(void*) { return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }   // nothing runs
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* m, hipStream_t* s)
{
    *g = dim3(1); *b = dim3(1); *m = 0; *s = nullptr;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h[1]; return h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}

}  // extern "C"
