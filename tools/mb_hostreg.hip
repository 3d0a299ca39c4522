// Micro-benchmark (diagnostic, not product): what does hipHostRegister / hipHostUnregister of a pageable host buffer
// cost per byte, and how fast is an H2D copy from the registered range vs from pinned staging?  Decides whether a
// zero-copy leg can beat "pack into pinned staging + DMA" for one-shot caller buffers.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t bytes = mb << 20;
    char* h = (char*)aligned_alloc(4096, bytes);
    memset(h, 1, bytes);                      // touched: pages exist
    void* d; hipMalloc(&d, bytes);
    void* pin; hipHostMalloc(&pin, bytes, hipHostMallocDefault); memset(pin, 2, bytes);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
        double t1 = now();
        if (e != hipSuccess) { printf("hipHostRegister failed: %s\n", hipGetErrorString(e)); return 1; }
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        double t2 = now();
        hipHostUnregister(h);
        double t3 = now();
        hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        double t4 = now();
        hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);        // pageable copy by the runtime
        double t5 = now();
        printf("%zu MiB: register %.1f ms (%.1f GB/s)  H2D from registered %.1f ms (%.1f GB/s)  unregister %.1f ms  H2D from pinned %.1f ms (%.1f GB/s)  pageable hipMemcpy %.1f ms (%.1f GB/s)\n",
               mb, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3,
               (t4 - t3) * 1e3, bytes / (t4 - t3) / 1e9, (t5 - t4) * 1e3, bytes / (t5 - t4) / 1e9);
    }
    // registration from several threads at once (disjoint 64 MiB pieces)
    for (int nt : {2, 4, 8, 16}) {
        const size_t piece = 64ull << 20;
        const size_t np = bytes / piece;
        double t0 = now();
        std::vector<std::thread> th;
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        for (int t = 0; t < nt; ++t) th.emplace_back([&] { for (;;) { size_t i = next.fetch_add(1); if (i >= np) break; if (hipHostRegister(h + i * piece, piece, hipHostRegisterDefault) != hipSuccess) ++bad; } });
        for (auto& t : th) t.join();
        double t1 = now();
        for (size_t i = 0; i < np; ++i) hipHostUnregister(h + i * piece);
        double t2 = now();
        printf("%d threads: register %zu x 64 MiB in %.1f ms (%.1f GB/s), failures %d, unregister all %.1f ms\n", nt, np, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, bad.load(), (t2 - t1) * 1e3);
    }
    return 0;
}
