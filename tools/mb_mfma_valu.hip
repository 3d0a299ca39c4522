// Micro-benchmark (diagnostic, not product): does v_mfma_f32_32x32x2_f32 overlap with VALU work on
// gfx950?  Each wave runs REP x { 1 MFMA (dependent chain) ; NV VALU ops } and reports shader cycles
// per MFMA.  Variants: NV VALU per MFMA in {0,2,4,6,8,10,12,16}; 1 or 2 waves per SIMD; VALU kinds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF "v_mfma_f32_32x32x2_f32 %[acc], %[a], %[b], %[acc]\n"
#define V0 "v_fma_f32 %[t0], %[a], %[b], %[t0]\n"
#define V1 "v_fma_f32 %[t1], %[a], %[b], %[t1]\n"
#define V2 "v_fma_f32 %[t2], %[a], %[b], %[t2]\n"
#define V3 "v_fma_f32 %[t3], %[a], %[b], %[t3]\n"
#define C0 "v_cmp_lt_f32 vcc, %[t0], %[t1]\n"
#define C1 "v_cndmask_b32 %[t2], %[t2], %[t3], vcc\n"
#define P0 "v_pk_fma_f32 %[p0], %[p2], %[p2], %[p0]\n"
#define P1 "v_pk_fma_f32 %[p1], %[p2], %[p2], %[p1]\n"
#define M3 "v_min3_f32 %[t0], %[t0], %[t1], %[t2]\n"

template <int KIND>
__global__ void kern(float* out, unsigned long long* cyc, int iters, float av, float bv)
{
    f32x16 acc = {0};
    float a = av + threadIdx.x * 1e-9f, b = bv;
    float t0 = a, t1 = b, t2 = a + b, t3 = a - b;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a, b}, p1 = {b, a}, p2 = {0.5f, 0.25f};
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define BODY(S) asm volatile(S S S S S S S S S S : [acc] "+v"(acc), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [p0] "+v"(p0), [p1] "+v"(p1) : [a] "v"(a), [b] "v"(b), [p2] "v"(p2) : "vcc")
        if (KIND == 0) BODY(MF);
        if (KIND == 2) BODY(MF V0 V1);
        if (KIND == 4) BODY(MF V0 V1 V2 V3);
        if (KIND == 6) BODY(MF V0 V1 V2 V3 V0 V1);
        if (KIND == 8) BODY(MF V0 V1 V2 V3 V0 V1 V2 V3);
        if (KIND == 10) BODY(MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1);
        if (KIND == 12) BODY(MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);
        if (KIND == 16) BODY(MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);
        if (KIND == 100) BODY(V0 V1 V2 V3 V0 V1 V2 V3);              // VALU only, 8 per "slot"
        if (KIND == 108) BODY(MF C0 C1 C0 C1 C0 C1 C0 C1);            // 8 cmp/cndmask
        if (KIND == 208) BODY(MF P0 P1 P0 P1);                        // 4 packed fma (= 8 elements)
        if (KIND == 308) BODY(MF M3 M3 M3 M3 M3 M3 M3 M3);            // 8 min3
        if (KIND == 101) BODY(C0 C1 C0 C1 C0 C1 C0 C1);               // cmp/cndmask only
        if (KIND == 201) BODY(P0 P1 P0 P1);                           // packed only
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = t0 + t1 + t2 + t3 + p0[0] + p1[1];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}

template <int KIND>
void run(const char* name, int block)
{
    const int grid = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    const int nw = grid * block / 64;
    hipMalloc(&out, sizeof(float) * grid * block);
    hipMalloc(&cyc, sizeof(unsigned long long) * nw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<KIND><<<grid, block>>>(out, cyc, 10, 1.0f, 0.5f);
    hipEventRecord(e0);
    kern<KIND><<<grid, block>>>(out, cyc, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double slots = iters * 10.0;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock); report both tick-based and wall-based
    printf("%-28s waves/SIMD=%d  wall=%.3f ms  ns/slot(wave)=%.2f  memtime ticks/slot med=%.3f\n", name,
           block / 256, ms, ms * 1e6 / slots, h[nw / 2] / slots);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int block : {256, 512}) {
        run<0>("mfma only", block);
        run<2>("mfma + 2 fma", block);
        run<4>("mfma + 4 fma", block);
        run<6>("mfma + 6 fma", block);
        run<8>("mfma + 8 fma", block);
        run<10>("mfma + 10 fma", block);
        run<12>("mfma + 12 fma", block);
        run<16>("mfma + 16 fma", block);
        run<100>("8 fma only", block);
        run<108>("mfma + 4x(cmp,cndmask)", block);
        run<101>("4x(cmp,cndmask) only", block);
        run<208>("mfma + 4 pk_fma", block);
        run<201>("4 pk_fma only", block);
        run<308>("mfma + 8 min3", block);
    }
    return 0;
}
