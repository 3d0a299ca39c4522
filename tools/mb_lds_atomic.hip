// Diagnostic micro-benchmark #2: cost of LDS atomics (min on 64-bit keys) and a few VALU forms beside
// a dependent v_mfma_f32_32x32x2_f32 chain, 2 waves per SIMD (the encode kernel's occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MF "v_mfma_f32_32x32x2_f32 %[acc], %[a], %[b], %[acc]\n"
#define V0 "v_fma_f32 %[t0], %[a], %[b], %[t0]\n"
#define V1 "v_fma_f32 %[t1], %[a], %[b], %[t1]\n"
#define V2 "v_fma_f32 %[t2], %[a], %[b], %[t2]\n"
#define V3 "v_fma_f32 %[t3], %[a], %[b], %[t3]\n"
#define A64 "ds_min_i64 %[addr], %[k64]\n"
#define AU64 "ds_min_u64 %[addr], %[k64]\n"
#define AF32 "ds_min_f32 %[addr], %[t0]\n"
#define W64 "ds_write_b64 %[addr], %[k64]\n"
#define PA "v_pk_add_f32 %[p0], %[p2], %[p0]\n"
#define PF "v_pk_fma_f32 %[p1], %[p2], %[p2], %[p1]\n"
#define CE "v_cmp_lt_f32_e64 %[sm], %[t0], %[t1]\n"
#define SE "v_cndmask_b32_e64 %[t2], %[t2], %[t3], %[sm]\n"
#define MN "v_min_f32 %[t0], %[t0], %[t1]\n"
#define WAITL "s_waitcnt lgkmcnt(0)\n"

template <int KIND>
__global__ void kern(float* out, unsigned long long* cyc, int iters, float av, float bv)
{
    __shared__ unsigned long long slots[512];
    slots[threadIdx.x] = ~0ull >> 1;
    __syncthreads();
    f32x16 acc = {0};
    float a = av + threadIdx.x * 1e-9f, b = bv;
    float t0 = a, t1 = b, t2 = a + b, t3 = a - b;
    f2 p0 = {a, b}, p1 = {b, a}, p2 = {0.5f, 0.25f};
    unsigned addr = (unsigned)(size_t)(&slots[threadIdx.x]);
    unsigned long long k64 = ((unsigned long long)__float_as_uint(a) << 32) | threadIdx.x;
    unsigned long long sm = 0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS : [acc] "+v"(acc), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [p0] "+v"(p0), [p1] "+v"(p1), [sm] "+s"(sm) : [a] "v"(a), [b] "v"(b), [p2] "v"(p2), [addr] "v"(addr), [k64] "v"(k64) : "vcc", "memory"
#define BODY(S) asm volatile(S S S S S S S S S S OPS)
#define BODY1(S) asm volatile(S OPS)
        if (KIND == 0) BODY(MF);
        if (KIND == 1) BODY(MF A64 A64 A64 A64 A64 A64 A64 A64);           // 8 LDS i64 min per MFMA
        if (KIND == 2) BODY(MF A64 A64);                                    // 1.6/MFMA ~ 16 per 10
        if (KIND == 3) BODY(A64 A64 A64 A64 A64 A64 A64 A64);              // atomics only
        if (KIND == 4) BODY(MF AF32 AF32);
        if (KIND == 5) BODY(MF W64 W64);
        if (KIND == 6) BODY(MF V0 V1 A64 A64 V2 V3);                       // 2 el: 4 VALU... + 2 atomics
        if (KIND == 7) BODY(MF PA PF A64 A64);                             // packed add+fma (2 el) + 2 atomics
        if (KIND == 8) BODY(MF PA PF PA PF PA PF PA PF);                   // 8 packed
        if (KIND == 9) BODY(MF CE SE CE SE CE SE CE SE);                   // e64 cmp/cndmask with SGPR mask
        if (KIND == 10) BODY(MF MN MN MN MN MN MN MN MN);
        if (KIND == 11) BODY1(MF MF MF MF MF MF MF MF MF MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);  // grouped 10 MFMA + 80 VALU
        if (KIND == 21) BODY1(MF MF MF MF MF MF MF MF MF MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);  // 10 MFMA + 16 VALU
        if (KIND == 22) BODY1(MF MF MF MF MF MF MF MF MF MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);  // + 24
        if (KIND == 23) BODY1(MF MF MF MF MF MF MF MF MF MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3);  // + 40
        if (KIND == 24) BODY1(MF V0 V1 V2 MF V3 V0 MF V1 V2 MF V3 V0 V1 MF V2 V3 MF V0 V1 V2 MF V3 V0 MF V1 V2 MF V3 V0 V1 MF V2 V3);  // interleaved 24 over 10
        if (KIND == 25) BODY1(MF MF MF MF MF MF MF MF MF MF V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 V0 V1 V2 V3 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64 A64);  // 10 MFMA + 24 VALU + 16 atomics
        if (KIND == 26) BODY1(MF A64 V0 V1 MF A64 A64 V2 V3 MF A64 V0 V1 V2 MF A64 A64 V3 V0 MF A64 V1 V2 V3 MF A64 A64 V0 V1 MF A64 V2 V3 MF A64 A64 V0 V1 V2 MF A64 V3 V0 MF A64 A64 V1 V2 V3);  // interleaved everything
        if (KIND == 12) BODY(MF A64 A64 WAITL);
        if (KIND == 13) BODY(MF AU64 AU64);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float s = t0 + t1 + t2 + t3 + p0[0] + p1[1] + (float)slots[threadIdx.x ^ 1] + (float)sm;
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
    (void)hipMalloc(&out, sizeof(float) * grid * block);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * nw);
    kern<KIND><<<grid, block>>>(out, cyc, 10, 1.0f, 0.5f);
    kern<KIND><<<grid, block>>>(out, cyc, iters, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s waves/SIMD=%d  cycles per 10-slot body /10 (per wave) = %.2f  -> per SIMD-slot %.2f\n", name,
           block / 256, h[nw / 2] / (iters * 10.0), h[nw / 2] / (iters * 10.0) / (block / 256));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int block : {256, 512}) {
        run<0>("mfma only", block);
        run<1>("mfma + 8 ds_min_i64", block);
        run<2>("mfma + 2 ds_min_i64", block);
        run<13>("mfma + 2 ds_min_u64", block);
        run<12>("mfma + 2 ds_min_i64 + wait", block);
        run<3>("8 ds_min_i64 only", block);
        run<4>("mfma + 2 ds_min_f32", block);
        run<5>("mfma + 2 ds_write_b64", block);
        run<6>("mfma + 4 fma + 2 ds_min_i64", block);
        run<7>("mfma + pk_add + pk_fma + 2 ds_min_i64", block);
        run<8>("mfma + 4x(pk_add, pk_fma)", block);
        run<9>("mfma + 4x(cmp_e64, cndmask_e64)", block);
        run<10>("mfma + 8 v_min_f32", block);
        run<11>("[10 mfma ; 80 fma] grouped (per 10 slots)/10", block);
        run<21>("[10 mfma ; 16 fma] grouped /10", block);
        run<22>("[10 mfma ; 24 fma] grouped /10", block);
        run<23>("[10 mfma ; 40 fma] grouped /10", block);
        run<24>("10 mfma, 24 fma interleaved /10", block);
        run<25>("[10 mfma ; 24 fma ; 16 ds_min_i64] grouped /10", block);
        run<26>("10 mfma, 24 fma, 16 ds_min_i64 interleaved /10", block);
    }
    return 0;
}
