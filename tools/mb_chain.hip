// Diagnostic micro-benchmark #3: the encode kernel's step, piece by piece (2 waves/SIMD unless noted):
//  A: 10 dependent MFMAs, distinct A/B registers, first with C = 0 (inline constant)
//  B: A + s_nop 16 (MFMA->VALU hazard) + 24 VALU reading the accumulator (16 v_fma reading acc, 8 v_pk_add)
//  C: B + 16 ds_min_i64 interleaved with the MFMAs
//  D: C + 10 ds_read_b32 (A fragments) + 4 ds_read_b128 + s_waitcnt lgkmcnt(0) placed as in the kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void kern(float* out, unsigned long long* cyc, int iters, const float* src)
{
    __shared__ __attribute__((aligned(16))) unsigned long long slots[8][512];
    __shared__ __attribute__((aligned(16))) float ldsf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ldsf[i] = src[i & 255];
    for (int t = 0; t < 8; ++t) slots[t][threadIdx.x] = ~0ull >> 1;
    __syncthreads();
    f32x16 acc = {0};
    float a[10], b[10];
    for (int i = 0; i < 10; ++i) { a[i] = src[threadIdx.x + i * 7]; b[i] = src[threadIdx.x + i * 13 + 3]; }
    float xx = src[threadIdx.x];
    f2 xx2 = {xx, xx};
    f4 c4[4];
    for (int g = 0; g < 4; ++g) c4[g] = *(const f4*)&ldsf[16 * g + 4 * (threadIdx.x >> 5 & 1)];
    unsigned long long key[16];
    for (int r = 0; r < 16; ++r) key[r] = r;
    float sum = 0.f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int t = it & 7;
        if (KIND >= 3) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            for (int i = 0; i < 10; ++i) a[i] = ldsf[(t * 10 + i) * 64 % 3968 + (threadIdx.x & 63)];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND >= 1) {
            for (int g = 0; g < 4; ++g) {
                f2 c01 = {c4[g][0], c4[g][1]}, c23 = {c4[g][2], c4[g][3]}, t01, t23;
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(t01) : "v"(xx2), "v"(c01));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(t23) : "v"(xx2), "v"(c23));
                float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
                for (int q = 0; q < 4; ++q) {
                    float d = __fmaf_rn(acc[4 * g + q], -2.0f, tt[q]);
                    key[4 * g + q] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(4 * g + q);
                }
                asm volatile("" ::"v"(t01), "v"(t23));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 nacc = {0};
        for (int s = 0; s < 10; ++s) {
            nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], nacc, 0, 0, 0);
            if (KIND >= 2)
                for (int r = (16 * s) / 10; r < (16 * (s + 1)) / 10; ++r)
                    (void)__hip_atomic_fetch_min((long long*)&slots[t][threadIdx.x], (long long)key[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        if (KIND >= 3)
            for (int g = 0; g < 4; ++g) c4[g] = *(const f4*)&ldsf[(t * 32 + 8 * g + 4 * (threadIdx.x >> 5 & 1)) & 1023];
        __builtin_amdgcn_sched_barrier(0);
        acc = nacc;
        if (KIND == 0) sum += acc[0];
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) sum += acc[i] + (float)key[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + (float)slots[3][threadIdx.x ^ 1];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}

// E: the same step for TWO row tiles per wave: 20 MFMAs in two independent chains that share the
// A-fragment and norm reads; 48 VALU, 32 atomics.  Would halve the LDS reads per distance and give
// the matrix pipe two independent chains per wave.
__global__ void kern2(float* out, unsigned long long* cyc, int iters, const float* src)
{
    __shared__ __attribute__((aligned(16))) unsigned long long slots[2][8][768];
    __shared__ __attribute__((aligned(16))) float ldsf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ldsf[i] = src[i & 255];
    for (int t = 0; t < 8; ++t) { slots[0][t][threadIdx.x] = ~0ull >> 1; slots[1][t][threadIdx.x] = ~0ull >> 1; }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    float a[10], b0[10], b1[10];
    for (int i = 0; i < 10; ++i) { a[i] = src[threadIdx.x + i * 7]; b0[i] = src[threadIdx.x + i * 13 + 3]; b1[i] = src[threadIdx.x + i * 11 + 5]; }
    float xx0 = src[threadIdx.x], xx1 = src[threadIdx.x + 9];
    f2 xx20 = {xx0, xx0}, xx21 = {xx1, xx1};
    f4 c4[4];
    for (int g = 0; g < 4; ++g) c4[g] = *(const f4*)&ldsf[16 * g + 4 * (threadIdx.x >> 5 & 1)];
    unsigned long long key0[16], key1[16];
    for (int r = 0; r < 16; ++r) { key0[r] = r; key1[r] = r; }
    float sum = 0.f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int t = it & 7;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < 10; ++i) a[i] = ldsf[(t * 10 + i) * 64 % 3968 + (threadIdx.x & 63)];
        __builtin_amdgcn_sched_barrier(0);
        for (int g = 0; g < 4; ++g) {
            f2 c01 = {c4[g][0], c4[g][1]}, c23 = {c4[g][2], c4[g][3]}, t01, t23, u01, u23;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(t01) : "v"(xx20), "v"(c01));
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(t23) : "v"(xx20), "v"(c23));
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(u01) : "v"(xx21), "v"(c01));
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(u23) : "v"(xx21), "v"(c23));
            float tt[4] = {t01[0], t01[1], t23[0], t23[1]}, uu[4] = {u01[0], u01[1], u23[0], u23[1]};
            for (int q = 0; q < 4; ++q) {
                float d0 = __fmaf_rn(acc0[4 * g + q], -2.0f, tt[q]);
                float d1 = __fmaf_rn(acc1[4 * g + q], -2.0f, uu[q]);
                key0[4 * g + q] = ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned)(4 * g + q);
                key1[4 * g + q] = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned)(4 * g + q);
            }
            asm volatile("" ::"v"(t01), "v"(t23), "v"(u01), "v"(u23));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 n0 = {0}, n1 = {0};
        for (int s = 0; s < 10; ++s) {
            n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], n0, 0, 0, 0);
            n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], n1, 0, 0, 0);
            for (int r = (16 * s) / 10; r < (16 * (s + 1)) / 10; ++r) {
                (void)__hip_atomic_fetch_min((long long*)&slots[0][t][threadIdx.x], (long long)key0[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                (void)__hip_atomic_fetch_min((long long*)&slots[1][t][threadIdx.x], (long long)key1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        for (int g = 0; g < 4; ++g) c4[g] = *(const f4*)&ldsf[(t * 32 + 8 * g + 4 * (threadIdx.x >> 5 & 1)) & 1023];
        __builtin_amdgcn_sched_barrier(0);
        acc0 = n0; acc1 = n1;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) sum += acc0[i] + acc1[i] + (float)key0[i] + (float)key1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + (float)slots[0][3][threadIdx.x ^ 1] + (float)slots[1][2][threadIdx.x ^ 1];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}

void run2(int block, const float* src)
{
    const int grid = 256, iters = 4000;
    float* out; unsigned long long* cyc;
    const int nw = grid * block / 64;
    (void)hipMalloc(&out, sizeof(float) * grid * block);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * nw);
    kern2<<<grid, block>>>(out, cyc, 16, src);
    kern2<<<grid, block>>>(out, cyc, iters, src);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_step_wave = h[nw / 2] / (double)iters;
    const double waves = block / 256.0;
    printf("%-60s waves/SIMD=%.0f  cycles/step/wave=%.1f  -> per SIMD per step %.1f  (MFMA share %.1f%%)\n",
           "E: two row tiles per wave (20 MFMA, 48 VALU, 32 atomics)", waves, per_step_wave, per_step_wave / waves,
           1280.0 / (per_step_wave / waves) * 100);
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int KIND>
void run(const char* name, int block, const float* src)
{
    const int grid = 256, iters = 4000;
    float* out; unsigned long long* cyc;
    const int nw = grid * block / 64;
    (void)hipMalloc(&out, sizeof(float) * grid * block);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * nw);
    kern<KIND><<<grid, block>>>(out, cyc, 16, src);
    kern<KIND><<<grid, block>>>(out, cyc, iters, src);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_step_wave = h[nw / 2] / (double)iters;
    const double waves = block / 256.0;
    printf("%-60s waves/SIMD=%.0f  cycles/step/wave=%.1f  -> per SIMD per step %.1f  (MFMA share %.1f%%)\n", name, waves,
           per_step_wave, per_step_wave / waves, 640.0 / (per_step_wave / waves) * 100);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    float* src; (void)hipMalloc(&src, 8192 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = 0.001f * (i % 977) - 0.4f;
    (void)hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int block : {256, 512, 768}) {
        run<0>("A: 10 dependent MFMA (distinct operands, C=0 first)", block, src);
        run<1>("B: A + 24 VALU epilogue on the accumulator", block, src);
        run<2>("C: B + 16 ds_min_i64 interleaved", block, src);
        run<3>("D: C + A-fragment ds_reads + cc reads + lgkmcnt(0) (kernel step)", block, src);
        if (block <= 512) run2(block, src);
    }
    return 0;
}
