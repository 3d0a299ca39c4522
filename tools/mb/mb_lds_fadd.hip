// mb_lds_fadd.hip -- is the LDS unit's ds_add_f32 the IEEE binary32 round-to-nearest-even addition of v_add_f32, bit for bit
// (subnormal operands and results, signed zeros, infinities)?  And does a wave's sequence of ds_add_f32 on one address execute
// in program order?  Prints the number of mismatches over 64 M random operand pairs per class.
// build: hipcc -O3 --offload-arch=gfx950 tools/mb/mb_lds_fadd.hip -o tools/mb/mb_lds_fadd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fadd(const float* a, const float* b, float* out_lds, float* out_valu, int n)
{
    __shared__ float s[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    s[threadIdx.x] = a[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("ds_add_f32 %0, %1" :: "v"((unsigned)(threadIdx.x * 4)), "v"(b[i]) : "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    out_lds[i] = s[threadIdx.x];
    out_valu[i] = __fadd_rn(a[i], b[i]);
}

// one wave, one address: acc = ((((0 + x0) + x1) + x2) ...) through ds_add_f32 issued back to back, against the same chain in a register
__global__ void k_chain(const float* x, int n, float* out)
{
    __shared__ float s[1];
    if (threadIdx.x == 0) s[0] = 0.f;
    __syncthreads();
    float acc = 0.f;
    if (threadIdx.x == 0) {
        for (int i = 0; i < n; ++i) {
            asm volatile("ds_add_f32 %0, %1" :: "v"(0u), "v"(x[i]) : "memory");
            acc = __fadd_rn(acc, x[i]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        out[0] = s[0];
        out[1] = acc;
    }
}

static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

int main()
{
    const int n = 1 << 24;
    std::vector<float> a(n), b(n), ol(n), ov(n);
    float *da, *db, *dl, *dv;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dl, n * 4)); CK(hipMalloc(&dv, n * 4));
    unsigned seed = 12345;
    const char* names[4] = {"any bit pattern", "normal, near exponents (rounding, cancellation)", "subnormal operands / results", "tiny + tiny (results cross 2^-126)"};
    for (int cls = 0; cls < 4; ++cls) {
        for (int i = 0; i < n; ++i) {
            unsigned ua = rnd(seed), ub = rnd(seed);
            if (cls == 1) { ua = (ua & 0x807fffffu) | (100u << 23) | ((ua >> 28 & 7u) << 23); ub = (ub & 0x807fffffu) | (100u << 23) | ((ub >> 28 & 7u) << 23); }
            if (cls == 2) { ua &= 0x807fffffu; ub &= 0x807fffffu; }
            if (cls == 3) { ua = (ua & 0x807fffffu) | ((ua >> 30 & 3u) << 23); ub = (ub & 0x807fffffu) | ((ub >> 30 & 3u) << 23); }
            memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4);
        }
        CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fadd, dim3(n / 256), dim3(256), 0, 0, da, db, dl, dv, n);
        CK(hipMemcpy(ol.data(), dl, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), dv, n * 4, hipMemcpyDeviceToHost));
        long bad = 0, bad_nan = 0;
        int shown = 0;
        for (int i = 0; i < n; ++i) {
            unsigned x, y;
            memcpy(&x, &ol[i], 4); memcpy(&y, &ov[i], 4);
            if (x != y) {
                const bool both_nan = (ol[i] != ol[i]) && (ov[i] != ov[i]);
                if (both_nan) { ++bad_nan; continue; }
                ++bad;
                if (shown++ < 3) { unsigned p, q; memcpy(&p, &a[i], 4); memcpy(&q, &b[i], 4); printf("   %08x + %08x: lds %08x valu %08x\n", p, q, x, y); }
            }
        }
        printf("%-52s mismatches %ld of %d (NaN payload differences: %ld)\n", names[cls], bad, n, bad_nan);
    }
    // program order on one address
    const int m = 100000;
    std::vector<float> x(m);
    for (int i = 0; i < m; ++i) x[i] = (float)((int)(rnd(seed) >> 8) - (1 << 23)) * 1.1920929e-7f * (float)(1 + (rnd(seed) >> 28));
    float *dx, *dout, out[2];
    CK(hipMalloc(&dx, m * 4)); CK(hipMalloc(&dout, 8));
    CK(hipMemcpy(dx, x.data(), m * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dx, m, dout);
    CK(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost));
    unsigned p, q; memcpy(&p, &out[0], 4); memcpy(&q, &out[1], 4);
    printf("chain of %d ds_add_f32 on one address: lds %08x register %08x  %s\n", m, p, q, p == q ? "equal" : "DIFFERENT");
    return 0;
}
