// kernels_gather.hip.h -- the reconstruct gather (plain and lookup form), the per-row rescale and the index-width
// conversion of 2- / 8-byte codes.  k_scale_rows is a non-template kernel: include from exactly one translation unit (pqhip_opq.hip).
#pragma once
#include "common.hip.h"

namespace pqhip {


// ---------------------------------------------------------------------------------------------
// Reconstruct gather  (primitives.rs:137-147, 169-172):
//   out[row, m*dsub + e] = cb[m][codes[row, m]][e]      -- a pure copy, bit exact.
// HBM-write bound: every lane stores VEC contiguous floats, a workgroup covers RB rows so its
// stores form one contiguous stream of RB*d*4 bytes.  chunk -> (m, e) comes from a small LDS
// table built once per workgroup; (row, chunk) advance incrementally (no per-element division).
// A code >= K (reference: index_axis panic) raises *err and is clamped so no access is OOB.
// ---------------------------------------------------------------------------------------------
// SEL = true is the lookup form ("next" row): output row i is the
// reconstruction of code row sel_rows[i] of a resident [n_codes][M] matrix, times
// sel_scales[sel_rows[i]] when scales are given (one rounded f32 multiply per element); a row index
// outside [0, n_codes) raises *err like an out-of-range code (ndarray `select` panics) and yields row 0.
// G = floats fetched per codebook access inside a 16-byte output chunk: 4 when sub-vectors are made
// of 16-byte groups, 2 or 1 when a chunk spans several sub-vectors (dsub = 2, 6, 10, .. / odd dsub);
// the store is one dword-aligned 16-byte store either way.
// CBL: the whole codebook (<= 48 KB) is copied to LDS once per workgroup and the centroids are gathered from there (round 4: small
// codebooks -- the reference's bench shape d = 128 / K = 16 is 8 KB, its test shape d = 20 / K = 128 10 KB; a 16- or 8-byte gather per
// output chunk through the vector memory path held those shapes at 0.63 / 0.44 of HBM).
template <typename IdxT, int VEC, bool SEL = false, int G = VEC, int NE = 16, bool CBL = false>
__global__ __launch_bounds__(256) void k_reconstruct(const IdxT* __restrict__ codes, int64_t n,
                                                     int64_t c_rs, float* __restrict__ out,
                                                     int64_t o_rs, const float* __restrict__ cb,
                                                     int M, int K, int dsub, int rows_per_block,
                                                     unsigned inv_cpr, int* __restrict__ err,
                                                     const int64_t* __restrict__ sel_rows = nullptr,
                                                     int64_t n_codes = 0,
                                                     const float* __restrict__ sel_scales = nullptr,
                                                     int64_t s_rs = 1 /* floats between the scales of consecutive code rows */)
{
    // LDS: chunk -> (m, e) table, then the codes of the current and of the next row block.
    // The codes are the only operand that comes from HBM with a dependent use (code -> gather ->
    // store); fetching them one whole block ahead into LDS takes that round trip off the chain
    // (measured: without it the kernel drops from 5.3 to 3.8 TB/s as soon as the code matrix no
    // longer fits the 256 MB Infinity Cache, i.e. beyond ~17 M rows at M = 15).
    // NE = code elements a thread prefetches per block (rows_per_block * M <= 256 * NE): 16 in general, 4 when the
    // block's codes fit 1024 elements (M <= 16 at 64 rows per block) -- 12 registers less, which is what lets a
    // fourth (lookup form) / fifth workgroup live on a CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = M * dsub;
    const int cpr = d / VEC;  // chunks per row
    // G == 0 ("unaligned wide", d % 4 == 0 but dsub % 4 != 0, dsub >= 4): ONE table entry per 16-byte chunk; a chunk that lies
    // inside one sub-vector is fetched with one dword-aligned 16-byte load (most of them: 11 of 15 at dsub = 15), a chunk
    // that straddles a sub-vector boundary element by element.  With one access per element (G = 1 / 2) these shapes ran
    // at 0.31-0.36 of HBM (d = 300 with M = 10, 20, 60 ...), half of the aligned shapes.
    constexpr int NG = G == 0 ? 1 : VEC / G;  // table entries per chunk
    const int ntbl = cpr * NG;
    int* tbl = reinterpret_cast<int*>(smem);  // [cpr * NG]: m | (e << 16) of every G-float group
    const int ncode = rows_per_block * M;
    IdxT* cl = reinterpret_cast<IdxT*>(smem + (((size_t)ntbl * 4 + 15) & ~(size_t)15));  // [2][ncode]
    // SEL: per-row scale of the current and of the next block, behind the codes
    float* scl = reinterpret_cast<float*>(smem + (((size_t)ntbl * 4 + 15) & ~(size_t)15) +
                                          (((size_t)2 * ncode * sizeof(IdxT) + 15) & ~(size_t)15));  // [2][rows_per_block]
    if constexpr (CBL) {
        // behind everything else: the codebook image [M][K][dsub]
        float* cbl = reinterpret_cast<float*>(smem + (((size_t)ntbl * 4 + 15) & ~(size_t)15) +
                                              (((size_t)2 * ncode * sizeof(IdxT) + 15) & ~(size_t)15) +
                                              (SEL ? (((size_t)2 * rows_per_block * sizeof(float) + 15) & ~(size_t)15) : 0));
        for (int i = threadIdx.x; i < M * K * dsub; i += blockDim.x) cbl[i] = cb[i];
        cb = cbl;                                              // (read after the first __syncthreads below)
    }
    for (int c = threadIdx.x; c < ntbl; c += blockDim.x) {
        const int f = c * (G == 0 ? VEC : G);
        tbl[c] = (f / dsub) | ((f % dsub) << 16);
    }

    // every workgroup owns one contiguous range of row blocks
    const int64_t nblocks = (n + rows_per_block - 1) / rows_per_block;
    const int64_t per_wg = (nblocks + gridDim.x - 1) / gridDim.x;
    const int64_t blk_begin = (int64_t)blockIdx.x * per_wg;
    const int64_t blk_end = (blk_begin + per_wg < nblocks) ? blk_begin + per_wg : nblocks;
    if (blk_begin >= blk_end) return;

    IdxT pre[NE];
    float pre_scale = 1.0f;
    bool bad = false;
    auto fetch_codes = [&](int64_t blk) {  // element e of the block = (row e / M, m e % M)
        const int64_t row0 = blk * rows_per_block;
        const int rows = (n - row0 < rows_per_block) ? (int)(n - row0) : rows_per_block;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + 256 * i;
            IdxT v = 0;
            if (e < rows * M) {
                const int r = e / M, mm = e - r * M;
                int64_t src = row0 + r;
                if (SEL && sel_rows) {           // (SEL without sel_rows: rows already selected, only the scales apply)
                    src = sel_rows[src];
                    if (src < 0 || src >= n_codes) { bad = true; src = 0; }
                }
                v = codes[src * c_rs + mm];
            }
            pre[i] = v;
        }
        if (SEL && sel_scales && (int)threadIdx.x < rows) {
            int64_t src = row0 + threadIdx.x;
            if (sel_rows) {
                src = sel_rows[src];
                if (src < 0 || src >= n_codes) src = 0;
            }
            pre_scale = sel_scales[src * s_rs];
        }
    };
    auto stash_codes = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + 256 * i;
            if (e < ncode) cl[buf * ncode + e] = pre[i];
        }
        if (SEL && sel_scales && (int)threadIdx.x < rows_per_block) scl[buf * rows_per_block + threadIdx.x] = pre_scale;
    };
    fetch_codes(blk_begin);
    stash_codes(0);
    __syncthreads();

    int cur = 0;
    for (int64_t blk = blk_begin; blk < blk_end; ++blk) {
        const bool more = blk + 1 < blk_end;
        if (more) fetch_codes(blk + 1);  // in flight while this block is gathered and stored
        const int64_t row0 = blk * rows_per_block;
        const int rows = (n - row0 < rows_per_block) ? (int)(n - row0) : rows_per_block;
        const int nchunks = rows * cpr;  // < 2^16 (host guarantees), so L / cpr == umulhi(L, inv) exactly
        const IdxT* cc = cl + cur * ncode;
#pragma unroll 4
        for (int L = threadIdx.x; L < nchunks; L += 256) {
            const int row = (cpr == 1) ? L : (int)__umulhi((unsigned)L, inv_cpr);
            const int c = L - row * cpr;
            float* dst = out + (row0 + row) * o_rs + (int64_t)c * VEC;
            const float sc = (SEL && sel_scales) ? scl[cur * rows_per_block + row] : 1.0f;
            float qv[VEC];
            if constexpr (G == 0) {
                const int me = tbl[c];
                const int m = me & 0xffff, e = me >> 16;
                if (e + VEC <= dsub) {
                    uint64_t code = (uint64_t)cc[row * M + m];
                    if (code >= (uint64_t)K) { bad = true; code = 0; }
                    const f32x4 t = *reinterpret_cast<const f32x4_u*>(cb + ((int64_t)m * K + (int64_t)code) * dsub + e);
                    qv[0] = t[0]; qv[1] = t[1]; qv[2] = t[2]; qv[3] = t[3];
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const int mi = (e + i < dsub) ? m : m + 1;            // dsub >= 4: at most one boundary inside a chunk
                        const int ei = (e + i < dsub) ? e + i : e + i - dsub;
                        uint64_t code = (uint64_t)cc[row * M + mi];
                        if (code >= (uint64_t)K) { bad = true; code = 0; }
                        qv[i] = cb[((int64_t)mi * K + (int64_t)code) * dsub + ei];
                    }
                }
            } else
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int me = tbl[c * NG + gi];
                const int m = me & 0xffff, e = me >> 16;
                uint64_t code = (uint64_t)cc[row * M + m];
                if (code >= (uint64_t)K) { bad = true; code = 0; }
                const float* src = cb + ((int64_t)m * K + (int64_t)code) * dsub + e;
                if (G == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4_u*>(src);
                    qv[0] = t[0]; qv[1] = t[1]; qv[2] = t[2]; qv[3] = t[3];
                } else if (G == 2) {
                    const f32x2 t = *reinterpret_cast<const f32x2_u*>(src);
                    qv[2 * gi] = t[0]; qv[2 * gi + 1] = t[1];
                } else {
                    qv[gi] = *src;
                }
            }
            if (SEL && sel_scales) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) qv[v] = fmul(qv[v], sc);
            }
            if (VEC == 4) {
                // streaming store: keep the L2 for the codebook, not for the 1.2 KB/row output
                const f32x4 q = {qv[0], qv[1], qv[2], qv[3]};
                __builtin_nontemporal_store(q, reinterpret_cast<f32x4_u*>(dst));
            } else {
                dst[0] = qv[0];
            }
        }
        if (more) stash_codes(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (bad) atomicOr(err, 1);
}

// Shapes outside k_reconstruct's per-block budgets (M > 4096 or a row of >= 2^16 chunks): one
// thread per output element.  Same copy, no throughput claim.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_reconstruct_any(const IdxT* __restrict__ codes, int64_t n,
                                                         int64_t c_rs, float* __restrict__ out,
                                                         int64_t o_rs, const float* __restrict__ cb,
                                                         int M, int K, int dsub, int* __restrict__ err,
                                                         const int64_t* __restrict__ sel_rows,
                                                         int64_t n_codes,
                                                         const float* __restrict__ sel_scales, int64_t s_rs)
{
    const int64_t d = (int64_t)M * dsub;
    const int64_t total = n * d;
    bool bad = false;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / d;
        const int64_t c = idx - row * d;
        const int m = (int)(c / dsub), e = (int)(c - (int64_t)m * dsub);
        int64_t src = row;
        if (sel_rows) {
            src = sel_rows[row];
            if (src < 0 || src >= n_codes) { bad = true; src = 0; }
        }
        uint64_t code = (uint64_t)codes[src * c_rs + m];
        if (code >= (uint64_t)K) { bad = true; code = 0; }
        float v = cb[((int64_t)m * K + (int64_t)code) * dsub + e];
        if (sel_scales) v = fmul(v, sel_scales[src * s_rs]);
        out[row * o_rs + c] = v;
    }
    if (bad) atomicOr(err, 1);
}

// out[row][0..d) *= scales[sel_rows[row]]  (lookup form with a projection: the scale follows the un-rotation)
__global__ __launch_bounds__(256) void k_scale_rows(float* __restrict__ out, int64_t n, int d, int64_t o_rs,
                                                    const int64_t* __restrict__ sel_rows, int64_t n_codes,
                                                    const float* __restrict__ sel_scales, int64_t s_rs)
{
    const int64_t total = n * d;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / d;
        const int c = (int)(idx - row * d);
        int64_t src = sel_rows[row];
        if (src < 0 || src >= n_codes) src = 0;
        out[row * o_rs + c] = fmul(out[row * o_rs + c], sel_scales[src * s_rs]);
    }
}

// First pass of a lookup into a LARGE resident matrix (gather_dev: code matrix beyond the Infinity Cache): the selected code
// rows and scales are copied into compact arrays, the reconstruct kernel then runs over those like a plain batch.
// Why two passes (profiles/r4_lookup_counters.json): with 100 M resident rows every lookup is a translation miss in the
// vector L1's TLB (3.3 M UTCL1 misses per 10 M lookups against 7 k at 10 M resident rows; L2 hit rate and DRAM requests
// unchanged), and a miss stalls that L1's in-order pipeline for ALL waves of the CU -- the kernel's own 12 GB store stream
// included: mean L1 -> L2 read latency 227 -> 474 cycles, the launch 2.66 -> 4.2 ms.  In this kernel the misses stall nothing
// but other lookups.  One thread per code element (a row's M elements on consecutive lanes; writes are contiguous).
template <typename IdxT>
__global__ __launch_bounds__(256) void k_select_code_rows(const IdxT* __restrict__ codes, int64_t c_rs, int64_t n_codes,
                                                          const int64_t* __restrict__ sel_rows, int64_t n, int M,
                                                          IdxT* __restrict__ out_codes, const float* __restrict__ sel_scales,
                                                          int64_t s_rs, float* __restrict__ out_scales, int* __restrict__ err)
{
    const int64_t total = n * M;
    bool bad = false;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / M;
        const int m = (int)(idx - row * M);
        int64_t src = sel_rows[row];
        if (src < 0 || src >= n_codes) { bad = true; src = 0; }
        out_codes[idx] = codes[src * c_rs + m];
        if (m == 0 && sel_scales) out_scales[row] = sel_scales[src * s_rs];
    }
    if (bad && err) atomicOr(err, 1);
}

// The same first pass for 1-byte codes with M <= 16: ONE thread per lookup.  A row's M bytes start at any byte address, so the
// thread reads the five aligned dwords that cover them (one 16-byte and one 4-byte load), shifts them into place and writes
// one aligned 16-byte record (compact stride 16).  Against one thread per byte: 15 x fewer memory instructions for the same
// number of translation misses (one per lookup either way).  `matrix_bytes` bounds the reads: the last rows of the matrix fall
// back to byte loads when their five-dword window would leave it.
__global__ __launch_bounds__(256) void k_select_code_rows16(const uint8_t* __restrict__ codes, int64_t c_rs, int64_t n_codes, int64_t matrix_bytes,
                                                            const int64_t* __restrict__ sel_rows, int64_t n, int M,
                                                            uint8_t* __restrict__ out_codes /* [n][16], 16-byte aligned */,
                                                            const float* __restrict__ sel_scales, int64_t s_rs, float* __restrict__ out_scales,
                                                            int* __restrict__ err)
{
    bool bad = false;
    const uintptr_t base0 = reinterpret_cast<uintptr_t>(codes);
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
        int64_t src = sel_rows[row];
        if (src < 0 || src >= n_codes) { bad = true; src = 0; }
        const uintptr_t a = base0 + (uintptr_t)(src * c_rs);
        const uintptr_t al = a & ~(uintptr_t)3;
        const unsigned sh = (unsigned)(a & 3);
        unsigned w[4];
        if (al + 20 <= base0 + (uintptr_t)matrix_bytes) {
            const uint4 q = *reinterpret_cast<const uint4 __attribute__((aligned(4)))*>(al);
            const unsigned q4 = *reinterpret_cast<const unsigned*>(al + 16);
            const unsigned d[5] = {q.x, q.y, q.z, q.w, q4};
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
        } else {
            unsigned char b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = (i < M) ? codes[src * c_rs + i] : (unsigned char)0;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (unsigned)b[4 * i] | ((unsigned)b[4 * i + 1] << 8) | ((unsigned)b[4 * i + 2] << 16) | ((unsigned)b[4 * i + 3] << 24);
        }
        *reinterpret_cast<uint4*>(out_codes + row * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        if (sel_scales) out_scales[row] = sel_scales[src * s_rs];
    }
    if (bad && err) atomicOr(err, 1);
}

// Index-width conversion for the device entry points with 2- and 8-byte codes (the reference is generic over the index
// type I, traits.rs:77-88; the kernels produce / consume u8 and u32): dst[row][m] = (Dst)src[row][m].  `K` > 0: a source
// value >= K raises *err BEFORE it is narrowed (reconstruct: index_axis panic, primitives.rs:146) and is clamped to K.
template <typename Src, typename Dst>
__global__ __launch_bounds__(256) void k_convert_codes(const Src* __restrict__ src, int64_t s_rs, Dst* __restrict__ dst, int64_t d_rs,
                                                       int64_t n, int M, unsigned long long K, int* __restrict__ err)
{
    const int64_t total = n * M;
    bool bad = false;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / M;
        const int m = (int)(idx - row * M);
        unsigned long long v = (unsigned long long)src[row * s_rs + m];
        if (K && v >= K) { bad = true; v = K; }
        dst[row * d_rs + m] = (Dst)v;
    }
    if (bad && err) atomicOr(err, 1);
}

}  // namespace pqhip
