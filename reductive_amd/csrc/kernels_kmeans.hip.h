// kernels_kmeans.hip.h -- the k-means step of PQ/OPQ training on the device ("next" row,
// SURVEY.md section 8f rank 1): update_centroids (kmeans.rs:166-198) and mean_squared_error
// (kmeans.rs:329-360) for all M subquantizers at once, bit-identical to the reference's
// sequential f32 arithmetic.  The assignment step is the encode kernel itself.
// Include from exactly one translation unit (pqhip_train.hip).
//
// update_centroids adds the instances of a cluster IN ROW ORDER, so the sum of every (cluster,
// dimension) is one sequential chain.  The chains are independent of one another: a stable
// partition of the row ids by code (histogram per row block, prefix, ranked scatter -- all
// integer work, deterministic) lines every cluster's rows up in ascending order, and then one
// lane per (m, k, e) walks its segment with plain rounded adds.  M*K*dsub lanes (76,800 for the
// headline shape) keep the device busy; each x element is read exactly once.
#pragma once
#include "common.hip.h"

namespace pqhip {

// U1  counts[m][b][k] = number of rows of block b whose code for subquantizer m is k.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_km_hist(const IdxT* __restrict__ codes, int64_t n,
                                                 int64_t c_rs, int K, int rows_per_block, int nb,
                                                 unsigned* __restrict__ counts)
{
    extern __shared__ unsigned km_h[];  // [K]
    const int b = blockIdx.x, m = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += 256) km_h[k] = 0u;
    __syncthreads();
    const int64_t row0 = (int64_t)b * rows_per_block;
    const int64_t rend = (row0 + rows_per_block < n) ? row0 + rows_per_block : n;
    for (int64_t r = row0 + threadIdx.x; r < rend; r += 256) {
        const unsigned c = (unsigned)codes[r * c_rs + m];
        if (c < (unsigned)K) atomicAdd(&km_h[c], 1u);
    }
    __syncthreads();
    unsigned* dst = counts + ((int64_t)m * nb + b) * K;
    for (int k = threadIdx.x; k < K; k += 256) dst[k] = km_h[k];
}

// U2  per subquantizer: counts[m][b][k] -> exclusive prefix over b; seg[m][k] = first position of
// cluster k in perm[m][.] (exclusive prefix of the cluster sizes over k), seg[m][K] = n.
__global__ __launch_bounds__(256) void k_km_scan(unsigned* __restrict__ counts, int K, int nb,
                                                 unsigned* __restrict__ seg)
{
    extern __shared__ unsigned km_t[];  // [K] cluster sizes, then [256] partial sums
    unsigned* part = km_t + K;
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += 256) {
        unsigned run = 0;
        unsigned* p = counts + (int64_t)m * nb * K + k;
#pragma unroll 8
        for (int b = 0; b < nb; ++b) {
            const unsigned c = p[(int64_t)b * K];
            p[(int64_t)b * K] = run;
            run += c;
        }
        km_t[k] = run;
    }
    __syncthreads();
    // exclusive scan of km_t[0..K): every thread owns one contiguous run of ceil(K/256) entries
    const int per = (K + 255) / 256;
    const int k0 = threadIdx.x * per;
    unsigned local = 0;
    for (int i = 0; i < per; ++i)
        if (k0 + i < K) local += km_t[k0 + i];
    part[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int t = 0; t < 256; ++t) { const unsigned v = part[t]; part[t] = run; run += v; }
        seg[(int64_t)m * (K + 1) + K] = run;
    }
    __syncthreads();
    unsigned run = part[threadIdx.x];
    for (int i = 0; i < per; ++i)
        if (k0 + i < K) { seg[(int64_t)m * (K + 1) + k0 + i] = run; run += km_t[k0 + i]; }
}

// U3  stable scatter: perm[m][seg[m][k] + (rank of row r among the rows with code k)] = r.
// One wave per (row block, subquantizer) walks its rows 64 at a time; the rank inside a 64-row
// step comes from the mask of lanes holding the same code (one ballot per code bit).
template <typename IdxT>
__global__ __launch_bounds__(64) void k_km_scatter(const IdxT* __restrict__ codes, int64_t n,
                                                   int64_t c_rs, int K, int rows_per_block, int nb,
                                                   const unsigned* __restrict__ base,
                                                   const unsigned* __restrict__ seg,
                                                   unsigned* __restrict__ perm, int64_t n_pad)
{
    extern __shared__ unsigned km_run[];  // [K] next free position of every cluster
    const int b = blockIdx.x, m = blockIdx.y, lane = threadIdx.x;
    const unsigned* bs = base + ((int64_t)m * nb + b) * K;
    const unsigned* sg = seg + (int64_t)m * (K + 1);
    for (int k = lane; k < K; k += 64) km_run[k] = sg[k] + bs[k];
    __syncthreads();
    int nbits = 0;
    while ((1ll << nbits) < K) ++nbits;
    const int64_t row0 = (int64_t)b * rows_per_block;
    const int64_t rend = (row0 + rows_per_block < n) ? row0 + rows_per_block : n;
    unsigned* pm = perm + (int64_t)m * n_pad;
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned c_next = (row0 + lane < rend) ? (unsigned)codes[(row0 + lane) * c_rs + m] : 0u;
    for (int64_t r0 = row0; r0 < rend; r0 += 64) {
        const int64_t r = r0 + lane;
        const bool valid = r < rend;
        unsigned c = c_next;
        // the next step's codes are requested before this step's ranks are worked out
        c_next = (r + 64 < rend) ? (unsigned)codes[(r + 64) * c_rs + m] : 0u;
        if (c >= (unsigned)K) c = 0u;  // cannot happen for codes written by the encode kernels
        uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (c >> bit) & 1u;
            const uint64_t bm = __builtin_amdgcn_ballot_w64(valid && one);
            peers &= one ? bm : ~bm;
        }
        const int rank = __builtin_popcountll(peers & below);
        const int cnt = __builtin_popcountll(peers);
        unsigned pos = 0;
        if (valid) pos = km_run[c] + (unsigned)rank;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): every lane has its base before any update
        if (valid) {
            pm[pos] = (unsigned)r;
            if (rank == cnt - 1) km_run[c] = pos + 1u;  // highest lane of the group moves the cursor
        }
    }
}

// U4  one lane per (m, k, e): sequential row-order sum of its cluster's instances, then the mean.
// Counts are f32 in the reference (they stop growing at 2^24); division is IEEE.
// The rows arrive in WINDOWS (consecutive row ranges, each with its own partition): the running
// sums are carried from window to window in `acc`, the cluster sizes in tot_in -> tot_out, and the
// last window divides.  (Windows exist so that this latency-bound walk can run beside the
// MFMA-bound assignment kernel of the following rows on a second stream.)
__global__ __launch_bounds__(256) void k_km_segsum(const float* __restrict__ x, int64_t x_rs,
                                                   int64_t x_ms, int64_t n_pad, const unsigned* __restrict__ perm,
                                                   const unsigned* __restrict__ seg, int M, int K,
                                                   int dsub, float* __restrict__ acc,
                                                   const unsigned* __restrict__ tot_in,
                                                   unsigned* __restrict__ tot_out, int first, int last)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)M * K * dsub) return;
    const int mk = (int)(gid / dsub);
    const int e = (int)(gid - (int64_t)mk * dsub);
    const int m = mk / K, k = mk - m * K;
    const unsigned beg = seg[(int64_t)m * (K + 1) + k], end = seg[(int64_t)m * (K + 1) + k + 1];
    const unsigned* pm = perm + (int64_t)m * n_pad;
    const char* xc = reinterpret_cast<const char*>(x + (int64_t)m * x_ms + e);
    const unsigned rsb = (unsigned)x_rs * 4u;  // row stride in bytes (< 2^32): one 32x32->64 mad per address
    auto at = [&](unsigned row) { return *reinterpret_cast<const float*>(xc + (uint64_t)row * rsb); };
    float s = first ? 0.f : acc[gid];
    unsigned i = beg;
    constexpr int U = 16;  // loads in flight per lane
    if (i + U <= end) {
        unsigned r[U];
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = pm[i + j];
        for (; i + 2 * U <= end; i += U) {
            float v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = at(r[j]);
#pragma unroll
            for (int j = 0; j < U; ++j) r[j] = pm[i + U + j];  // row ids of the next step
#pragma unroll
            for (int j = 0; j < U; ++j) s = fadd(s, v[j]);
        }
        float v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = at(r[j]);
#pragma unroll
        for (int j = 0; j < U; ++j) s = fadd(s, v[j]);
        i += U;
    }
    for (; i < end; ++i) s = fadd(s, at(pm[i]));
    const unsigned cnt = (first ? 0u : tot_in[mk]) + (end - beg);
    if (last && cnt) s = __fdiv_rn(s, (float)(cnt < (1u << 24) ? cnt : (1u << 24)));
    acc[gid] = s;
    if (e == 0) tot_out[mk] = cnt;
}

// U4w  the same walk with one WAVE per cluster (m, k).  A wave cannot keep more than about 16
// vector-memory instructions in flight, so what bounds the lane-per-chain form above is
// latency / (rows per instruction): there every instruction fetches ONE row of each of its
// clusters.  Here the 64 lanes fetch RPL = 64 / q consecutive rows of ONE cluster per instruction
// (q lanes cover a sub-vector: dsub/4 lanes with 16-byte loads, or dsub lanes with 4-byte loads),
// PF instructions deep; the rows are staged through a wave-private LDS slab and lanes 0..dsub-1
// then add them to their dimension's chain in row order.  RPL * PF (96 for dsub = 20) rows per
// memory round trip instead of 16, which also makes the walk insensitive to unbalanced clusters.
template <bool VEC>
__global__ __launch_bounds__(256) void k_km_segsum_w(const float* __restrict__ x, int64_t x_rs,
                                                     int64_t x_ms, int64_t n_pad,
                                                     const unsigned* __restrict__ perm,
                                                     const unsigned* __restrict__ seg, int M, int K,
                                                     int dsub, float* __restrict__ acc,
                                                     const unsigned* __restrict__ tot_in,
                                                     unsigned* __restrict__ tot_out, int first, int last)
{
    constexpr int PF = 8;
    extern __shared__ __attribute__((aligned(16))) float km_slab[];  // [4 waves][PF][RPL][dsub]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mk = blockIdx.x * 4 + wave;
    if (mk >= M * K) return;  // (no workgroup barrier below)
    const int m = mk / K, k = mk - m * K;
    const int q = VEC ? (dsub >> 2) : dsub;  // lanes per row
    const int RPL = 64 / q;                  // rows per load instruction
    const int j = lane / q, c = lane - j * q;
    const bool active = j < RPL;
    float* buf = km_slab + (size_t)wave * PF * RPL * dsub;
    const unsigned beg = seg[(int64_t)m * (K + 1) + k], end = seg[(int64_t)m * (K + 1) + k + 1];
    const unsigned* pm = perm + (int64_t)m * n_pad;
    const char* xc = reinterpret_cast<const char*>(x + (int64_t)m * x_ms + (VEC ? 4 * c : c));
    const unsigned rsb = (unsigned)x_rs * 4u;  // row stride in bytes (< 2^32)
    const int slot_off = j * dsub + (VEC ? 4 * c : c);  // this lane's place inside one step of the slab
    float s = 0.f;
    if (!first && lane < dsub) s = acc[(int64_t)mk * dsub + lane];

    // Two batches in flight: while the PF * RPL rows of batch b are added (in row order, from the wave's LDS slab), the
    // rows of batch b + 1 are on their way from HBM and the row ids of batch b + 2 from the permutation -- the walk of
    // the largest cluster of a window is what the launch waits for, and with one batch in flight every batch paid a
    // whole memory round trip in front of its adds (round 3: 0.54-0.73 ms per 512 k-row window, the k-means iteration
    // waited for the update stream, not for the assignment).
    const unsigned per_iter = (unsigned)(PF * RPL);
    auto ids = [&](unsigned (&r)[PF], unsigned i0) {
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const unsigned idx = i0 + t * RPL + j;
            r[t] = (active && idx < end) ? pm[idx] : 0u;     // (row 0 is a valid address; never added)
        }
    };
    auto fetch = [&](const unsigned (&r)[PF], f32x4 (&v4)[PF], float (&v1)[PF]) {
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const char* p = xc + (uint64_t)r[t] * rsb;
            if (VEC) v4[t] = *reinterpret_cast<const f32x4*>(p);
            else v1[t] = *reinterpret_cast<const float*>(p);
        }
    };
    unsigned r[PF];
    f32x4 c4[PF], n4[PF];
    float c1[PF], n1[PF];
    ids(r, beg);
    fetch(r, c4, c1);                       // batch 0
    ids(r, beg + per_iter);                 // ids of batch 1
    for (unsigned i = beg; i < end; i += per_iter) {
        fetch(r, n4, n1);                   // batch b + 1 (clamped ids past the end: loads of row 0, never used)
        ids(r, i + 2 * per_iter);           // ids of batch b + 2
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            if (active) {
                float* d = buf + t * RPL * dsub + slot_off;
                if (VEC) *reinterpret_cast<f32x4*>(d) = c4[t];
                else *d = c1[t];
            }
        }
        const unsigned left = end - i;
        const int cnt = left < per_iter ? (int)left : (int)per_iter;
        if (lane < dsub) {
            const float* col = buf + lane;
#pragma unroll 8
            for (int jj = 0; jj < cnt; ++jj) s = fadd(s, col[jj * dsub]);
        }
#pragma unroll
        for (int t = 0; t < PF; ++t) { c4[t] = n4[t]; c1[t] = n1[t]; }
    }
    const unsigned cntk = (first ? 0u : tot_in[mk]) + (end - beg);
    if (lane < dsub) {
        if (last && cntk) s = __fdiv_rn(s, (float)(cntk < (1u << 24) ? cntk : (1u << 24)));
        acc[(int64_t)mk * dsub + lane] = s;
    }
    if (lane == 0) tot_out[mk] = cntk;
}

// U5  mean_squared_error: ONE sequential f32 fold over all n*dsub squared errors of a
// subquantizer, in row-major order.  It cannot be split, so a workgroup per subquantizer pipelines
// it: waves 1..3 produce the squared errors (c - x)^2 into a double-buffered LDS ring, wave 0
// folds them in order (every lane redundantly, the values arrive as LDS broadcasts).  About 4.5
// cycles per element: it is only run for the iteration whose loss is asked for.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_km_loss(const float* __restrict__ x, int64_t x_rs, int64_t n,
                                                 const IdxT* __restrict__ codes, int64_t c_rs,
                                                 const float* __restrict__ cb, int K, int dsub,
                                                 float inv_len_den, float* __restrict__ loss)
{
    constexpr int PL = 192;        // producer lanes
    constexpr int PE = 8;          // elements per producer lane per chunk
    constexpr int CH = PL * PE;    // 1536 elements per chunk
    __shared__ __attribute__((aligned(16))) float sq[2][CH];
    const int m = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t total = n * dsub;
    const int64_t nchunks = (total + CH - 1) / CH;
    const float* cbm = cb + (int64_t)m * K * dsub;
    const float* xm = x + (int64_t)m * dsub;

    // producer state: element index idx = chunk * CH + j * PL + pl  <->  (row i, column e)
    const int pl = (int)threadIdx.x - 64;
    int64_t i = 0;
    int e = 0;
    const int q_step = PL / dsub, r_step = PL % dsub;
    if (wave != 0) { i = pl / dsub; e = pl - (int)i * dsub; }
    auto produce = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            float v = 0.f;
            if (i < n) {
                unsigned c = (unsigned)codes[i * c_rs + m];
                if (c >= (unsigned)K) c = 0u;
                const float err = fsub(cbm[(int64_t)c * dsub + e], xm[i * x_rs + e]);
                v = fmul(err, err);
            }
            sq[buf][j * PL + pl] = v;
            i += q_step;
            e += r_step;
            if (e >= dsub) { e -= dsub; ++i; }
        }
    };

    if (wave != 0) produce(0);
    __syncthreads();
    float s = 0.f;
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (wave != 0) {
            if (ch + 1 < nchunks) produce(buf ^ 1);
        } else {
            const int64_t left = total - ch * CH;
            const int cnt = left < CH ? (int)left : CH;
            const f32x4* q = reinterpret_cast<const f32x4*>(&sq[buf][0]);
            int t = 0;
#pragma unroll 4
            for (; t + 4 <= cnt; t += 4) {
                const f32x4 v = q[t >> 2];
                s = fadd(fadd(fadd(fadd(s, v[0]), v[1]), v[2]), v[3]);
            }
            for (; t < cnt; ++t) s = fadd(s, sq[buf][t]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[m] = __fdiv_rn(s, inv_len_den);
}


// (C = A^T . B of the OPQ training step: kernels_atb.hip.h)

}  // namespace pqhip
