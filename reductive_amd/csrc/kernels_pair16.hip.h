// kernels_pair16.hip.h -- PQ encode for codebooks of K <= 16 centroids with sub-vectors of 2, 4, 8 or 16 floats
// (the reference's own Criterion shape is d = 128, M = 16, K = 16: benches/pq.rs:9-10).  HBM-bound work: 4 d + M bytes per
// vector against 2 K d = 4,096 flop.  (Template kernel: instantiated from pqhip_encode.hip.)
//
// Why another kernel: per (row, subquantizer) there are only 16 distances, so what a kernel pays PER TILE decides.  The
// default MFMA kernel spends a 32-centroid tile (half of it padding) and its whole row-tile seam on every subquantizer
// (4.8e9 vectors/s at the bench shape); the VALU kernel (kernels_smallk.hip.h) needs ~160 vector instructions per
// (64 rows, subquantizer) and reaches 65 % of that issue bound (6.7e9).  This kernel lets ONE 32x32 matrix tile serve TWO
// subquantizers: the A operand is block-diagonal,
//     A[i][k] = c_{2p+hh}[r][k - hh dsub]  for hh dsub <= k < (hh + 1) dsub,   0 elsewhere,      i = (r & 3) + 8 (r >> 2) + 4 hh,
// and the B operand is the 2 dsub contiguous floats of the pair, so accumulator register r of lane (row j, half hh) is the
// dot product of row j's sub-vector 2p + hh with its centroid r -- every lane of the wave holds 16 real distances of ITS
// subquantizer, there is no half-wave merge and no padding centroid.  The zero blocks are exact: fma(0, x, acc) = acc for
// finite x (rows with NaN / Inf take the exact path), and a chain that starts with zero products still starts from +0.
// Per (32 rows, pair): dsub matrix instructions, 8 packed adds + 16 fmas for the distances (rule 3), a lane-local strict-<
// scan (first minimum) -- 72 vector instructions for 1,024 distances where the VALU kernel issues 320 -- and two code bytes.
// x is read once, in whole 128-byte lines, through a wave-private double-buffered LDS slab whose rows hold each pair's
// floats de-interleaved (even k first, odd k second), so that a lane's eight B operands are two ds_read_b128.
#pragma once
#include "kernels_mfma.hip.h"

namespace pqhip {

struct Pair16Args {
    const float* x;       // [n][x_rs]
    int64_t n;
    int64_t x_rs;
    uint8_t* out;         // [n][o_rs]
    int64_t o_rs;
    const float* fragp;   // [NP][DSUB][64]   block-diagonal A fragments (k_build_pair_frags)
    const float* ccp;     // [NP][2][16]      norms by (half, register), +inf for padding
    const float* cb;      // [M][K][dsub]     (exact path)
    const float* cc;      // [M][k_pad]       (exact path)
    int M, K, k_pad, NP;  // NP = ceil(M / 2) pairs
    int64_t n_tiles;      // ceil(n / 32)
};

// (k_build_pair_frags, the kernel that writes fragp / ccp, is a preparation kernel: kernels_prep.hip.h)

template <int DSUB>
__global__ __launch_bounds__(256, 3) void k_encode_pair16(Pair16Args a)
{
    static_assert(DSUB == 2 || DSUB == 4 || DSUB == 8 || DSUB == 16, "sub-vector length");
    constexpr int PF = 2 * DSUB;              // floats of a pair
    constexpr int PPC = 32 / PF;              // pairs per 128-byte chunk of a row
    constexpr int XS = 36;                    // slab row stride in floats (conflict-free ds_read_b128 of 32 rows)
    constexpr int NQ = DSUB / 4 > 0 ? DSUB / 4 : 1;   // 16-byte operand reads per lane and pair (DSUB = 2: one 8-byte read)
    extern __shared__ __attribute__((aligned(16))) float p16_s[];
    float* frag_s = p16_s;                                  // [NP][DSUB][64]
    float* cc_s = frag_s + (size_t)a.NP * DSUB * 64;        // [NP][2][16]
    float* slab_all = cc_s + (size_t)a.NP * 32;             // [4 waves][2][32][XS]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < a.NP * DSUB * 64; i += 256) frag_s[i] = a.fragp[i];
    for (int i = threadIdx.x; i < a.NP * 32; i += 256) cc_s[i] = a.ccp[i];
    __syncthreads();
    float* slab = slab_all + (size_t)wave * 2 * 32 * XS;
    const int d = a.M * DSUB;
    const int nchunk = (d + 31) / 32;
    const int sr = lane >> 3, sc = lane & 7;  // staging role: rows sr + 8 i (i = 0..3), 16-byte piece sc of the chunk

    // the wave's tiles: tile = wave-global index + k * (number of waves)
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
    if (wave_id >= a.n_tiles) return;
    f32x4 st[4];
    auto fetch = [&](int64_t tile, int c) {   // chunk c of tile -> registers (whole 128-byte lines: 8 lanes per row)
        const int64_t r0 = tile * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = (r0 + sr + 8 * i < a.n) ? r0 + sr + 8 * i : a.n - 1;
            const int k = 32 * c + 4 * sc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k + 4 <= d) v = *reinterpret_cast<const f32x4_u*>(a.x + row * a.x_rs + k);
            else if (k < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < d) v[e] = a.x[row * a.x_rs + k + e];
            }
            st[i] = v;
        }
    };
    auto stash = [&](int buf) {               // de-interleave: a pair's even k first, odd k second
        float* sb = slab + (size_t)buf * 32 * XS;
        // piece sc holds floats 4 sc .. 4 sc + 3 of the chunk = local k 4 sc' .. of pair (4 sc) / PF
        const int pair = (4 * sc) / PF, lk = 4 * sc - pair * PF;          // lk is a multiple of 4
        const int ev = pair * PF + lk / 2, od = pair * PF + PF / 2 + lk / 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* rowp = sb + (sr + 8 * i) * XS;
            *reinterpret_cast<f32x2*>(rowp + ev) = (f32x2){st[i][0], st[i][2]};
            *reinterpret_cast<f32x2*>(rowp + od) = (f32x2){st[i][1], st[i][3]};
        }
    };

    int64_t tile = wave_id;
    fetch(tile, 0);
    stash(0);
    {   // chunk 1 (or the next tile's chunk 0) on its way
        const bool more_c = 1 < nchunk;
        const int64_t t1 = more_c ? tile : tile + n_waves;
        if (t1 < a.n_tiles) fetch(t1, more_c ? 1 : 0);
    }
    int buf = 0;
    for (; tile < a.n_tiles; tile += n_waves) {
        const int64_t row0 = tile * 32;
        const int64_t row = row0 + j;
        const bool valid = row < a.n;
        for (int c = 0; c < nchunk; ++c) {
            // the chunk after this one goes to the other slab buffer, the one after that leaves HBM
            {
                const bool last_c = c + 1 == nchunk;
                const int64_t tn = last_c ? tile + n_waves : tile;
                if (tn < a.n_tiles) stash(buf ^ 1);
                const int c2 = last_c ? 1 : c + 2;
                const bool wrap = c2 >= nchunk;
                const int64_t t2 = wrap ? tn + n_waves : tn;
                const int cc2 = wrap ? c2 - nchunk : c2;
                if (tn < a.n_tiles && t2 < a.n_tiles) fetch(t2, cc2 < nchunk ? cc2 : 0);
            }
            const float* sb = slab + (size_t)buf * 32 * XS + j * XS;
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const int p = c * PPC + q;               // pair index
                if (2 * p >= a.M) break;                 // wave-uniform
                // B operands: this half's DSUB floats (even k in half 0, odd k in half 1)
                float bop[DSUB];
                if constexpr (DSUB >= 4) {
#pragma unroll
                    for (int e = 0; e < NQ; ++e) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sb + q * PF + h * DSUB + 4 * e);
                        bop[4 * e] = v[0]; bop[4 * e + 1] = v[1]; bop[4 * e + 2] = v[2]; bop[4 * e + 3] = v[3];
                    }
                } else {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(sb + q * PF + h * DSUB);
                    bop[0] = v[0]; bop[1] = v[1];
                }
                // ||x_m||^2 (rule 1) of THIS half's subquantizer m = 2 p + h.  The lane holds, of both sub-vectors, the
                // elements of its own parity: bop[s] = x_pair[2 s + h]; sub-vector hh covers k-steps hh DSUB/2 .. (hh+1) DSUB/2 - 1.
                float xx;
                {
                    constexpr int HS = DSUB / 2;                 // k-steps (pairs of elements) per sub-vector
                    float sq[DSUB];
#pragma unroll
                    for (int s = 0; s < DSUB; ++s) sq[s] = fmul(bop[s], bop[s]);
                    // partial sums of unrolled_dot restricted to this lane's parity, for sub-vector 0 (a*) and 1 (b*):
                    // element e = 2 s' + h of a sub-vector feeds p[e & 7]; the lane owns e & 7 in {h, 2 + h, 4 + h, 6 + h}
                    auto partial = [&](int base, float (&u)[2], float (&tail)[4], int& ntail) {
                        // DSUB >= 8: u[0] = p[h] + p[4 + h], u[1] = p[2 + h] + p[6 + h] (each p a sum over chunks of 8)
                        // DSUB < 8: no full chunk, elements go to the sequential tail in index order
                        ntail = 0;
                        if constexpr (DSUB >= 8) {
                            float pp[4];
#pragma unroll
                            for (int l = 0; l < 4; ++l) {
                                pp[l] = sq[base + l];
#pragma unroll
                                for (int cch = 1; cch < DSUB / 8; ++cch) pp[l] = fadd(pp[l], sq[base + 4 * cch + l]);
                            }
                            u[0] = fadd(pp[0], pp[2]);           // p[h] + p[4 + h]
                            u[1] = fadd(pp[1], pp[3]);           // p[2 + h] + p[6 + h]
                        } else {
                            u[0] = u[1] = 0.f;
#pragma unroll
                            for (int l = 0; l < HS; ++l) tail[l] = sq[base + l];
                            ntail = HS;
                        }
                    };
                    float ua[2], ub[2], ta[4], tb[4];
                    int na, nb;
                    partial(0, ua, ta, na);
                    partial(HS, ub, tb, nb);
                    // exchange: half 0 needs sub-vector 0's odd-parity partials (held by half 1), half 1 needs
                    // sub-vector 1's even-parity partials (held by half 0).  swap(X = sub-vector 0's, Y = sub-vector 1's):
                    // X.upper <-> Y.lower, so afterwards half 0 holds (X: own sv0 partial, Y: half 1's sv0 partial) and
                    // half 1 holds (X: half 0's sv1 partial, Y: own sv1 partial).
                    auto xchg = [&](float x0, float y1, float& even_part, float& odd_part) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(y1), false, false);
                        even_part = __uint_as_float(r[0]);       // half 0: own sv0 (even parity) | half 1: half 0's sv1 (even parity)
                        odd_part = __uint_as_float(r[1]);        // half 0: half 1's sv0 (odd parity) | half 1: own sv1 (odd parity)
                    };
                    if constexpr (DSUB >= 8) {
                        float e0, o0, e1, o1;
                        xchg(ua[0], ub[0], e0, o0);              // (p0+p4 , p1+p5) of the lane's sub-vector
                        xchg(ua[1], ub[1], e1, o1);              // (p2+p6 , p3+p7)
                        xx = fadd(fadd(fadd(e0, o0), e1), o1);   // 0 + (p0+p4) is exact
                    } else {
                        float s = 0.f;
#pragma unroll
                        for (int l = 0; l < HS; ++l) {
                            float ev, od;
                            xchg(ta[l], tb[l], ev, od);          // elements 2 l and 2 l + 1 of the lane's sub-vector
                            s = fadd(fadd(s, ev), od);
                        }
                        xx = s;
                    }
                }
                // distance chains: DSUB matrix instructions
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float* fp = frag_s + (size_t)p * DSUB * 64 + lane;
#pragma unroll
                for (int s = 0; s < DSUB; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fp[s * 64], bop[s], acc, 0, 0, 0);
                // rule 3 + first minimum, lane-local
                const float* ccl = cc_s + p * 32 + h * 16;
                const f32x2 xx2 = {xx, xx};
                float dv[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(ccl + 4 * g);
                    const f32x2 t01 = pk_add(xx2, (f32x2){c4[0], c4[1]}), t23 = pk_add(xx2, (f32x2){c4[2], c4[3]});
                    const float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[4 * g + e] = ffma(acc[4 * g + e], -2.0f, tt[e]);   // == fl(t - fl(dp + dp)) below kBigNorm
                }
                // first minimum as a tree (depth 4 instead of a 16-step dependent scan): the right operand, which holds the
                // HIGHER centroid indices, replaces the left one only when strictly smaller
                int iv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) iv[r] = r;
#define P16_LEVEL(W)                                                        \
                _Pragma("unroll") for (int r = 0; r < 16; r += 2 * (W)) {          \
                    const bool lt = dv[r + (W)] < dv[r];                            \
                    dv[r] = lt ? dv[r + (W)] : dv[r];                               \
                    iv[r] = lt ? iv[r + (W)] : iv[r];                               \
                }
                P16_LEVEL(1) P16_LEVEL(2) P16_LEVEL(4) P16_LEVEL(8)
#undef P16_LEVEL
                const int bidx = iv[0];
                const int m = 2 * p + h;
                const bool mine = valid && m < a.M;
                // A NaN / Inf in EITHER sub-vector of the pair reaches both chains (0 x NaN through the zero blocks), so a row
                // is sent to the exact path for both subquantizers when either of its two norms is not finite or too large
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && !(xx < kBigNorm));
                const unsigned need = (unsigned)bal | (unsigned)(bal >> 32);
                if (mine && !((need >> j) & 1u)) a.out[row * a.o_rs + m] = (uint8_t)bidx;
                if (need) {
                    encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, 2 * p, row0, need);
                    if (2 * p + 1 < a.M)
                        encode_rows_slow_v<uint8_t>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, DSUB, a.k_pad, 0, 2 * p + 1, row0, need);
                }
            }
            buf ^= 1;
        }
    }
}

}  // namespace pqhip
