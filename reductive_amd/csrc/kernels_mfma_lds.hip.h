// kernels_mfma_lds.hip.h -- PQ encode, second generation: MFMA distance chains + LDS-atomic argmin.
//
// Measured on MI355X (tools/mb_mfma_valu.hip, tools/mb_lds_atomic.hip; numbers in DESIGN.md):
// v_mfma_f32_32x32x2_f32 and f32 VALU instructions do NOT overlap on a SIMD -- the f32 matrix
// path shares the vector FP32 datapath (that is why both peaks are 157.3 TFLOP/s).  Every VALU
// instruction of the epilogue therefore costs matrix time (~4.4 cycles each against 64 cycles per
// MFMA).  LDS atomics, in contrast, run beside the MFMAs for free (up to ~2.6 per MFMA per SIMD).
//
// So the argmin moves out of the VALU: per (row, centroid) the VALU only forms the distance
//     d = fl(fl(xx + cc) - 2 dp)           (one packed add per two elements + one fma)
// straight into the high half of a 64-bit key {hi = bits(d), lo = centroid offset}, and a
// no-return `ds_min_i64` folds the key into a per-lane, per-centroid-tile LDS slot while the
// matrix core already runs the next tile's fmaf chains.  For non-negative d the signed 64-bit
// order of the key is exactly the (distance, index) lexicographic order, i.e. "first minimum".
// A tile whose minimum is negative (possible only when a row coincides with a centroid up to
// rounding) or not finite is re-evaluated by the exact VALU / scalar paths, so results stay
// bit-identical to CANON-F32 in every case.
#pragma once
#include "kernels_mfma.hip.h"

#ifndef ENC_ABLATE
#define ENC_ABLATE 0      // timing experiments only (results are wrong): 1 no fold/store, 2 no keys/atomics/fold, 3 no atomics, 4 no x loads,
                          // 5 32-bit atomics (distance bits only), 6 fragments of tile 0 kept in registers (no LDS fragment reads),
                          // 7 the chain as 2 S v_mfma_f32_16x16x4_f32 over four 4-register accumulators (same registers, same flop)
#elif ENC_ABLATE != 0 && !defined(PQHIP_TIMING_ONLY_BUILD)
#error "ENC_ABLATE produces wrong results: only `make TIMING=1` (libpqhip_timing.so, -DPQHIP_TIMING_ONLY_BUILD) may set it"
#endif

#ifndef ENC_HYBRID_OFF
#define ENC_HYBRID_OFF 0    // 1: every distance takes its own LDS atomic also for short sub-vectors (the round-2 form; A/B builds)
#endif

namespace pqhip {

// ---------------------------------------------------------------------------------------------
// K1, third generation: as k_encode_mfma_lds, but the sub-codebook's MFMA fragments live in a
// per-workgroup LDS image instead of 80 resident VGPRs, so THREE waves fit on a SIMD (<= 168
// VGPRs).  All four waves of a workgroup work on the same subquantizer m (different 32-row
// streams); A fragments of the next chain are read from LDS while the VALU forms the current
// tile's keys; ||c||^2 of the next tile is read behind the atomics, during the chain.
// Occupancy experiment on MI355X (same kernel body, 1 vs 2 waves/SIMD): 51.8 % -> 66.4 % of MFMA
// peak -- the loop is latency-bound per wave, so the third wave is worth more than the registers.
// Wide sub-vectors (32 < dsub <= 64, DP in {40, 48, 56, 64}) use the same body with one wave per
// SIMD: their chains are 20..32 MFMAs long, so there is little left to hide.
// ---------------------------------------------------------------------------------------------
template <int T, int DP, bool VEC, typename IdxT>
__global__ __launch_bounds__(256, (DP <= 32 ? 3 : 1)) void k_encode_mfma_lds3(EncodeArgs a)
{
    constexpr int S = DP / 2;
    // distances per step that are reduced lane-locally before the LDS atomic (see the epilogue), by chain length
    constexpr int NL = (ENC_HYBRID_OFF || sizeof(IdxT) == 8) ? 0 : (S == 1 ? 6 : S == 2 ? 5 : S == 3 ? 3 : S == 4 ? 2 : 0);
    constexpr int NA = NL > 0 ? 17 - NL : 16;      // atomics (the first one a plain store) per step
    __shared__ __attribute__((aligned(16))) float afrag_s[T][S][64];
    __shared__ __attribute__((aligned(16))) long long slot_s[4][T + 1][64];   // [T]: the tile fold's target
    __shared__ __attribute__((aligned(16))) float cc_s[T * 32];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;

    // ---- workgroup -> (row group, m); XCD-aware: the M workgroups of one row group share an XCD
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int64_t g_local = q / a.M;
    const int m = (int)(q - g_local * a.M);
    const int64_t group = g_local * 8 + xcd;                  // group of 4 * rows_per_item rows
    const bool wg_active = (g_local < a.chunks_per_xcd) && (group < a.n_chunks);

    constexpr long long kKeyInit = 0x7fffffffffffffffll;
    if (wg_active) {
        const float* fp = a.frags + (int64_t)m * T * S * 64;
        float* dst = &afrag_s[0][0][0];
        for (int i = threadIdx.x; i < T * S * 64; i += 256) dst[i] = fp[i];
        const float* ccm = a.cc + (int64_t)m * T * 32;
        for (int i = threadIdx.x; i < T * 32; i += 256) cc_s[i] = ccm[i];
#pragma unroll
        for (int t = 0; t <= T; ++t) slot_s[wave][t][lane] = kKeyInit;
    }
    __syncthreads();
    const int64_t row_begin = (group * 4 + wave) * a.rows_per_item;
    if (!wg_active || row_begin >= a.n) return;
    int64_t row_end = row_begin + a.rows_per_item;
    if (row_end > a.n) row_end = a.n;
    // IdxT == u64 is the key mode of grouped codebooks (K > 256): m is a virtual subquantizer
    constexpr bool KEYS = sizeof(IdxT) == 8;
    const int m_real = KEYS ? m / a.groups : m;
    const float* xcol = a.x + (int64_t)m_real * a.dsub;
    const bool bad_codebook = a.bad_flag != nullptr && *a.bad_flag != 0;  // wave-uniform

    // x tile.  SPLIT (every float real, DP a multiple of 4): lane (row j, half h) fetches only floats
    // [h DP/2, (h + 1) DP/2) of its row's sub-vector -- half the load traffic of "both halves read the whole
    // sub-vector" -- and one v_permlane32_swap per register pair turns (x[2i], x[2i+1] | x[DP/2+2i], x[DP/2+2i+1])
    // into the MFMA B operands of k-steps i and DP/4 + i (half 0: k = 2s, half 1: k = 2s + 1): no lane-half
    // selects.  ||x||^2 (rule 1) is then summed across the halves, the adds in ndarray's order.
    // Otherwise: lane j reads the DP floats of its row's sub-vector (both halves the same bytes).
    // Rows past the end are clamped to the last row (their result is never stored).
    constexpr bool SPLIT = VEC && (DP % 4 == 0) && DP <= 32;
    constexpr int NV2 = SPLIT ? DP / 4 : DP / 2;            // float pairs a lane holds per tile
    const float* const plast = xcol + (a.n - 1) * a.x_rs + (SPLIT ? h * (DP / 2) : 0);
    const float* prow = xcol + (row_begin + j) * a.x_rs + (SPLIT ? h * (DP / 2) : 0);   // this lane's row of the tile being loaded
    const int64_t tile_step = 32 * a.x_rs;
    auto load_tile = [&](f32x2 (&v2)[NV2], int64_t tile_row0) {
        const int left = (int)((a.n - tile_row0 < 32) ? a.n - tile_row0 : 32);  // wave-uniform
        const float* p = (j < left) ? prow : plast;
        if constexpr (SPLIT) {
            float v[DP / 2];
            load_row_floats<DP / 2, DP / 2>(p, v);
#pragma unroll
            for (int e = 0; e < DP / 2; e += 2) v2[e / 2] = (f32x2){v[e], v[e + 1]};
        } else {
            // VEC: all DP floats are real (dsub == DP).  Otherwise: DP <= 32 -> dsub == DP - 1 and the
            // last one is padding; DP > 32 (wide sub-vectors, DP a multiple of 8) -> run-time dsub < DP
            float v[DP];
            if (VEC || DP <= 32) load_row_floats<VEC ? DP : DP - 1, DP>(p, v);
            else load_row_floats_rt<DP>(p, a.dsub, v);
#pragma unroll
            for (int e = 0; e < DP; e += 2) v2[e / 2] = (f32x2){v[e], v[e + 1]};
        }
    };
    auto prep_tile = [&](const f32x2 (&v2)[NV2], float (&bop)[S], float& xx) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < DP / 4; ++i) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v2[i][0]), __float_as_uint(v2[i][1]), false, false);
                bop[i] = __uint_as_float(r[0]);             // half 0: x[2i],        half 1: x[2i + 1]
                bop[DP / 4 + i] = __uint_as_float(r[1]);    // half 0: x[DP/2 + 2i], half 1: x[DP/2 + 2i + 1]
            }
            // rule 1 across the halves: half h holds the elements e = 2 s + h
            constexpr int C = DP / 8, NT = (DP - 8 * C) / 2;
            float sq[S];
#pragma unroll
            for (int s = 0; s < S; ++s) sq[s] = fmul(bop[s], bop[s]);
            float sum = 0.f;
            auto halves = [](float v, float& e, float& o) {
                const unsigned u = __float_as_uint(v);
                const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                e = __uint_as_float(r[0]);
                o = __uint_as_float(r[1]);
            };
            if (C > 0) {
                float p4[4];                                // p[l], l = 2 i + h
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p4[i] = sq[i];                          // 0 + x == x exactly for x >= +0 or NaN
#pragma unroll
                    for (int c = 1; c < C; ++c) p4[i] = fadd(p4[i], sq[4 * c + i]);
                }
                const float u0 = fadd(p4[0], p4[2]), u1 = fadd(p4[1], p4[3]);   // half 0: p0+p4, p2+p6; half 1: p1+p5, p3+p7
                float e0, o0, e1, o1;
                halves(u0, e0, o0);
                halves(u1, e1, o1);
                sum = fadd(fadd(fadd(e0, o0), e1), o1);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float e, o;
                halves(sq[4 * C + i], e, o);
                sum = fadd(fadd(sum, e), o);
            }
            xx = sum;
        } else {
            // the sub-dimension is a compile-time fact (DP or DP - 1): with a run-time test the compiler
            // if-converts and executes BOTH norm variants for every tile (~100 VALU)
            if (VEC) {
                xx = norm_unrolled_packed<DP>(v2);
            } else if (DP <= 32) {
                float v[DP - 1];
#pragma unroll
                for (int e = 0; e < DP - 1; ++e) v[e] = v2[e / 2][e & 1];
                xx = norm_unrolled_static<DP - 1>(v);
            } else {
                float v[DP];
#pragma unroll
                for (int e = 0; e < DP; ++e) v[e] = v2[e / 2][e & 1];
                xx = norm_unrolled_padded<DP>(v, a.dsub);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) bop[s] = h ? v2[s][1] : v2[s][0];
        }
    };
    auto read_cc = [&](int t, f32x4 (&c)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            c[g] = *reinterpret_cast<const f32x4*>(&cc_s[32 * t + 8 * g + 4 * h]);
    };

    int lo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        lo[r] = (r & 3) + 8 * (r >> 2);
        asm volatile("" : "+v"(lo[r]));
    }

    const int64_t last_tile0 = row_begin + ((row_end - row_begin - 1) / 32) * 32;
    f32x2 vn[NV2];
    float bop[S];
    float xx;
    load_tile(vn, row_begin);
    prep_tile(vn, bop, xx);
    // (the row pointer only ever moves forward; once the last tile has been requested it stays there)
    if (row_begin + 32 <= last_tile0) prow += tile_step;
    load_tile(vn, (row_begin + 32 <= last_tile0) ? row_begin + 32 : last_tile0);

    float a0[ENC_ABLATE == 6 ? S : 1];
    if (ENC_ABLATE == 6) {
#pragma unroll
        for (int s = 0; s < S; ++s) { a0[s] = afrag_s[0][s][lane]; asm volatile("" : "+v"(a0[s])); }
    }
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag_s[0][s][lane], bop[s], acc, 0, 0, 0);
    f32x4 c4[4];
    read_cc(0, c4);

    // ---- end of a row tile.  The T per-tile slots are folded by the LDS unit, not the VALU: each slot is read, gets
    // its tile number OR-ed into the index bits (1 VALU) and goes into slot [T] -- the first one by a plain store, the
    // others by ds_min -- and the final key is read back.  Signed 64-bit order of {bits(d), 32 t + offset} is the
    // (distance, index) order, i.e. the first minimum over all 32 T centroids (for d >= 0; a negative minimum sorts
    // first and sends the row to the exact path).  Nothing is ever re-armed: the first key of every step and of every
    // fold is a store.  (Round 2 measured the LDS unit, not the VALU, as this kernel's busiest resource: without the
    // 128 atomics per row tile it runs 10 % faster, without the fold 6 %.  Moving the fold into the next tile's steps
    // -- one slot per step -- removed the 3.7 k-cycle seam and 5 % of the cycles per tile, and the GPU answered with a
    // 2 % lower clock under its power cap: no gain in ms, so the simple form stays.)
    long long* const fin = &slot_s[wave][T][lane];
    auto finish_tile = [&](long long kf, int64_t trow0, int big) {
        float best = __int_as_float((int)(kf >> 32));
        int bidx = (int)(unsigned)kf;
        const bool neg = best < 0.f;
        bidx += 4 * h;
        // the other half's candidate through v_permlane32_swap (VALU) instead of two ds_bpermute round trips:
        // the LDS unit is this kernel's busiest resource (128 atomics + 80 fragment reads per row tile and wave)
        // (swap(u, u): [0] = the lower half's value, [1] = the upper half's value, in every lane)
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)bidx, (unsigned)bidx, false, false);
        const float od = __uint_as_float(h ? s0[0] : s0[1]);
        const int oi = (int)(h ? s1[0] : s1[1]);
        if (od < best || (od == best && oi < bidx)) bidx = oi;
        const int64_t row = trow0 + j;
        const bool valid = row < a.n;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && (big != 0 || neg));
        const unsigned need = (ENC_ABLATE >= 5) ? 0u : (unsigned)(bal | (bal >> 32));  // rows of this tile that need the exact path
        if (h == 0 && valid && !((need >> j) & 1u)) {
            if (KEYS) {
                const float bd = (od < best) ? od : best;   // finite and >= 0 here
                const unsigned gidx = (unsigned)bidx + 256u * (unsigned)(m - m_real * a.groups);
                reinterpret_cast<unsigned long long*>(a.out)[row * a.o_rs + m] =
                    ((unsigned long long)ord_key(bd) << 32) | (unsigned long long)gidx;
            } else {
                reinterpret_cast<IdxT*>(a.out)[row * a.o_rs + m] = (IdxT)bidx;
            }
        }
        if (need)
            encode_rows_slow_v<IdxT>(a.x, a.x_rs, a.out, a.o_rs, a.cb, a.cc, a.K, a.dsub, a.k_pad,
                                     KEYS ? a.groups : 0, m, trow0, need);
    };

    unsigned long long st_tiles = 0, st_steps = 0, st_seam = 0;
    const unsigned long long st_t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int64_t row0 = row_begin; row0 < row_end; row0 += 32) {
        const unsigned long long st_a = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
        float bop_n[S];
        float xx_n = 0.f;
        const f32x2 xx2 = {xx, xx};

#pragma unroll
        for (int t = 0; t < T; ++t) {
            // LDS queue is drained here for free: the previous chain took >= 640 cycles
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            if (t == T - 1) {
                // operands of the next x tile are formed only now, when `bop` is dead (its last
                // chain was issued one step ago): no register copies at the loop seam; the tile
                // after next starts its trip from HBM right away
                prep_tile(vn, bop_n, xx_n);
                if (row0 + 64 <= last_tile0) prow += tile_step;
                if (ENC_ABLATE != 4) load_tile(vn, (row0 + 64 <= last_tile0) ? row0 + 64 : last_tile0);
            }
            float an[S];  // A fragments of the NEXT chain: in flight while the VALU works below
#pragma unroll
            for (int s = 0; s < S; ++s) an[s] = (ENC_ABLATE == 6) ? a0[s] : afrag_s[(t + 1) % T][s][lane];
            __builtin_amdgcn_sched_barrier(0);
            // ---- VALU: 16 distances -> 16 keys ----
            long long key[16];
            float dl[NL > 0 ? NL : 1];
#pragma unroll
            for (int g = 0; g < 4 && ENC_ABLATE != 2; ++g) {
                const f32x2 c01 = {c4[g][0], c4[g][1]}, c23 = {c4[g][2], c4[g][3]};
                f32x2 t01, t23;
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(t01) : "v"(xx2), "v"(c01));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(t23) : "v"(xx2), "v"(c23));
                const float tt[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int r = 4 * g + qq;
                    const float d = ffma(acc[r], -2.0f, tt[qq]);
                    if (NL > 0 && r >= 16 - NL) dl[r - (16 - NL)] = d;
                    else key[r] = ((long long)__float_as_int(d) << 32) | (long long)(unsigned)lo[r];
                }
                asm volatile("" ::"v"(t01), "v"(t23));
            }
            if constexpr (NL > 0) {
                // short sub-vectors: the chain is 1-4 matrix instructions, so the 16 LDS atomics of a step -- not the
                // matrix core -- set its length (one CU retires a 64-bit ds_min per ~6 cycles for all its waves).  The
                // last NL distances are therefore reduced in the lane (strict < in ascending centroid order = first
                // minimum; 3 vector instructions each) and enter the slot as ONE key: 17 - NL atomics per step.
                float bd = dl[0];
                int bi = lo[16 - NL];
#pragma unroll
                for (int e = 1; e < NL; ++e) {
                    const bool lt = dl[e] < bd;
                    bd = lt ? dl[e] : bd;
                    bi = lt ? lo[16 - NL + e] : bi;
                }
                key[16 - NL] = ((long long)__float_as_int(bd) << 32) | (long long)(unsigned)bi;
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- next chain + this tile's atomics + next tile's norms (queued behind the atomics) ----
            f32x16 nacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                           0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            long long* slot = &slot_s[wave][t][lane];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (ENC_ABLATE == 7) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q4 = (2 * s + u) & 3;
                        f32x4 part = {nacc[4 * q4], nacc[4 * q4 + 1], nacc[4 * q4 + 2], nacc[4 * q4 + 3]};
                        const float bo = (t + 1 < T) ? bop[s] : bop_n[s];
                        part = __builtin_amdgcn_mfma_f32_16x16x4f32(u ? bo : an[s], u ? an[s] : bo, part, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) nacc[4 * q4 + e] = part[e];
                    }
                } else
                nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(an[s], (t + 1 < T) ? bop[s] : bop_n[s],
                                                           nacc, 0, 0, 0);
#pragma unroll
                for (int r = (NA * s) / S; r < (NA * (s + 1)) / S; ++r) {
                    if (ENC_ABLATE == 2) continue;
                    if (ENC_ABLATE == 3) { asm volatile("" ::"v"(key[r])); continue; }
                    if (ENC_ABLATE == 5) {
                        asm volatile("" ::"v"(key[r]));
                        (void)__hip_atomic_fetch_min(reinterpret_cast<int*>(slot) + 1, (int)(key[r] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        continue;
                    }
                    // the step's first key is stored (a ds_write instead of a read-modify-write, and the slot needs
                    // no re-arming after the previous row tile), the other 15 are min-ed into it
                    if (r == 0) __hip_atomic_store(slot, key[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    else (void)__hip_atomic_fetch_min(slot, key[r], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            read_cc((t + 1) % T, c4);
            __builtin_amdgcn_sched_barrier(0);
            acc = nacc;
        }

        const unsigned long long st_b = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
        // fold + code byte (see finish_tile above); the chain of the next row tile is already on the matrix core
        if (ENC_ABLATE == 1 || ENC_ABLATE == 2) {
            asm volatile("" ::"v"(acc));
#pragma unroll
            for (int s = 0; s < S; ++s) bop[s] = bop_n[s];
            xx = xx_n;
            continue;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long k = __hip_atomic_load(&slot_s[wave][t][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (t > 0) k |= (long long)(32 * t);
            if (t == 0) __hip_atomic_store(fin, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else (void)__hip_atomic_fetch_min(fin, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        finish_tile(__hip_atomic_load(fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT), row0,
                    (bad_codebook || !(xx < kBigNorm)) ? 1 : 0);
#pragma unroll
        for (int s = 0; s < S; ++s) bop[s] = bop_n[s];
        xx = xx_n;
        if (a.stamps) { const unsigned long long st_c = __builtin_amdgcn_s_memtime(); st_tiles += 1; st_steps += st_b - st_a; st_seam += st_c - st_b; }
    }
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 5;
        o[0] = st_tiles; o[1] = st_steps; o[2] = st_seam;
        o[3] = __builtin_amdgcn_s_memtime() - st_t0; o[4] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

}  // namespace pqhip
