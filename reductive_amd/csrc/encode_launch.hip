// encode_launch.hip -- compiled once per (PQ_KIND, PQ_T, PQ_DPSET) by the Makefile.
// PQ_KIND 0: VALU-argmin kernel, 2: LDS-argmin kernel with LDS-resident fragments, 3: the same epilogue on
// v_mfma_f32_16x16x4_f32 with four waves per SIMD (kernels_mfma16.hip.h; the default where it is instantiated:
// T >= 2, DP = dsub = 0 (mod 4), DPSET 0 only).
// PQ_DPSET 0: padded sub-dimension DP = 0 (mod 4), 1: DP = 2 (mod 4), 2: wide sub-vectors DP in
// {40, 48, 56, 64, 80, 96, 112, 128} (default kernel only).
#include "encode_launch.h"
#include "kernels_mfma_lds.hip.h"
#include "kernels_mfma16.hip.h"

#if !defined(PQ_KIND) || !defined(PQ_T) || !defined(PQ_DPSET)
#error "PQ_KIND, PQ_T and PQ_DPSET must be defined"
#endif

namespace pqhip {

template <int KIND, int T, int DP>
static bool launch_vec(bool vec, int code_bytes, const EncodeArgs& a, dim3 grid, hipStream_t st, unsigned pad)
{
    if constexpr (KIND == 3) {
        if constexpr (T >= 2 && DP % 4 == 0 && DP <= 32) {
            if (!vec || a.rows_per_item > 32 * kMfma16MaxTiles) return false;
            if (code_bytes == 1) hipLaunchKernelGGL((k_encode_mfma16<T, DP, uint8_t>), grid, dim3(256), pad, st, a);
            else if (code_bytes == 4) hipLaunchKernelGGL((k_encode_mfma16<T, DP, uint32_t>), grid, dim3(256), pad, st, a);
            else return false;
            return true;
        } else {
            return false;
        }
    } else {
    if (code_bytes == 8) {   // key mode of grouped codebooks: default kernel, 8 tiles per group
        if constexpr (KIND == 2 && T == 8) {
            if (vec) hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, true, unsigned long long>), grid, dim3(256), pad, st, a);
            else hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, false, unsigned long long>), grid, dim3(256), pad, st, a);
            return true;
        } else {
            return false;
        }
    }
    if (code_bytes == 4) {
        if constexpr (KIND == 2) {
            if (vec) hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, true, uint32_t>), grid, dim3(256), pad, st, a);
            else hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, false, uint32_t>), grid, dim3(256), pad, st, a);
            return true;
        } else {
            return false;
        }
    }
    if (code_bytes != 1) return false;
    if constexpr (KIND == 0) {
        if (vec) hipLaunchKernelGGL((k_encode_mfma<T, DP, true, uint8_t>), grid, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((k_encode_mfma<T, DP, false, uint8_t>), grid, dim3(256), pad, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, true, uint8_t>), grid, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, false, uint8_t>), grid, dim3(256), pad, st, a);
    }
    return true;
    }
}

template <int KIND, int T, int DPSET>
bool launch_encode_mfma_t(int DP, bool vec, int code_bytes, const EncodeArgs& a, dim3 grid, hipStream_t st, unsigned pad)
{
    if constexpr (DPSET == 0) {
        switch (DP) {
        case 4: return launch_vec<KIND, T, 4>(vec, code_bytes, a, grid, st, pad);
        case 8: return launch_vec<KIND, T, 8>(vec, code_bytes, a, grid, st, pad);
        case 12: return launch_vec<KIND, T, 12>(vec, code_bytes, a, grid, st, pad);
        case 16: return launch_vec<KIND, T, 16>(vec, code_bytes, a, grid, st, pad);
        case 20: return launch_vec<KIND, T, 20>(vec, code_bytes, a, grid, st, pad);
        case 24: return launch_vec<KIND, T, 24>(vec, code_bytes, a, grid, st, pad);
        case 28: return launch_vec<KIND, T, 28>(vec, code_bytes, a, grid, st, pad);
        case 32: return launch_vec<KIND, T, 32>(vec, code_bytes, a, grid, st, pad);
        default: return false;
        }
    } else if constexpr (DPSET == 2) {
        if constexpr (KIND == 2) {
            switch (DP) {
            case 40: return launch_vec<KIND, T, 40>(vec, code_bytes, a, grid, st, pad);
            case 48: return launch_vec<KIND, T, 48>(vec, code_bytes, a, grid, st, pad);
            case 56: return launch_vec<KIND, T, 56>(vec, code_bytes, a, grid, st, pad);
            case 64: return launch_vec<KIND, T, 64>(vec, code_bytes, a, grid, st, pad);
            case 80: return launch_vec<KIND, T, 80>(vec, code_bytes, a, grid, st, pad);
            case 96: return launch_vec<KIND, T, 96>(vec, code_bytes, a, grid, st, pad);
            case 112: return launch_vec<KIND, T, 112>(vec, code_bytes, a, grid, st, pad);
            case 128: return launch_vec<KIND, T, 128>(vec, code_bytes, a, grid, st, pad);
            default: return false;
            }
        } else {
            return false;
        }
    } else {
        switch (DP) {
        case 2: return launch_vec<KIND, T, 2>(vec, code_bytes, a, grid, st, pad);
        case 6: return launch_vec<KIND, T, 6>(vec, code_bytes, a, grid, st, pad);
        case 10: return launch_vec<KIND, T, 10>(vec, code_bytes, a, grid, st, pad);
        case 14: return launch_vec<KIND, T, 14>(vec, code_bytes, a, grid, st, pad);
        case 18: return launch_vec<KIND, T, 18>(vec, code_bytes, a, grid, st, pad);
        case 22: return launch_vec<KIND, T, 22>(vec, code_bytes, a, grid, st, pad);
        case 26: return launch_vec<KIND, T, 26>(vec, code_bytes, a, grid, st, pad);
        case 30: return launch_vec<KIND, T, 30>(vec, code_bytes, a, grid, st, pad);
        default: return false;
        }
    }
}

template bool launch_encode_mfma_t<PQ_KIND, PQ_T, PQ_DPSET>(int, bool, int, const EncodeArgs&, dim3, hipStream_t, unsigned);

}  // namespace pqhip
