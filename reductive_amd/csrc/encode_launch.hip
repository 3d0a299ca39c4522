// encode_launch.hip -- compiled once per (PQ_KIND, PQ_T) by the Makefile.
#include "encode_launch.h"
#include <cstdlib>
#include "kernels_mfma_lds.hip.h"

#ifndef PQ_KIND
#error "PQ_KIND and PQ_T must be defined"
#endif

namespace pqhip {

// PQHIP_DEBUG_LDS_PAD=<bytes>: extra dynamic LDS per workgroup (occupancy experiments only)
static unsigned debug_lds_pad()
{
    static const unsigned v = [] { const char* e = getenv("PQHIP_DEBUG_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    return v;
}

template <int KIND, int T, int DP>
static void launch_vec(bool vec, const EncodeArgs& a, dim3 grid, hipStream_t st)
{
    const unsigned pad = debug_lds_pad();
    if (KIND == 0) {
        if (vec) hipLaunchKernelGGL((k_encode_mfma<T, DP, true, uint8_t>), grid, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((k_encode_mfma<T, DP, false, uint8_t>), grid, dim3(256), pad, st, a);
    } else if (KIND == 1) {
        if (vec) hipLaunchKernelGGL((k_encode_mfma_lds<T, DP, true, uint8_t>), grid, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((k_encode_mfma_lds<T, DP, false, uint8_t>), grid, dim3(256), pad, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, true, uint8_t>), grid, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((k_encode_mfma_lds3<T, DP, false, uint8_t>), grid, dim3(256), pad, st, a);
    }
}

template <int KIND, int T>
bool launch_encode_mfma_t(int DP, bool vec, const EncodeArgs& a, dim3 grid, hipStream_t st)
{
    switch (DP) {
    case 4: launch_vec<KIND, T, 4>(vec, a, grid, st); return true;
    case 8: launch_vec<KIND, T, 8>(vec, a, grid, st); return true;
    case 12: launch_vec<KIND, T, 12>(vec, a, grid, st); return true;
    case 16: launch_vec<KIND, T, 16>(vec, a, grid, st); return true;
    case 20: launch_vec<KIND, T, 20>(vec, a, grid, st); return true;
    case 24: launch_vec<KIND, T, 24>(vec, a, grid, st); return true;
    case 28: launch_vec<KIND, T, 28>(vec, a, grid, st); return true;
    case 32: launch_vec<KIND, T, 32>(vec, a, grid, st); return true;
    default: return false;
    }
}

template bool launch_encode_mfma_t<PQ_KIND, PQ_T>(int, bool, const EncodeArgs&, dim3, hipStream_t);

}  // namespace pqhip
