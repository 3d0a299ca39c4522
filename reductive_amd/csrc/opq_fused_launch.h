// opq_fused_launch.h -- host-side launcher of the fused OPQ rotate -> encode kernel
// (kernels_opq_fused.hip.h); one instantiation per sub-dimension, each in its own translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_opq_fused.hip.h"

namespace pqhip {

// returns a hipError_t as int (0 = launched)
template <int DP, int T>
int launch_opq_fused_t(const OpqFusedArgs& a, dim3 grid, size_t lds, hipStream_t st);

// instantiated sub-dimensions (x centroid tiles T in {4, 8}: 97 <= K <= 256)
#define PQHIP_OPQ_FUSED_DPS(X) X(4) X(8) X(10) X(12) X(16) X(20) X(24) X(32)
#define PQHIP_DECL(DP)                                                                                     \
    extern template int launch_opq_fused_t<DP, 4>(const OpqFusedArgs&, dim3, size_t, hipStream_t);         \
    extern template int launch_opq_fused_t<DP, 8>(const OpqFusedArgs&, dim3, size_t, hipStream_t);
PQHIP_OPQ_FUSED_DPS(PQHIP_DECL)
#undef PQHIP_DECL

inline bool opq_fused_has(int DP, int T)
{
    if (T != 4 && T != 8) return false;
#define PQHIP_CASE(D) if (DP == D) return true;
    PQHIP_OPQ_FUSED_DPS(PQHIP_CASE)
#undef PQHIP_CASE
    return false;
}

inline int launch_opq_fused(int DP, int T, const OpqFusedArgs& a, dim3 grid, size_t lds, hipStream_t st)
{
#define PQHIP_CASE(D)                                                   \
    if (DP == D && T == 4) return launch_opq_fused_t<D, 4>(a, grid, lds, st); \
    if (DP == D && T == 8) return launch_opq_fused_t<D, 8>(a, grid, lds, st);
    PQHIP_OPQ_FUSED_DPS(PQHIP_CASE)
#undef PQHIP_CASE
    return -1;
}

// dynamic LDS of k_opq_encode_fused<DP> at dimension d
inline size_t opq_fused_lds_bytes(int DP, int d)
{
    const size_t ngroups = (size_t)(d + 3) / 4;
    const size_t nm = 64 / DP;
    return (ngroups * 256 + 8 * 2 * 32 * 36 + nm * 256) * sizeof(float) + 8 * 64 * sizeof(long long);
}
}  // namespace pqhip
