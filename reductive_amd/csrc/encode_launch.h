// encode_launch.h -- host-side launchers of the MFMA encode kernels.  Each (KIND, T) pair is
// instantiated in its own translation unit (encode_launch.hip, compiled 8x by the Makefile) so
// the 128 kernel instantiations build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_mfma.hip.h"

namespace pqhip {
// KIND 0: k_encode_mfma (VALU argmin epilogue); KIND 1: k_encode_mfma_lds (LDS-atomic argmin,
// resident A fragments, 2 waves/SIMD); KIND 2: k_encode_mfma_lds3 (A fragments in LDS, 3 waves/SIMD).
// Returns false when (T, DP, code_bytes) has no instantiation: u8 codes for every kind, u32 codes
// (k-means assignment step / wide index types) for KIND 2 only.
template <int KIND, int T>
bool launch_encode_mfma_t(int DP, bool vec, int code_bytes, const EncodeArgs& a, dim3 grid, hipStream_t st);

#define PQHIP_DECL_LAUNCH(KIND, T) \
    extern template bool launch_encode_mfma_t<KIND, T>(int, bool, int, const EncodeArgs&, dim3, hipStream_t);
PQHIP_DECL_LAUNCH(0, 1) PQHIP_DECL_LAUNCH(0, 2) PQHIP_DECL_LAUNCH(0, 4) PQHIP_DECL_LAUNCH(0, 8)
PQHIP_DECL_LAUNCH(1, 1) PQHIP_DECL_LAUNCH(1, 2) PQHIP_DECL_LAUNCH(1, 4) PQHIP_DECL_LAUNCH(1, 8)
PQHIP_DECL_LAUNCH(2, 1) PQHIP_DECL_LAUNCH(2, 2) PQHIP_DECL_LAUNCH(2, 4) PQHIP_DECL_LAUNCH(2, 8)
#undef PQHIP_DECL_LAUNCH

inline bool launch_encode_mfma(int kind, int T, int DP, bool vec, int code_bytes, const EncodeArgs& a,
                               dim3 grid, hipStream_t st)
{
#define PQHIP_CASE(KIND, TT) \
    if (kind == KIND && T == TT) return launch_encode_mfma_t<KIND, TT>(DP, vec, code_bytes, a, grid, st);
    PQHIP_CASE(0, 1) PQHIP_CASE(0, 2) PQHIP_CASE(0, 4) PQHIP_CASE(0, 8)
    PQHIP_CASE(1, 1) PQHIP_CASE(1, 2) PQHIP_CASE(1, 4) PQHIP_CASE(1, 8)
    PQHIP_CASE(2, 1) PQHIP_CASE(2, 2) PQHIP_CASE(2, 4) PQHIP_CASE(2, 8)
#undef PQHIP_CASE
    return false;
}
}  // namespace pqhip
