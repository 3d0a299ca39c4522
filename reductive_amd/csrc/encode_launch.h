// encode_launch.h -- host-side launchers of the MFMA encode kernels.  Each (KIND, T, DPSET) triple
// is instantiated in its own translation unit (encode_launch.hip, compiled 16x by the Makefile) so
// the kernel instantiations build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_mfma.hip.h"

namespace pqhip {
// KIND 0: k_encode_mfma (VALU argmin epilogue); KIND 2: k_encode_mfma_lds3 (LDS-atomic argmin,
// A fragments in LDS, 3 waves/SIMD); KIND 3: k_encode_mfma16 (same epilogue on the 16x16x4 matrix instruction, 4 waves/SIMD;
// T in {2, 4, 8}, DPSET 0, dsub == DP only).  DPSET 0: DP in {4, 8, .., 32}; DPSET 1: DP in {2, 6, .., 30};
// DPSET 2 (KIND 2 only): wide sub-vectors, DP in {40, 48, 56, 64} and, one wave per SIMD with the operands spread over both
// register files, {80, 96, 112, 128}.
// Returns false when (T, DP, code_bytes) has no instantiation: u8 codes for both kinds, u32 codes
// (k-means assignment step / wide index types) for KIND 2 only.
template <int KIND, int T, int DPSET>
bool launch_encode_mfma_t(int DP, bool vec, int code_bytes, const EncodeArgs& a, dim3 grid, hipStream_t st, unsigned lds_pad);

#define PQHIP_DECL_LAUNCH(KIND, T)                                                                              \
    extern template bool launch_encode_mfma_t<KIND, T, 0>(int, bool, int, const EncodeArgs&, dim3, hipStream_t, unsigned); \
    extern template bool launch_encode_mfma_t<KIND, T, 1>(int, bool, int, const EncodeArgs&, dim3, hipStream_t, unsigned);
#define PQHIP_DECL_WIDE(T) \
    extern template bool launch_encode_mfma_t<2, T, 2>(int, bool, int, const EncodeArgs&, dim3, hipStream_t, unsigned);
PQHIP_DECL_WIDE(1) PQHIP_DECL_WIDE(2) PQHIP_DECL_WIDE(4) PQHIP_DECL_WIDE(8)
#undef PQHIP_DECL_WIDE
#define PQHIP_DECL_16(T) \
    extern template bool launch_encode_mfma_t<3, T, 0>(int, bool, int, const EncodeArgs&, dim3, hipStream_t, unsigned);
PQHIP_DECL_16(2) PQHIP_DECL_16(4) PQHIP_DECL_16(8)
#undef PQHIP_DECL_16
PQHIP_DECL_LAUNCH(0, 1) PQHIP_DECL_LAUNCH(0, 2) PQHIP_DECL_LAUNCH(0, 4) PQHIP_DECL_LAUNCH(0, 8)
PQHIP_DECL_LAUNCH(2, 1) PQHIP_DECL_LAUNCH(2, 2) PQHIP_DECL_LAUNCH(2, 4) PQHIP_DECL_LAUNCH(2, 8)
#undef PQHIP_DECL_LAUNCH

// lds_pad: extra dynamic LDS per workgroup (occupancy experiments of diagnostic builds; 0 in the default build)
inline bool launch_encode_mfma(int kind, int T, int DP, bool vec, int code_bytes, const EncodeArgs& a,
                               dim3 grid, hipStream_t st, unsigned lds_pad = 0)
{
#define PQHIP_WIDE(TT) \
    if (kind == 2 && T == TT && DP > 32) return launch_encode_mfma_t<2, TT, 2>(DP, vec, code_bytes, a, grid, st, lds_pad);
    PQHIP_WIDE(1) PQHIP_WIDE(2) PQHIP_WIDE(4) PQHIP_WIDE(8)
#undef PQHIP_WIDE
    if (DP > 32) return false;   // (beyond 128: no matrix-core kernel)
#define PQHIP_16(TT) \
    if (kind == 3 && T == TT) return (DP % 4 == 0) ? launch_encode_mfma_t<3, TT, 0>(DP, vec, code_bytes, a, grid, st, lds_pad) : false;
    PQHIP_16(2) PQHIP_16(4) PQHIP_16(8)
#undef PQHIP_16
#define PQHIP_CASE(KIND, TT)                                                                              \
    if (kind == KIND && T == TT)                                                                          \
        return (DP % 4 == 0) ? launch_encode_mfma_t<KIND, TT, 0>(DP, vec, code_bytes, a, grid, st, lds_pad)        \
                             : launch_encode_mfma_t<KIND, TT, 1>(DP, vec, code_bytes, a, grid, st, lds_pad);
    PQHIP_CASE(0, 1) PQHIP_CASE(0, 2) PQHIP_CASE(0, 4) PQHIP_CASE(0, 8)
    PQHIP_CASE(2, 1) PQHIP_CASE(2, 2) PQHIP_CASE(2, 4) PQHIP_CASE(2, 8)
#undef PQHIP_CASE
    return false;
}
}  // namespace pqhip
