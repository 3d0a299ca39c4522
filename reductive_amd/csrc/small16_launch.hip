// small16_launch.hip -- instantiations of k_encode_small16 (kernels_small16.hip.h).
#include "small16_launch.h"
#include "kernels_small16.hip.h"

namespace pqhip {

bool launch_small16(int KP, int dsub, const SmallKArgs& a, dim3 grid, size_t lds, hipStream_t st)
{
#define PQHIP_CASE(T, D)                                                                          \
    if (KP == 16 * T && dsub == D) {                                                              \
        if (a.M % (16 * small16_pieces_per_lane(D) / D) == 0) {                                                                \
            if (lds > 48 * 1024 &&                                                                \
                hipFuncSetAttribute((const void*)k_encode_small16<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) \
                return false;                                                                     \
            hipLaunchKernelGGL((k_encode_small16<T, D, true>), grid, dim3(256), lds, st, a);      \
        } else {                                                                                  \
            if (lds > 48 * 1024 &&                                                                \
                hipFuncSetAttribute((const void*)k_encode_small16<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) \
                return false;                                                                     \
            hipLaunchKernelGGL((k_encode_small16<T, D, false>), grid, dim3(256), lds, st, a);     \
        }                                                                                         \
        return true;                                                                              \
    }
    PQHIP_CASE(1, 4) PQHIP_CASE(1, 8) PQHIP_CASE(1, 12) PQHIP_CASE(1, 16) PQHIP_CASE(1, 20) PQHIP_CASE(1, 24) PQHIP_CASE(1, 32)
    PQHIP_CASE(2, 4) PQHIP_CASE(2, 8) PQHIP_CASE(2, 12) PQHIP_CASE(2, 16) PQHIP_CASE(2, 20) PQHIP_CASE(2, 24) PQHIP_CASE(2, 32)
#undef PQHIP_CASE
    return false;
}

}  // namespace pqhip
