// smallk_launch.h -- host-side launcher of the small-codebook encode kernel (kernels_smallk.hip.h); one
// translation unit per padded centroid count KP in {16, 32, 64}.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_smallk.hip.h"

namespace pqhip {
// returns false when (KP, dsub) has no instantiation
template <int KP>
bool launch_smallk_t(int dsub, const SmallKArgs& a, dim3 grid, hipStream_t st);
extern template bool launch_smallk_t<16>(int, const SmallKArgs&, dim3, hipStream_t);
extern template bool launch_smallk_t<32>(int, const SmallKArgs&, dim3, hipStream_t);
extern template bool launch_smallk_t<64>(int, const SmallKArgs&, dim3, hipStream_t);

inline bool smallk_has(int dsub)
{
    switch (dsub) {
    case 2: case 4: case 6: case 8: case 10: case 12: case 16: case 20: case 24: case 32: return true;
    default: return false;
    }
}
inline int smallk_kp(int64_t K) { return K <= 16 ? 16 : K <= 32 ? 32 : K <= 64 ? 64 : 0; }

inline bool launch_smallk(int KP, int dsub, const SmallKArgs& a, dim3 grid, hipStream_t st)
{
    if (KP == 16) return launch_smallk_t<16>(dsub, a, grid, st);
    if (KP == 32) return launch_smallk_t<32>(dsub, a, grid, st);
    if (KP == 64) return launch_smallk_t<64>(dsub, a, grid, st);
    return false;
}
}  // namespace pqhip
