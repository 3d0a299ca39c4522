// opq_fused2_launch.h -- host-side launcher of the second-generation fused OPQ rotate -> encode kernel
// (kernels_opq_fused2.hip.h).  The kernel is instantiated for the (sub-dimension, centroid tiles, burst structure)
// combinations listed below -- the shapes it is worth dispatching for; anything else keeps the two-kernel path.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_opq_common.hip.h"

namespace pqhip {

// (DP, T, d > 256, odd number of full 32-k bursts, partial last burst, slots of the P block)
#define PQHIP_OPQ_FUSED2_LIST(X) X(20, 8, true, true, true, 64) X(16, 8, false, false, false, 64) X(16, 8, true, false, false, 64) \
                                 X(16, 8, true, false, false, 32)

// slots of the P block for (DP, T, d): 64 when that image fits LDS next to the fragments, else 32 (whole sub-vectors only),
// 0 when neither has an instantiation / fits
int opq_fused2_slots(int DP, int T, int d);
inline bool opq_fused2_has(int DP, int T, int d) { return opq_fused2_slots(DP, T, d) != 0; }
size_t opq_fused2_lds_bytes(int DP, int T, int d, int slots);
// returns a hipError_t as int (0 = launched), -1 when there is no instantiation
int launch_opq_fused2(int DP, int T, const OpqFusedArgs& a, dim3 grid, hipStream_t st);

}  // namespace pqhip
