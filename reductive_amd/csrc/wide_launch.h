// wide_launch.h -- host-side launcher of k_encode_mfma_wide (sub-vectors of 129 .. 256 floats: T in {1, 2, 4} x DP in
// {144, 160, .., 256}) and of k_encode_mfma_wide2 (257 .. 1,024 floats: DP in {320, 384, .., 1024}, T = 2 up to 512, 1 beyond);
// the instantiations live in their own translation unit, wide_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_mfma.hip.h"

namespace pqhip {
// false: no instantiation for (T, DP)
bool launch_encode_wide(int T, int DP, const EncodeArgs& a, const float* xx, dim3 grid, hipStream_t st);
// xx[n][M] = rule-1 squared norms of the sub-vectors (the wide kernel's pre-pass)
void launch_row_norms(const float* x, int64_t n, int64_t x_rs, int M, int dsub, float* xx, hipStream_t st);
}  // namespace pqhip
