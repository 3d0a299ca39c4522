// vor2_launch.h -- host-side launcher of k_encode_vor2 (2-float sub-vectors, K <= 256, u8 codes); the kernel lives in its own
// translation unit, vor2_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pqhip {
struct Vor2Launch {
    const float* x; int64_t n, x_rs; uint8_t* out; int64_t o_rs;
    const float* cb; const float* cc; const uint32_t* tab; const uint32_t* off;
    int M, K, k_pad, dsub;
    uint32_t max_region_words;
    int n_cus;
};
// false: the tables of one subquantizer do not fit LDS
bool launch_vor2(const Vor2Launch& l, hipStream_t st);
}  // namespace pqhip
