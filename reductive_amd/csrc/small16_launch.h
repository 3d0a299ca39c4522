// small16_launch.h -- host-side launcher of k_encode_small16 (K <= 32, sub-vectors of 4 / 8 / 12 / 16 / 20 / 24 / 32 floats, 16-byte aligned rows);
// the instantiations live in their own translation unit, small16_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_smallk.hip.h"

namespace pqhip {
constexpr int kSmall16TilesMax = 64;   // tiles per wave: one bit each in the kernel's exact-path mask
constexpr bool small16_has(int KP, int dsub)
{
    return (KP == 16 || KP == 32) && (dsub == 4 || dsub == 8 || dsub == 12 || dsub == 16 || dsub == 20 || dsub == 24 || dsub == 32);
}
// 16-byte pieces a lane holds per row block and stage -- whole sub-vectors -- and rows of a wave's tile (kernels_small16.hip.h):
// 4 / 8 floats: 2 pieces, 64 rows; 12: 3, 32; 16: 4, 32; 20: 5, 32; 24: 6, 16; 32: 8, 16
constexpr int small16_pieces_per_lane(int dsub) { return dsub <= 8 ? 2 : dsub / 4; }
constexpr int small16_tile_rows(int dsub) { return dsub <= 8 ? 64 : dsub <= 20 ? 32 : 16; }
// dynamic LDS of a workgroup: the transposed codebook image and the centroid norms
inline size_t small16_lds_bytes(int M, int dsub, int KP) { return ((size_t)M * dsub * KP + (size_t)M * KP) * sizeof(float); }
// false: no instantiation for (KP, dsub)
bool launch_small16(int KP, int dsub, const SmallKArgs& a, dim3 grid, size_t lds, hipStream_t st);
}  // namespace pqhip
