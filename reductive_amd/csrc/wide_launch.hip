// wide_launch.hip -- instantiations of k_encode_mfma_wide (kernels_mfma_wide.hip.h).
#include "wide_launch.h"
#include <algorithm>
#include "kernels_mfma_wide.hip.h"

namespace pqhip {

template <int T, int DP>
static bool launch_one(const EncodeArgs& a, const float* xx, dim3 grid, hipStream_t st)
{
    const size_t lds = ((size_t)T * (DP / 2) * 64 + (size_t)T * 32) * sizeof(float);
    if (hipFuncSetAttribute((const void*)k_encode_mfma_wide<T, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    hipLaunchKernelGGL((k_encode_mfma_wide<T, DP>), grid, dim3(256), lds, st, a, xx);
    return true;
}

template <int T>
static bool launch_t(int DP, const EncodeArgs& a, const float* xx, dim3 grid, hipStream_t st)
{
    switch (DP) {
    case 144: return launch_one<T, 144>(a, xx, grid, st);
    case 160: return launch_one<T, 160>(a, xx, grid, st);
    case 176: return launch_one<T, 176>(a, xx, grid, st);
    case 192: return launch_one<T, 192>(a, xx, grid, st);
    case 208: return launch_one<T, 208>(a, xx, grid, st);
    case 224: return launch_one<T, 224>(a, xx, grid, st);
    case 240: return launch_one<T, 240>(a, xx, grid, st);
    case 256: return launch_one<T, 256>(a, xx, grid, st);
    default: return false;
    }
}

template <int T, int DP>
static bool launch_two(const EncodeArgs& a, const float* xx, dim3 grid, hipStream_t st)
{
    const size_t lds = ((size_t)T * (DP / 2) * 64 + (size_t)T * 32) * sizeof(float);
    if (hipFuncSetAttribute((const void*)k_encode_mfma_wide2<T, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    hipLaunchKernelGGL((k_encode_mfma_wide2<T, DP>), grid, dim3(256), lds, st, a, xx);
    return true;
}

bool launch_encode_wide(int T, int DP, const EncodeArgs& a, const float* xx, dim3 grid, hipStream_t st)
{
    if (DP > 256) {   // several rule-2 blocks per dot product: T = 2 up to 512 floats, T = 1 beyond (wide_geometry)
        switch (DP) {
        case 320: return T == 2 ? launch_two<2, 320>(a, xx, grid, st) : launch_two<1, 320>(a, xx, grid, st);
        case 384: return T == 2 ? launch_two<2, 384>(a, xx, grid, st) : launch_two<1, 384>(a, xx, grid, st);
        case 448: return T == 2 ? launch_two<2, 448>(a, xx, grid, st) : launch_two<1, 448>(a, xx, grid, st);
        case 512: return T == 2 ? launch_two<2, 512>(a, xx, grid, st) : launch_two<1, 512>(a, xx, grid, st);
        case 576: return T == 1 && launch_two<1, 576>(a, xx, grid, st);
        case 640: return T == 1 && launch_two<1, 640>(a, xx, grid, st);
        case 704: return T == 1 && launch_two<1, 704>(a, xx, grid, st);
        case 768: return T == 1 && launch_two<1, 768>(a, xx, grid, st);
        case 832: return T == 1 && launch_two<1, 832>(a, xx, grid, st);
        case 896: return T == 1 && launch_two<1, 896>(a, xx, grid, st);
        case 960: return T == 1 && launch_two<1, 960>(a, xx, grid, st);
        case 1024: return T == 1 && launch_two<1, 1024>(a, xx, grid, st);
        default: return false;
        }
    }
    switch (T) {
    case 1: return launch_t<1>(DP, a, xx, grid, st);
    case 2: return launch_t<2>(DP, a, xx, grid, st);
    case 4: return launch_t<4>(DP, a, xx, grid, st);
    default: return false;
    }
}

void launch_row_norms(const float* x, int64_t n, int64_t x_rs, int M, int dsub, float* xx, hipStream_t st)
{
    const unsigned grid = (unsigned)std::min<int64_t>((n * M * 8 + 255) / 256, 256 * 64);   // eight lanes per (row, m)
    hipLaunchKernelGGL(k_row_norms, dim3(grid), dim3(256), 0, st, x, n, x_rs, M, dsub, xx);
}

}  // namespace pqhip
