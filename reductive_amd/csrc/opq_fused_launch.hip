// opq_fused_launch.hip -- compiled once per (PQ_DP, PQ_T) by the Makefile (one fused OPQ kernel instantiation
// per translation unit so that they build in parallel).
#include "opq_fused_launch.h"

#if !defined(PQ_DP) || !defined(PQ_T)
#error "PQ_DP and PQ_T must be defined"
#endif

namespace pqhip {

template <int DP, int T>
int launch_opq_fused_t(const OpqFusedArgs& a, dim3 grid, size_t lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute((const void*)k_opq_encode_fused<DP, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_opq_encode_fused<DP, T>), grid, dim3(512), lds, st, a);
    return (int)hipGetLastError();
}

template int launch_opq_fused_t<PQ_DP, PQ_T>(const OpqFusedArgs&, dim3, size_t, hipStream_t);

}  // namespace pqhip
