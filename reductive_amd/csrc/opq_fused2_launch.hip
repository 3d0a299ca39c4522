// opq_fused2_launch.hip -- the instantiations of k_opq_encode_fused2 (own translation unit: they build beside the others).
#include "opq_fused2_launch.h"
#include "kernels_opq_fused2.hip.h"

namespace pqhip {

size_t opq_fused2_lds_bytes(int DP, int T, int d)
{
    const size_t ngroups = (size_t)(d + 3) / 4, nm = 64 / DP, s = DP / 2;
    return ((ngroups + 1) * 256 + nm * T * s * 64 + nm * 256) * sizeof(float) + 8 * 64 * sizeof(long long);
}

static bool facts_match(int d, bool splitk, bool odd, bool tail)
{
    return (d > kKC) == splitk && (((d >> 5) & 1) != 0) == odd && ((d & 31) != 0) == tail;
}

bool opq_fused2_has(int DP, int T, int d)
{
    if (d % 4 != 0 || opq_fused2_lds_bytes(DP, T, d) > 160 * 1024) return false;
#define PQHIP_CASE(D, TT, S, O, TL) if (DP == D && T == TT && facts_match(d, S, O, TL)) return true;
    PQHIP_OPQ_FUSED2_LIST(PQHIP_CASE)
#undef PQHIP_CASE
    return false;
}

int launch_opq_fused2(int DP, int T, const OpqFusedArgs& a, dim3 grid, hipStream_t st)
{
    const size_t lds = opq_fused2_lds_bytes(DP, T, a.d);
#define PQHIP_CASE(D, TT, S, O, TL)                                                                                           \
    if (DP == D && T == TT && facts_match(a.d, S, O, TL)) {                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)k_opq_encode_fused2<D, TT, S, O, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e != hipSuccess) return (int)e;                                                                                   \
        hipLaunchKernelGGL((k_opq_encode_fused2<D, TT, S, O, TL>), grid, dim3(512), lds, st, a);                              \
        return (int)hipGetLastError();                                                                                        \
    }
    PQHIP_OPQ_FUSED2_LIST(PQHIP_CASE)
#undef PQHIP_CASE
    return -1;
}

}  // namespace pqhip
