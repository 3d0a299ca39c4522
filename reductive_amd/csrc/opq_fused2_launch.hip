// opq_fused2_launch.hip -- the instantiations of k_opq_encode_fused2 (own translation unit: they build beside the others).
#include "opq_fused2_launch.h"
#include "kernels_opq_fused2.hip.h"

namespace pqhip {

size_t opq_fused2_lds_bytes(int DP, int T, int d, int slots)
{
    const size_t ngroups = (size_t)(d + 3) / 4, nm = (size_t)slots / DP, s = DP / 2;
    return ((ngroups + 1) * (size_t)slots * 4 + nm * T * s * 64 + nm * 256) * sizeof(float) + 8 * 64 * sizeof(long long);
}

static bool facts_match(int d, bool splitk, bool odd, bool tail)
{
    return (d > kKC) == splitk && (((d >> 5) & 1) != 0) == odd && ((d & 31) != 0) == tail;
}

int opq_fused2_slots(int DP, int T, int d)
{
    if (d % 4 != 0) return 0;
    for (int ns : {64, 32}) {
        if (ns % DP != 0 && ns != 64) continue;
        if (opq_fused2_lds_bytes(DP, T, d, ns) > 160 * 1024) continue;
#define PQHIP_CASE(D, TT, S, O, TL, NS) if (DP == D && T == TT && ns == NS && facts_match(d, S, O, TL)) return ns;
        PQHIP_OPQ_FUSED2_LIST(PQHIP_CASE)
#undef PQHIP_CASE
    }
    return 0;
}

int launch_opq_fused2(int DP, int T, const OpqFusedArgs& a, dim3 grid, hipStream_t st)
{
    const int ns = opq_fused2_slots(DP, T, a.d);
    if (ns == 0) return -1;
    const size_t lds = opq_fused2_lds_bytes(DP, T, a.d, ns);
#define PQHIP_CASE(D, TT, S, O, TL, NS)                                                                                       \
    if (DP == D && T == TT && ns == NS && facts_match(a.d, S, O, TL)) {                                                       \
        hipError_t e = hipFuncSetAttribute((const void*)k_opq_encode_fused2<D, TT, S, O, TL, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e != hipSuccess) return (int)e;                                                                                   \
        hipLaunchKernelGGL((k_opq_encode_fused2<D, TT, S, O, TL, NS>), grid, dim3(512), lds, st, a);                          \
        return (int)hipGetLastError();                                                                                        \
    }
    PQHIP_OPQ_FUSED2_LIST(PQHIP_CASE)
#undef PQHIP_CASE
    return -1;
}

}  // namespace pqhip
