// smallk_launch.hip -- compiled once per PQ_KP by the Makefile.
#include "smallk_launch.h"

#ifndef PQ_KP
#error "PQ_KP must be defined"
#endif

namespace pqhip {

template <int KP>
bool launch_smallk_t(int dsub, const SmallKArgs& a, dim3 grid, hipStream_t st)
{
#define PQHIP_CASE(D) case D: hipLaunchKernelGGL((k_encode_smallk<KP, D>), grid, dim3(256), 0, st, a); return true;
    switch (dsub) {
    PQHIP_CASE(2) PQHIP_CASE(4) PQHIP_CASE(6) PQHIP_CASE(8) PQHIP_CASE(10) PQHIP_CASE(12) PQHIP_CASE(16) PQHIP_CASE(20)
    PQHIP_CASE(24) PQHIP_CASE(32)
    default: return false;
    }
#undef PQHIP_CASE
}

template bool launch_smallk_t<PQ_KP>(int, const SmallKArgs&, dim3, hipStream_t);

}  // namespace pqhip
