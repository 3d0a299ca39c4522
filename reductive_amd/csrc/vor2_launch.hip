// vor2_launch.hip -- k_encode_vor2 and its launch geometry.
#include "vor2_launch.h"
#include "kernels_vor2.hip.h"

#include <algorithm>

namespace pqhip {

bool launch_vor2(const Vor2Launch& l, hipStream_t st)
{
    // subquantizers per workgroup: as many as keep the group's tables, centroids and norms within 64 KB (two workgroups per CU)
    const size_t per_m = ((size_t)l.max_region_words + 4 + (size_t)l.K * 4) * 4;
    if (per_m > 150 * 1024) return false;
    int mg = (int)std::max<size_t>(1, (64 * 1024) / per_m);
    mg = std::min(std::min(mg, l.M), 8);
    const int n_groups = (l.M + mg - 1) / mg;
    mg = (l.M + n_groups - 1) / n_groups;                          // even groups
    const size_t lds = per_m * (size_t)mg;
    Vor2Args a;
    a.x = l.x; a.n = l.n; a.x_rs = l.x_rs; a.out = l.out; a.o_rs = l.o_rs; a.cb = l.cb; a.cc = l.cc; a.tab = l.tab; a.off = l.off;
    a.M = l.M; a.K = l.K; a.k_pad = l.k_pad; a.dsub = l.dsub; a.mg = mg;
    // rows per thread: the tables are staged once per workgroup, so as many as leave about four rounds of workgroups per group
    // (rows per thread 4 / 8 / 16 / 32 / 64 at the shape above: 0.835 / 0.786 / 0.775 / 0.786 / 0.790 ms)
    const int64_t slots = (int64_t)l.n_cus * std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / (int64_t)std::max<size_t>(lds, 1))) * 4;
    a.rows_per_thread = (int)std::max<int64_t>(1, std::min<int64_t>(64, l.n / (256 * slots)));
    const int64_t rows_per_wg = 256ll * a.rows_per_thread;
    const dim3 grid((unsigned)((l.n + rows_per_wg - 1) / rows_per_wg), (unsigned)n_groups);
#define PQHIP_VOR2(MGT)                                                                                                  \
    do {                                                                                                                 \
        if (lds > 48 * 1024 &&                                                                                           \
            hipFuncSetAttribute((const void*)k_encode_vor2<MGT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return false;                                                                                                \
        hipLaunchKernelGGL(k_encode_vor2<MGT>, grid, dim3(256), lds, st, a);                                             \
    } while (0)
    if (mg <= 2) PQHIP_VOR2(2);
    else if (mg <= 4) PQHIP_VOR2(4);
    else PQHIP_VOR2(8);
#undef PQHIP_VOR2
    return true;
}

}  // namespace pqhip
