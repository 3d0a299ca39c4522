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
    // (smaller groups for more workgroups per CU lose: every group is one more pass over the rows' cache lines -- d = 300, M = 150,
    // K = 256 with 64 / 48 / 36 / 24 / 16 KB per workgroup: 16.5 / 19.7 / 28.6 / 38.4 / 53.1 ms per 10 M rows)
    int mg = (int)std::max<size_t>(1, (64 * 1024) / per_m);
    mg = std::min(std::min(mg, l.M), 8);
    const int n_groups = (l.M + mg - 1) / mg;
    mg = (l.M + n_groups - 1) / n_groups;                          // even groups
    const size_t lds = per_m * (size_t)mg;
    const int nt = n_groups >= 8 ? 512 : 256;                      // threads per workgroup (kernels_vor2.hip.h)
    Vor2Args a;
    a.x = l.x; a.n = l.n; a.x_rs = l.x_rs; a.out = l.out; a.o_rs = l.o_rs; a.cb = l.cb; a.cc = l.cc; a.tab = l.tab; a.off = l.off;
    a.M = l.M; a.K = l.K; a.k_pad = l.k_pad; a.dsub = l.dsub; a.mg = mg;
    // rows per thread: the tables are staged once per workgroup, so as many as leave about four rounds of workgroups per group,
    // two with eight groups or more (10 M rows, rows per thread 1 / 2 / 4 / 8 / 19 / 32 / 48 / 64: d = 20, M = 10, K = 128
    // 0.84 / 0.69 / 0.62 / 0.585 / 0.595 / 0.60 / 0.61 / 0.69 ms; d = 300, M = 150, K = 256 22.7 / 18.9 / 17.1 / 15.9 / 14.7 / 14.4 / 14.3 / 14.3 ms)
    // (workgroups per CU: by LDS, and 16 waves at the kernel's ~108 registers)
    const int64_t slots = (int64_t)l.n_cus * std::max<int64_t>(1, std::min<int64_t>(nt == 512 ? 2 : 4, (160 * 1024) / (int64_t)std::max<size_t>(lds, 1))) * (n_groups >= 8 ? 2 : 4);
    a.rows_per_thread = (int)std::max<int64_t>(1, std::min<int64_t>(64, l.n / (nt * slots)));
    if (n_groups >= 8) {
        // many groups: the workgroups of a row block share the rows' lines in the XCD's L2 only while they walk in step, so they are
        // kept short -- about six times the staged bytes in row pieces (512 threads, rows per thread 4 / 8 / 12 / 19 / 32:
        // d = 300, M = 150, K = 256 11.4 / 10.7 / 10.7 / 11.4 / 12.4 ms; K = 64 10.0 / 10.0 / 10.6 / 11.5 / 12.5 ms; one-float sub-vectors,
        // d = 128, K = 256 5.6 / 5.1 / 5.0 / 4.85 / 4.8 ms)
        const int64_t piece = (int64_t)mg * l.dsub * 4;
        const int64_t by_stage = std::max<int64_t>(4, std::min<int64_t>(32, (6 * (int64_t)lds) / (nt * piece)));
        a.rows_per_thread = (int)std::max<int64_t>(1, std::min<int64_t>(a.rows_per_thread, by_stage));
    }
    const int64_t rows_per_wg = (int64_t)nt * a.rows_per_thread;
    // a one-dimensional grid: the kernel maps workgroup ids to (row block, group) so that the groups of a row block share an XCD
    a.n_groups = n_groups;
    a.n_row_blocks = (l.n + rows_per_wg - 1) / rows_per_wg;
    const int64_t n_wg = ((a.n_row_blocks + 7) / 8) * 8 * n_groups;
    if (n_wg > 0x7fffffffll) return false;
    const dim3 grid((unsigned)n_wg);
#define PQHIP_VOR2_NT(MGT, NTT)                                                                                          \
    do {                                                                                                                 \
        if (lds > 48 * 1024 &&                                                                                           \
            hipFuncSetAttribute((const void*)k_encode_vor2<MGT, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return false;                                                                                                \
        hipLaunchKernelGGL((k_encode_vor2<MGT, NTT>), grid, dim3(NTT), lds, st, a);                                      \
    } while (0)
#define PQHIP_VOR2(MGT)                                                                                                  \
    do {                                                                                                                 \
        if (nt == 512) PQHIP_VOR2_NT(MGT, 512);                                                                          \
        else PQHIP_VOR2_NT(MGT, 256);                                                                                    \
    } while (0)
    if (mg <= 2) PQHIP_VOR2(2);
    else if (mg <= 4) PQHIP_VOR2(4);
    else PQHIP_VOR2(8);
#undef PQHIP_VOR2
#undef PQHIP_VOR2_NT
    return true;
}

}  // namespace pqhip
