// vor2_prep.h -- candidate tables for the 2-float sub-vector encode kernel (kernels_vor2.hip.h), built on the host when a
// codebook handle is created.
//
// With dsub = 2 a subquantizer is a set of K points in the plane and `cluster_assignment` (kmeans.rs:149-156 over the
// distances of linalg.rs:173-174) is a nearest-point query.  The reference's own statistical test quantizes d = 20 with M = 10,
// K = 128 (pq.rs:431-440): 1,280 distances per vector, and every kernel that evaluates all of them ends at ~6.5e12
// distances/s (the per-distance epilogue, DESIGN K1).  The grid below lets the kernel evaluate only the centroids that CAN
// win for the cell a point falls into -- 3 to 9 of 128 -- with the SAME float operations as everywhere else, so the code is
// the oracle's as long as the winner is on the cell's list.  That is what the construction guarantees:
//
//   * D_j(p) = |p|^2 + |c_j|^2 - 2 p.c_j is the exact squared distance, d~_j(p) what CANON-F32 computes (rule 1 norms, rule 2
//     dot product, fl(fl(xx + cc) - 2 dp)).  For finite inputs without overflow |d~_j(p) - D_j(p)| <= 8 u (|p|^2 + |c_j|^2),
//     u = 2^-24: 2u |p|^2 and 2u |c|^2 from the two norms, u (xx + cc) from their sum, 2 * u (|p|^2 + |c|^2) from the dot
//     product (one rounded product, one rounded fma, doubled), u (t + 2 |dp|) <= 2u (|p|^2 + |c|^2) from the last operation.
//     E(R) = 16 u (max_R |p|^2 + max_j |c_j|^2) + 2^-120 is used (twice the bound, plus room for results below 2^-126).
//   * For a rectangle R and two centroids, D_j(p) - D_i(p) = |c_j|^2 - |c_i|^2 - 2 p.(c_j - c_i) is LINEAR in p: its minimum
//     over R is at a corner.  If min_R (D_j - D_i) > 2 E(R) for some i, then d~_j(p) >= D_j - E > D_i + E >= d~_i(p) at every
//     float point of R: j is never a minimum there (not even a tied one) and is left off R's list.  Whatever i is, the
//     chain of strict inequalities ends on the list, so the first minimum over the list is the first minimum over all K.
//   * The cell of a point is what the kernel computes, int(fl(fl(x - lo) * inv)) per axis with lo, inv the FLOATS stored in the
//     table; the real numbers mapped to cell i lie in [lo + i / inv, lo + (i + 1) / inv] up to 3 u G cells of rounding: the
//     rectangle is widened by 2^-10 of a cell on every side.
//   * A fine grid (G x G cells over the centroids' bounding box widened by half its size) serves the bulk of the data, a coarse
//     one (16 x 16 over 17 x the box) the outliers; points outside both, NaN / Inf, and codebooks with non-finite or extreme
//     (|c| > 2^40, box below 2^-40) entries take the exact paths that exist already.
// All arithmetic here is double precision on float inputs (relative error 1e-16 against margins of 1e-7).
#pragma once
#include <cstdint>
#include <vector>

namespace pqhip {

// Per subquantizer, 32-bit words: [0] lo0 [1] inv0 [2] lo1 [3] inv1 [4] G0 (cells along axis 0, as float) [16] G1 (float)
// [5..8] the same for the coarse grid [9] CG0 (float) [17] CG1 (float) [10] index of the fine cell table's first 16-bit entry
// [11] of the coarse one's [15] of the sub-cell table's (all counted in 16-bit units from the region's first word) [12] BYTE
// offset of the lists [13] G1 (int: cell = i0 G1 + i1) [14] CG1 (int) [18] [19] unused; then the
// cell tables and the lists.  A cell entry is u16: list offset in words << 4 | words - 1 (lists: u8 centroid indices, ascending;
// every list starts on a word and is padded to whole words with its last index; at most 16 words).  A FINE cell entry whose
// length field is 15 is subdivided: bits 15..4 number a group of four entries in the sub-cell table, one per half cell
// (2 [t0 - floor(t0) >= 0.5] + [t1 - floor(t1) >= 0.5]), each a plain entry (up to 16 words; plain FINE entries: 15).  The list of a coarse cell
// covers the part of the cell outside the fine grid only.  Offsets are relative to the region's first word.
constexpr int kVor2HeaderWords = 20;

struct Vor2Tables {
    std::vector<uint32_t> words;        // all regions, back to back
    std::vector<uint32_t> region_off;   // [M + 1] word offsets into `words`
    uint32_t max_region_words = 0;
};

// false: the codebook is not eligible (non-finite or extreme centroids, K > 256, a list region too large)
bool vor2_build(const float* quantizers, int64_t M, int64_t K, int64_t dsub, Vor2Tables& out);   // dsub 1 or 2

}  // namespace pqhip
