// vor2_prep.hip -- host-side builder of the candidate tables of kernels_vor2.hip.h (see vor2_prep.h for the argument).
#include "vor2_prep.h"

#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstring>
#include <map>
#include <thread>

namespace pqhip {
namespace {

constexpr double kU = 1.0 / 16777216.0;           // 2^-24
constexpr int kCoarseG = 16;
constexpr size_t kMaxListBytes = 48 * 1024;       // per subquantizer

struct Axis {
    float lo, inv;                                // what the kernel uses
    int G;
    // real interval mapped to cell i, widened
    void cell(int i, double& a, double& b) const
    {
        const double w = 1.0 / (double)inv;
        a = (double)lo + (i - 1.0 / 1024) * w;
        b = (double)lo + (i + 1 + 1.0 / 1024) * w;
    }
    // half h of cell i (the kernel: fl(t - floor(t)) >= 0.5, exact on the computed t), widened
    void half(int i, int h, double& a, double& b) const
    {
        const double w = 1.0 / (double)inv;
        a = (double)lo + (i + 0.5 * h - 1.0 / 1024) * w;
        b = (double)lo + (i + 0.5 * (h + 1) + 1.0 / 1024) * w;
    }
};

inline uint32_t f2u(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// centroids that can be a minimum somewhere in [a0, b0] x [a1, b1], ascending
void candidates(const double* c, const double* cc, int K, double ccmax, double a0, double b0, double a1, double b1,
                std::vector<int>& keep, std::vector<double>& v)
{
    const double px[4] = {a0, a0, b0, b0}, py[4] = {a1, b1, a1, b1};
    const double p2 = std::max(a0 * a0, b0 * b0) + std::max(a1 * a1, b1 * b1);
    const double E2 = 2.0 * (16.0 * kU * (p2 + ccmax) + 7.5e-37);          // 2 E(R)
    // first filter: min_R (D_j - D_i) >= min_R D_j - max_R D_i with D the distance itself -- rectangle to point, farthest corner
    double best_upper = INFINITY;
    for (int j = 0; j < K; ++j) {
        const double f0 = std::max(std::fabs(a0 - c[2 * j]), std::fabs(b0 - c[2 * j]));
        const double f1 = std::max(std::fabs(a1 - c[2 * j + 1]), std::fabs(b1 - c[2 * j + 1]));
        best_upper = std::min(best_upper, f0 * f0 + f1 * f1);
    }
    std::vector<int> surv;
    for (int j = 0; j < K; ++j) {
        const double n0 = std::max(std::max(a0 - c[2 * j], c[2 * j] - b0), 0.0);
        const double n1 = std::max(std::max(a1 - c[2 * j + 1], c[2 * j + 1] - b1), 0.0);
        if (!(n0 * n0 + n1 * n1 > best_upper + E2)) surv.push_back(j);
    }
    // v[j][corner] = D_j(corner) - |corner|^2 (linear in the corner) for the survivors
    v.resize((size_t)4 * K);
    for (int j : surv)
        for (int q = 0; q < 4; ++q) v[(size_t)4 * j + q] = cc[j] - 2.0 * (px[q] * c[2 * j] + py[q] * c[2 * j + 1]);
    // pairwise among the survivors (any witness is a valid one)
    keep.clear();
    for (int j : surv) {
        bool dominated = false;
        for (int i : surv) {
            if (i == j) continue;
            double mn = INFINITY;
            for (int q = 0; q < 4; ++q) mn = std::min(mn, v[(size_t)4 * j + q] - v[(size_t)4 * i + q]);
            if (mn > E2) { dominated = true; break; }
        }
        if (!dominated) keep.push_back(j);
    }
}

// the region of one subquantizer; false: not eligible
static bool build_region(const float* q, int K, int dsub, int G0, int G1, int CG0, int CG1, std::vector<uint32_t>& region)
{
    std::vector<double> c((size_t)2 * K), cc((size_t)K), v;
    std::vector<int> keep;
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY}, ccmax = 0.0;
    for (int j = 0; j < K; ++j) {
        for (int a = 0; a < 2; ++a) {
            const double val = a < dsub ? (double)q[dsub * j + a] : 0.0;                // (1-float sub-vectors: points on the x axis)
            if (!std::isfinite(val) || std::fabs(val) > 1.0995116e12) return false;     // 2^40
            c[2 * j + a] = val;
            lo[a] = std::min(lo[a], val);
            hi[a] = std::max(hi[a], val);
        }
        cc[j] = c[2 * j] * c[2 * j] + c[2 * j + 1] * c[2 * j + 1];
        ccmax = std::max(ccmax, cc[j]);
    }
    Axis fine[2], coarse[2];
    for (int a = 0; a < 2; ++a) {
        double s = hi[a] - lo[a];
        const double mag = std::max(std::fabs(lo[a]), std::fabs(hi[a]));
        s = std::max(s, mag * 9.5367431640625e-7);                                       // 2^-20 of the magnitude
        if (!(s >= 9.094947e-13)) s = 9.094947e-13;                                      // 2^-40
        fine[a].G = a == 0 ? G0 : G1;
        fine[a].lo = (float)(lo[a] - 0.5 * s);
        fine[a].inv = (float)(fine[a].G / (2.0 * s));
        coarse[a].G = a == 0 ? CG0 : CG1;
        coarse[a].lo = (float)(lo[a] - 8.0 * s);
        coarse[a].inv = (float)(coarse[a].G / (17.0 * s));
        if (!std::isfinite(fine[a].inv) || !std::isfinite(coarse[a].inv) || !(fine[a].inv > 0.f) || !(coarse[a].inv > 0.f)) return false;
    }
    // cell tables: 16-bit entries (list offset in words << 3 | words - 1), two per 32-bit word
    const size_t fine_words = ((size_t)G0 * G1 + 1) / 2, coarse_words = ((size_t)CG0 * CG1 + 1) / 2;
    region.assign((size_t)kVor2HeaderWords + fine_words + coarse_words, 0u);
    uint16_t* cells16 = reinterpret_cast<uint16_t*>(region.data() + kVor2HeaderWords);
    std::vector<uint8_t> lists;
    std::vector<uint16_t> sub16;              // entries of the subdivided fine cells, four per cell
    // appends a list; the entry (list offset in words << 4 | words - 1), or 0xffff when it does not fit the format
    std::map<std::vector<int>, uint16_t> seen;    // neighbouring cells often share a list: stored once
    auto emit = [&](const std::vector<int>& l, size_t max_words) -> uint16_t {
        const size_t nwords = (l.size() + 3) / 4;
        if (l.empty() || nwords > max_words) return 0xffff;
        const auto it = seen.find(l);
        if (it != seen.end()) return it->second;
        if (lists.size() / 4 >= 4096 || lists.size() + l.size() > kMaxListBytes) return 0xffff;
        const uint16_t e = (uint16_t)((lists.size() / 4) << 4 | (nwords - 1));
        seen.emplace(l, e);
        for (int j : l) lists.push_back((uint8_t)j);
        while (lists.size() % 4) lists.push_back((uint8_t)l.back());       // whole words: the kernel reads four indices at a time
        return e;
    };
    region[0] = f2u(fine[0].lo); region[1] = f2u(fine[0].inv); region[2] = f2u(fine[1].lo); region[3] = f2u(fine[1].inv);
    region[4] = f2u((float)G0);
    region[16] = f2u((float)G1);
    region[17] = f2u((float)CG1);
    region[5] = f2u(coarse[0].lo); region[6] = f2u(coarse[0].inv); region[7] = f2u(coarse[1].lo); region[8] = f2u(coarse[1].inv);
    region[9] = f2u((float)CG0);
    region[10] = (uint32_t)(kVor2HeaderWords * 2);                               // first 16-bit entry of the fine table
    region[11] = (uint32_t)((kVor2HeaderWords + fine_words) * 2);                // of the coarse one
    region[13] = (uint32_t)G1;
    region[14] = (uint32_t)CG1;
    for (int level = 0; level < 2; ++level) {
        const Axis* ax = level == 0 ? fine : coarse;
        const int g0 = ax[0].G, g1 = ax[1].G;
        const uint32_t base = region[10 + level] - (uint32_t)(kVor2HeaderWords * 2);   // index into cells16
        for (int i0 = 0; i0 < g0; ++i0)
            for (int i1 = 0; i1 < g1; ++i1) {
                double a0, b0, a1, b1;
                ax[0].cell(i0, a0, b0);
                ax[1].cell(i1, a1, b1);
                if (level == 0) {
                    candidates(c.data(), cc.data(), K, ccmax, a0, b0, a1, b1, keep, v);
                } else {
                    // A coarse cell serves only the points the fine grid does not take: its list covers the cell minus the fine
                    // grid's extent (shrunk by 2^-10 of a fine cell: a point that close to the edge may land on either side) --
                    // up to four rectangles; a coarse cell inside the fine grid is never looked up.
                    double f0a, f0b, f1a, f1b, t;
                    fine[0].cell(0, f0a, t); fine[0].cell(G0 - 1, t, f0b);
                    fine[1].cell(0, f1a, t); fine[1].cell(G1 - 1, t, f1b);
                    const double m0 = 2.0 / 1024 / (double)fine[0].inv, m1 = 2.0 / 1024 / (double)fine[1].inv;
                    f0a += m0; f0b -= m0; f1a += m1; f1b -= m1;          // cell() widened them; shrink past the true edge
                    std::vector<int> all, part;
                    auto add = [&](double r0a, double r0b, double r1a, double r1b) {
                        if (!(r0a < r0b) || !(r1a < r1b)) return;
                        candidates(c.data(), cc.data(), K, ccmax, r0a, r0b, r1a, r1b, part, v);
                        all.insert(all.end(), part.begin(), part.end());
                    };
                    add(a0, std::min(b0, f0a), a1, b1);                                          // left of the fine grid
                    add(std::max(a0, f0b), b0, a1, b1);                                          // right
                    if (dsub == 2) {                                                             // (1 float: the second coordinate is exactly 0, inside the fine range)
                        add(std::max(a0, f0a), std::min(b0, f0b), a1, std::min(b1, f1a));        // below
                        add(std::max(a0, f0a), std::min(b0, f0b), std::max(a1, f1b), b1);        // above
                    }
                    std::sort(all.begin(), all.end());
                    all.erase(std::unique(all.begin(), all.end()), all.end());
                    if (all.empty()) all.push_back(0);                                           // unreachable cell
                    keep = all;
                }
                // A list is 1 .. 16 words (64 candidates) at a word offset below 4,096 (fine cells: 15 words -- 15 in the length
                // field marks a subdivided cell); anything else is not eligible.  A fine cell with more than two words of
                // candidates is split into 2 x 2 half cells when that shortens its longest list: the wave walks to the longest
                // list among its 64 rows, so the dense cells set the pace.
                uint16_t entry = 0xffff;
                if (level == 0 && keep.size() > 8 && sub16.size() / 4 < 4096) {
                    std::vector<int> part[4];
                    size_t longest = 0;
                    for (int h = 0; h < 4; ++h) {
                        double s0a, s0b, s1a, s1b;
                        ax[0].half(i0, h >> 1, s0a, s0b);
                        ax[1].half(i1, h & 1, s1a, s1b);
                        candidates(c.data(), cc.data(), K, ccmax, s0a, s0b, s1a, s1b, part[h], v);
                        longest = std::max(longest, part[h].size());
                    }
                    if (longest < keep.size() || keep.size() > 60) {       // (a 16-word list cannot be a plain fine entry)
                        uint16_t se[4];
                        bool ok4 = true;
                        for (int h = 0; h < 4; ++h) { se[h] = emit(part[h], 16); ok4 = ok4 && se[h] != 0xffff; }
                        if (!ok4) return false;
                        entry = (uint16_t)((sub16.size() / 4) << 4 | 15u);
                        for (int h = 0; h < 4; ++h) sub16.push_back(se[h]);
                    }
                }
                if (entry == 0xffff) entry = emit(keep, level == 0 ? 15 : 16);
                if (entry == 0xffff) return false;
                cells16[base + (uint32_t)(i0 * g1 + i1)] = entry;
            }
    }
    // the sub-cell entries behind the two cell tables
    region[15] = (uint32_t)(region.size() * 2);
    {
        while (sub16.size() % 2) sub16.push_back(0);
        const size_t w = region.size();
        region.resize(w + sub16.size() / 2);
        if (!sub16.empty()) std::memcpy(region.data() + w, sub16.data(), sub16.size() * 2);
    }
    region[12] = (uint32_t)(region.size() * 4);
    const size_t w0 = region.size();
    region.resize(w0 + lists.size() / 4);
    std::memcpy(region.data() + w0, lists.data(), lists.size());
    return true;
}

}  // namespace

bool vor2_build(const float* quantizers, int64_t M, int64_t K, int64_t dsub, Vor2Tables& out)
{
    out.words.clear();
    out.region_off.assign(1, 0u);
    out.max_region_words = 0;
    if (K < 1 || K > 256 || M < 1 || dsub < 1 || dsub > 2) return false;
    // cells per axis of the fine grid: finer cells shorten the lists, but the tables of a workgroup's subquantizers share LDS with
    // its occupancy (d = 20, M = 10, 10 M rows on one box, 16-bit cell entries: K = 128: G = 16 0.87 ms, 20 0.79, 24 0.735, 28 0.78,
    // 32 0.74, 40 0.90, 48 0.88; K = 256: G = 24 1.06, 32 1.14, 40 1.01; with 32-bit entries G = 24 took 0.78 and 32 0.87-0.90)
    const int G = K >= 48 ? 24 : 16;
    // 1-float sub-vectors (a codebook per dimension): the same construction with every centroid on the x axis -- the distances of
    // CANON-F32 are bit for bit those of the 2-float formulas with a zero second coordinate (x^2 + 0 and fma(0, 0, fl(x c)) are
    // exact) -- on a grid of 8 K x 1 cells (lists of two to three neighbours)
    const int G0 = dsub == 1 ? (int)std::min<int64_t>(1024, std::max<int64_t>(16, 8 * K)) : G, G1 = dsub == 1 ? 1 : G;
    const int CG0 = kCoarseG, CG1 = dsub == 1 ? 1 : kCoarseG;
    // the subquantizers are independent: a few host threads (M = 150, K = 256: 430 ms on one thread)
    std::vector<std::vector<uint32_t>> regions((size_t)M);
    std::atomic<int64_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&] {
        for (int64_t m = next.fetch_add(1); m < M && ok.load(std::memory_order_relaxed); m = next.fetch_add(1))
            if (!build_region(quantizers + m * K * dsub, (int)K, (int)dsub, G0, G1, CG0, CG1, regions[(size_t)m])) ok.store(false);
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_threads = (int)std::min<int64_t>(std::min<int64_t>(8, hw), (M * K + 1023) / 1024);
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (!ok.load()) return false;
    for (int64_t m = 0; m < M; ++m) {
        out.words.insert(out.words.end(), regions[(size_t)m].begin(), regions[(size_t)m].end());
        out.region_off.push_back((uint32_t)out.words.size());
        out.max_region_words = std::max<uint32_t>(out.max_region_words, (uint32_t)regions[(size_t)m].size());
    }
    return true;
}

}  // namespace pqhip
