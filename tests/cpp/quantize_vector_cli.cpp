// Single-vector path of the C++ host mirror (include/reductive_amd/pq.hpp: Pq::quantize_vector /
// Pq::reconstruct, the reference's pq.rs:285-298 / 329-343) driven from a file, so that
// tests/test_single_vector.py can compare it with the oracle's pqo_quantize_vector on seeded data.
//   in : int64 M, K, dsub, has_proj, n ; f32 quantizers[M*K*dsub] ; f32 P[d*d] (if has_proj) ; f32 x[n*d]
//   out: int64 codes[n*M] ; f32 reconstructions[n*d] (of those codes)
// No device is touched (host-only methods).
#include <cstdio>
#include <vector>
#include "reductive_amd/pq.hpp"

using namespace reductive_amd;

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    int64_t hdr[5];
    if (std::fread(hdr, sizeof(int64_t), 5, fi) != 5) return 2;
    const int64_t M = hdr[0], K = hdr[1], dsub = hdr[2], has_proj = hdr[3], n = hdr[4], d = M * dsub;
    std::vector<float> q((size_t)(M * K * dsub)), P, x((size_t)(n * d));
    if (std::fread(q.data(), sizeof(float), q.size(), fi) != q.size()) return 2;
    if (has_proj) {
        P.resize((size_t)(d * d));
        if (std::fread(P.data(), sizeof(float), P.size(), fi) != P.size()) return 2;
    }
    if (std::fread(x.data(), sizeof(float), x.size(), fi) != x.size()) return 2;
    std::fclose(fi);
    Pq pq(has_proj ? std::optional<std::vector<float>>(P) : std::nullopt, q, M, K, dsub);
    std::vector<int64_t> codes((size_t)(n * M));
    std::vector<float> rec((size_t)(n * d));
    for (int64_t i = 0; i < n; ++i) {
        auto c = pq.quantize_vector<uint64_t>(x.data() + i * d, d);
        for (int64_t m = 0; m < M; ++m) codes[(size_t)(i * M + m)] = (int64_t)c[(size_t)m];
        auto r = pq.reconstruct<uint64_t>(c.data(), M);
        for (int64_t k = 0; k < d; ++k) rec[(size_t)(i * d + k)] = r[(size_t)k];
    }
    FILE* fo = std::fopen(argv[2], "wb");
    if (!fo) return 2;
    std::fwrite(codes.data(), sizeof(int64_t), codes.size(), fo);
    std::fwrite(rec.data(), sizeof(float), rec.size(), fo);
    std::fclose(fo);
    return 0;
}
