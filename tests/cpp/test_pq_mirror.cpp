// C++ host-mirror test: reads like the reference's own unit tests (src/pq/pq.rs:409-490).
// Build: g++ -std=c++17 -I include tests/cpp/test_pq_mirror.cpp -L reductive_amd -lpqhip
// Exit code 0 = all checks passed; 77 = no GPU (panic/shape checks still ran).
#include <cstdio>
#include <memory>
#include <limits>
#include "reductive_amd/pq.hpp"

using namespace reductive_amd;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
template <typename F> static bool panics(F f) { try { f(); } catch (const Panic&) { return true; } return false; }

static Pq test_pq()
{   // pq.rs:400-407
    return Pq(std::nullopt, {1, 0, 0, 0, 1, 0, 1, -1, 0, 0, 1, 0}, 2, 2, 3);
}

int main()
{
    const std::vector<float> vectors = {0, 2, 0, -0.5f, 0, 0, 1, -0.2f, 0, 0.5f, 0.5f, 0,
                                        -0.2f, 0.2f, 0, 0, -2, 0, 1, 0.2f, 0, 0, -2, 0};  // pq.rs:378-385
    const std::vector<uint64_t> quant = {1, 1, 0, 1, 1, 0, 0, 0};                          // pq.rs:387-389
    const std::vector<float> recon = {0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0,
                                      0, 1, 0, 1, -1, 0, 1, 0, 0, 1, -1, 0};              // pq.rs:391-398
    Pq pq = test_pq();
    CHECK(pq.quantized_len() == 2 && pq.reconstructed_len() == 6);                         // pq.rs:463-469
    // host-side panics (no device needed)
    CHECK(panics([&] { Pq(std::nullopt, {}, 0, 2, 3); }));
    CHECK(panics([&] { Pq(std::vector<float>(25, 0.f), std::vector<float>(12, 0.f), 2, 2, 3); }));
    CHECK(panics([&] { std::vector<float> x(10); pq.quantize_batch<uint8_t>(View2<const float>(x.data(), 2, 5)); }));
    // single-vector path (host), pq.rs:419-429 and 480-490
    for (int i = 0; i < 4; ++i) {
        auto c = pq.quantize_vector<uint64_t>(vectors.data() + 6 * i, 6);
        CHECK(c[0] == quant[2 * i] && c[1] == quant[2 * i + 1]);
        auto r = pq.reconstruct<uint64_t>(quant.data() + 2 * i, 2);
        for (int k = 0; k < 6; ++k) CHECK(r[k] == recon[6 * i + k]);
    }
    {   // pq.rs:452-461: K = 257 does not fit u8
        Pq wide(std::nullopt, std::vector<float>(257 * 10, 0.5f), 1, 257, 10);
        std::vector<float> x(10, 0.25f);
        CHECK(panics([&] { wide.quantize_vector<uint8_t>(x.data(), 10); }));
    }
    // batch path: needs the GPU
    int32_t ndev = 0;
    if (pqhip_device_count(&ndev) != PQHIP_OK || ndev == 0) {
        bool threw = false;
        try { pq.quantize_batch<uint8_t>(View2<const float>(vectors.data(), 4, 6)); } catch (const HipError& e) { threw = e.status == PQHIP_ENODEV; }
        CHECK(threw);   // fails loudly, no CPU fallback
        std::printf("no GPU: host checks passed\n");
        return 77;
    }
    auto codes = pq.quantize_batch<uint64_t>(View2<const float>(vectors.data(), 4, 6));   // pq.rs:409-417
    for (int i = 0; i < 8; ++i) CHECK(codes[i] == quant[i]);
    auto codes8 = pq.quantize_batch<uint8_t>(View2<const float>(vectors.data(), 4, 6));
    for (int i = 0; i < 8; ++i) CHECK(codes8[i] == quant[i]);
    auto rec = pq.reconstruct_batch<uint64_t>(View2<const uint64_t>(quant.data(), 4, 2));  // pq.rs:471-478
    for (int i = 0; i < 24; ++i) CHECK(rec[i] == recon[i]);
    CHECK(panics([&] { std::vector<uint8_t> bad = {0, 2}; pq.reconstruct_batch<uint8_t>(View2<const uint8_t>(bad.data(), 1, 2)); }));
    {   // kmeans.rs:401-434 (correct_update_centroids): started from the expected means the KAT's
        // assignments are the nearest-centroid assignments, so one iteration must reproduce them
        Context ctx;
        std::vector<float> c = {0.5f, 0.5f, 0, -1.5f, -1, 0, 0, 0, 1.5f};
        const std::vector<float> want = c;
        const std::vector<float> inst = {-1, -1, 0, 1, 1, 0, -2, -1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 2};
        auto loss = kmeans_iterations(ctx, c, 1, 3, 3, View2<const float>(inst.data(), 6, 3), 1);
        for (int i = 0; i < 9; ++i) CHECK(c[i] == want[i]);
        CHECK(loss.size() == 1 && loss[0] == 2.0f / 18.0f);   // squared errors .25 .5 .25 .5 .25 .25 (exact), 18 elements
        CHECK(panics([&] { std::vector<float> q(6); kmeans_iterations(ctx, q, 1, 2, 3, View2<const float>(inst.data(), 9, 2), 1); }));
    }
    {   // OPQ training step on resident instances: identity projection, centroids at their fixed point
        Context ctx;
        const std::vector<float> inst = {-1, -1, 0, 1, 1, 0, -2, -1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 2};
        ResidentMatrix rm(ctx, View2<const float>(inst.data(), 6, 3));
        CHECK(rm.rows() == 6 && rm.device_ptr() != nullptr);
        std::vector<float> c = {0.5f, 0.5f, 0, -1.5f, -1, 0, 0, 0, 1.5f};
        const std::vector<float> want = c;
        const std::vector<float> eye = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        auto cross = opq_train_step(rm, c, 1, 3, 3, eye);
        for (int i = 0; i < 9; ++i) CHECK(c[i] == want[i]);
        // cross = X^T . reconstructed with reconstructed = centroid of every row: exact small sums
        // rows -> centroids: (-1.5,-1,0) (0.5,0.5,0) (-1.5,-1,0) (0.5,0.5,0) (0,0,1.5) (0,0,1.5)
        const float expect[9] = {-1 * -1.5f + 1 * 0.5f + -2 * -1.5f, -1 * -1.0f + 1 * 0.5f + -2 * -1.0f, 0,
                                 -1 * -1.5f + 1 * 0.5f + -1 * -1.5f, -1 * -1.0f + 1 * 0.5f + -1 * -1.0f, 0,
                                 0, 0, 1 * 1.5f + 2 * 1.5f};
        for (int i = 0; i < 9; ++i) CHECK(cross[i] == expect[i]);
        CHECK(panics([&] { std::vector<float> bad(4); opq_train_step(rm, c, 1, 3, 3, bad); }));
    }
    std::printf("all checks passed (GPU)\n");
    return 0;
}
