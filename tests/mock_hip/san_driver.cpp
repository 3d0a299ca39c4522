// Drives the host-resident entry points of the sanitized, mock-backed build of libpqhip (tests/mock_hip/
// Makefile) with the shapes that stress the host logic: strided and transposed inputs, every index width,
// wide strided outputs, several staging chunks per shard, two device slots, OPQ and K > 256 (scratch
// leases from several host threads), range errors, handle lifetimes.  Nothing is computed (kernel launches
// are no-ops in the mock): the pass criterion is "no AddressSanitizer / UBSan report and the status codes
// the C ABI promises".
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
#include "pqhip.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    int32_t nd = 0;
    CHECK(pqhip_device_count(&nd) == PQHIP_OK && nd == 2);
    pqhip_ctx* ctx = nullptr;
    CHECK(pqhip_ctx_create(nullptr, 0, &ctx) == PQHIP_OK && pqhip_ctx_n_devices(ctx) == 2);
    const int64_t M = 15, K = 256, dsub = 20, d = M * dsub;
    std::vector<float> q((size_t)(M * K * dsub), 0.25f), P((size_t)(d * d), 0.f);
    for (int64_t i = 0; i < d; ++i) P[(size_t)(i * d + i)] = 1.f;
    pqhip_codebook *pq = nullptr, *opq = nullptr, *wide = nullptr;
    CHECK(pqhip_codebook_create(ctx, q.data(), M, K, dsub, nullptr, &pq) == PQHIP_OK);
    CHECK(pqhip_codebook_create(ctx, q.data(), M, K, dsub, P.data(), &opq) == PQHIP_OK);
    std::vector<float> qw((size_t)(3 * 700 * 8), 0.5f);
    CHECK(pqhip_codebook_create(ctx, qw.data(), 3, 700, 8, nullptr, &wide) == PQHIP_OK);
    CHECK(pqhip_codebook_quantized_len(pq) == M && pqhip_codebook_reconstructed_len(opq) == d && pqhip_codebook_has_projection(opq));

    // 500 k rows: two shards (one per mock device), three 256 MB staging chunks each
    const int64_t n = 500'000;
    std::vector<float> x((size_t)(n * (d + 7)), 1.0f);                 // row stride d + 7
    for (int bytes : {1, 2, 4, 8}) {
        std::vector<uint8_t> codes((size_t)(n * (M + 3) * bytes), 0xee);   // row stride M + 3 elements
        CHECK(pqhip_quantize_batch_f32(pq, x.data(), n, d + 7, 1, codes.data(), bytes, M + 3, 1) == PQHIP_OK);
        for (int64_t i = 0; i < n; i += 99991)
            for (int b = 0; b < 3 * bytes; ++b) CHECK(codes[(size_t)((i * (M + 3) + M) * bytes + b)] == 0xee);   // gaps untouched
    }
    {   // transposed input (column stride n), column-major codes
        std::vector<float> xt((size_t)(d * 4099), 2.0f);
        std::vector<uint32_t> ct((size_t)(M * 4099));
        CHECK(pqhip_quantize_batch_f32(opq, xt.data(), 4099, 1, 4099, ct.data(), 4, 1, 4099) == PQHIP_OK);
        CHECK(pqhip_quantize_batch_f32(pq, xt.data(), 0, 1, 4099, ct.data(), 4, 1, 4099) == PQHIP_OK);   // empty batch
    }
    {   // index width: K - 1 = 699 does not fit u8
        std::vector<float> xs((size_t)(5000 * 24), 1.f);
        std::vector<uint8_t> c8((size_t)(5000 * 3));
        std::vector<uint16_t> c16((size_t)(5000 * 3));
        CHECK(pqhip_quantize_batch_f32(wide, xs.data(), 5000, 24, 1, c8.data(), 1, 3, 1) == PQHIP_EINDEX_WIDTH);
        CHECK(pqhip_quantize_batch_f32(wide, xs.data(), 5000, 24, 1, c16.data(), 2, 3, 1) == PQHIP_OK);
    }
    {   // reconstruct: strided codes of every width into a strided output; a code >= K in the LAST row
        for (int bytes : {1, 2, 4, 8}) {
            std::vector<uint8_t> codes((size_t)(n * (M + 1) * bytes), 0);
            std::vector<float> out((size_t)(n * (d + 5)), -1.f);
            CHECK(pqhip_reconstruct_batch_f32(opq, codes.data(), bytes, n, M + 1, 1, out.data(), d + 5, 1) == PQHIP_OK);
            CHECK(out[(size_t)(d)] == -1.f && out[(size_t)((n - 1) * (d + 5) + d + 4)] == -1.f);
            if (bytes > 1) {
                codes[(size_t)(((n - 1) * (M + 1) + M - 1) * bytes + 1)] = 0x7f;        // 0x7f00 >= 256
                CHECK(pqhip_reconstruct_batch_f32(pq, codes.data(), bytes, n, M + 1, 1, out.data(), d + 5, 1) == PQHIP_ECODE_RANGE);
            }
        }
    }
    {   // scratch leases and flag tables from several host threads on the same OPQ / K > 256 codebooks
        std::vector<std::thread> th;
        std::vector<int32_t> rc(6, -1);
        for (int t = 0; t < 6; ++t)
            th.emplace_back([&, t] {
                std::vector<float> xs((size_t)((20000 + 7000 * t) * d), 1.f);
                std::vector<uint32_t> c((size_t)((20000 + 7000 * t) * M));
                int32_t r = pqhip_quantize_batch_f32_dev(opq, t & 1, xs.data(), 20000 + 7000 * t, d, c.data(), 1, M, nullptr);
                std::vector<float> xw((size_t)((3000 + 900 * t) * 24), 1.f);
                if (r == PQHIP_OK) r = pqhip_quantize_batch_f32_dev(wide, t & 1, xw.data(), 3000 + 900 * t, 24, c.data(), 4, 3, nullptr);
                if (r == PQHIP_OK) r = pqhip_check_codes_dev(opq, t & 1, (void*)(intptr_t)(0x100 + t));
                rc[(size_t)t] = r;
            });
        for (auto& t : th) t.join();
        for (int32_t r : rc) CHECK(r == PQHIP_OK);
    }
    {   // training entry points: resident matrix upload from a strided host matrix, k-means work buffers
        std::vector<float> xs((size_t)(70000 * (d + 2)), 0.5f);
        pqhip_matrix* mx = nullptr;
        CHECK(pqhip_matrix_upload_f32(ctx, 1, xs.data(), 70000, d, d + 2, 1, &mx) == PQHIP_OK && pqhip_matrix_rows(mx) == 70000);
        std::vector<float> qq = q, loss((size_t)M);
        CHECK(pqhip_kmeans_iterations_f32_dev(ctx, 1, qq.data(), M, K, dsub, pqhip_matrix_device_ptr(mx), 70000, d, 2, loss.data(), nullptr) == PQHIP_OK);
        std::vector<float> cross((size_t)(d * d));
        CHECK(pqhip_opq_train_step_f32_dev(ctx, 1, qq.data(), M, K, dsub, P.data(), pqhip_matrix_device_ptr(mx), 70000, d, cross.data(), nullptr) == PQHIP_OK);
        pqhip_matrix_destroy(mx);
        std::vector<uint64_t> a((size_t)9000);
        CHECK(pqhip_cluster_assignments_f32(ctx, q.data(), K, dsub, xs.data(), 9000, d + 2, 1, a.data(), 8) == PQHIP_OK);
    }
    // argument errors never touch memory
    CHECK(pqhip_quantize_batch_f32(pq, nullptr, 5, d, 1, nullptr, 1, M, 1) == PQHIP_EINVAL);
    CHECK(pqhip_quantize_batch_f32_dev(pq, 7, x.data(), 5, d, x.data(), 1, M, nullptr) == PQHIP_ENODEV);
    pqhip_codebook* bad = nullptr;
    CHECK(pqhip_codebook_create(ctx, q.data(), 0, K, dsub, nullptr, &bad) == PQHIP_ESHAPE && bad == nullptr);
    pqhip_codebook_destroy(wide);
    pqhip_codebook_destroy(opq);
    pqhip_codebook_destroy(pq);
    pqhip_ctx_destroy(ctx);
    std::printf("host logic under sanitizers: all checks passed\n");
    return 0;
}
