// Drives the host-resident entry points of the sanitized, mock-backed build of libpqhip (tests/mock_hip/
// Makefile) with the shapes that stress the host logic: strided and transposed inputs, every index width,
// wide strided outputs, several staging chunks per shard, two device slots, OPQ and K > 256 (scratch
// leases from several host threads), range errors, handle lifetimes.  Nothing is computed (kernel launches
// are no-ops in the mock): the pass criterion is "no AddressSanitizer / UBSan report and the status codes
// the C ABI promises".
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include "pqhip.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

// PQHIP_TEST_SCRATCH_ROWS=<n> (read by THIS driver, not by the library): n-row OPQ scratch chunks through the context
// option, so that every OPQ call walks many chunks and a remainder through one lease
static bool apply_test_options(pqhip_ctx* ctx)
{
    const char* e = std::getenv("PQHIP_TEST_SCRATCH_ROWS");
    if (e && pqhip_ctx_set_option(ctx, "opq_scratch_rows", std::atoll(e)) != PQHIP_OK) return false;
    return pqhip_ctx_set_option(ctx, "no_such_option", 1) == PQHIP_EINVAL;
}

extern "C" int mock_hip_violations(void);
extern "C" int mock_hip_selftest(void);
extern "C" int mock_hip_registrations(void);

// "devices" mode (MOCK_HIP_DEVICES=8; both sanitizer builds run it): the library's own row sharder over EIGHT device slots,
// every allocation / stream / event of the mock tagged with its device -- a pointer, stream or event of
// slot A used while the thread is on device B, or through a stream of B, fails the run.  Host-resident encode at d = 300 and
// at d = 768 / M = 48 (the size of BASELINE configs[4]) with a ragged row count, OPQ through the scratch leases, every index
// width, strided outputs, reconstruct with a range error in the LAST shard, device entry points on every slot from eight
// threads at once, a k-means call on the last slot.  With PQHIP_HOST_ZERO_COPY=1 the registered input leg runs too.
static int devices_mode()
{
    int32_t nd = 0;
    CHECK(pqhip_device_count(&nd) == PQHIP_OK && nd == 8);
    CHECK(mock_hip_selftest() == 5);                        // the mock does see cross-device misuse
    pqhip_ctx* ctx = nullptr;
    CHECK(pqhip_ctx_create(nullptr, 0, &ctx) == PQHIP_OK && pqhip_ctx_n_devices(ctx) == 8);
    CHECK(apply_test_options(ctx));
    struct Shape { int64_t M, K, dsub, n; bool opq; };
    // (the last two: the small-codebook kernels of round 4 -- 16x16x4 with the codebook image in LDS, and the candidate-list
    // kernel for 2-float sub-vectors, whose tables are built on the host and replicated on every device)
    for (const Shape sh : {Shape{15, 256, 20, 400003, false}, Shape{48, 256, 16, 90001, false}, Shape{15, 256, 20, 70001, true},
                           Shape{16, 16, 8, 50001, false}, Shape{10, 128, 2, 60001, false},
                           Shape{150, 256, 2, 30001, false}, Shape{128, 64, 1, 20001, false}}) {   // many groups of subquantizers (candidate lists, grouped reconstruct)
        const int64_t M = sh.M, K = sh.K, dsub = sh.dsub, d = M * dsub, n = sh.n;
        std::vector<float> q((size_t)(M * K * dsub)), P;
        for (size_t i = 0; i < q.size(); ++i) q[i] = (float)((i * 2654435761ull >> 8) & 0xffff) / 65536.0f;   // distinct centroids
        if (sh.opq) { P.assign((size_t)(d * d), 0.f); for (int64_t i = 0; i < d; ++i) P[(size_t)(i * d + i)] = 1.f; }
        pqhip_codebook* cb = nullptr;
        CHECK(pqhip_codebook_create(ctx, q.data(), M, K, dsub, sh.opq ? P.data() : nullptr, &cb) == PQHIP_OK);
        std::vector<float> x((size_t)(n * (d + 3)), 1.0f);                       // row stride d + 3
        for (int bytes : {1, 2, 8}) {
            std::vector<uint8_t> codes((size_t)(n * (M + 2) * bytes), 0xee);     // row stride M + 2 elements
            CHECK(pqhip_quantize_batch_f32(cb, x.data(), n, d + 3, 1, codes.data(), bytes, M + 2, 1) == PQHIP_OK);
            for (int64_t i = 0; i < n; i += 49999)
                for (int b = 0; b < 2 * bytes; ++b) CHECK(codes[(size_t)((i * (M + 2) + M) * bytes + b)] == 0xee);   // gaps untouched
        }
        {   // reconstruct: 8-byte codes, strided output; then a code >= K in the last row = the last shard
            std::vector<uint64_t> codes((size_t)(n * M), 0);
            std::vector<float> out((size_t)(n * (d + 1)), -1.f);
            CHECK(pqhip_reconstruct_batch_f32(cb, codes.data(), 8, n, M, 1, out.data(), d + 1, 1) == PQHIP_OK);
            CHECK(out[(size_t)d] == -1.f && out[(size_t)((n - 1) * (d + 1) + d)] == -1.f);
            codes[(size_t)((n - 1) * M + M - 1)] = (1ull << 32) + 1;             // (a valid index once truncated to 32 bits)
            CHECK(pqhip_reconstruct_batch_f32(cb, codes.data(), 8, n, M, 1, out.data(), d + 1, 1) == PQHIP_ECODE_RANGE);
        }
        {   // device entry points of all eight slots at once (host vectors stand in for caller-owned device memory)
            std::vector<std::thread> th;
            std::vector<int32_t> rc(8, -1);
            for (int t = 0; t < 8; ++t)
                th.emplace_back([&, t] {
                    const int64_t rows = 5000 + 1000 * t;
                    std::vector<float> xs((size_t)(rows * d), 1.f), out((size_t)(rows * d));
                    std::vector<uint16_t> c((size_t)(rows * M));
                    int32_t r = pqhip_quantize_batch_f32_dev(cb, t, xs.data(), rows, d, c.data(), 2, M, nullptr);
                    if (r == PQHIP_OK) r = pqhip_reconstruct_batch_f32_dev(cb, t, c.data(), 2, rows, M, out.data(), d, nullptr);
                    if (r == PQHIP_OK) r = pqhip_check_codes_dev(cb, t, nullptr);
                    rc[(size_t)t] = r;
                });
            for (auto& t : th) t.join();
            for (int32_t r : rc) CHECK(r == PQHIP_OK);
        }
        pqhip_codebook_destroy(cb);
    }
    {   // training entry points on the LAST slot: workspaces, the second stream, the cross-product groups
        const int64_t M = 15, K = 256, dsub = 20, d = M * dsub;
        std::vector<float> q((size_t)(M * K * dsub), 0.25f), P((size_t)(d * d), 0.f), xs((size_t)(30000 * d), 0.5f), loss((size_t)M), cross((size_t)(d * d));
        for (int64_t i = 0; i < d; ++i) P[(size_t)(i * d + i)] = 1.f;
        pqhip_matrix* mx = nullptr;
        CHECK(pqhip_matrix_upload_f32(ctx, 7, xs.data(), 30000, d, d, 1, &mx) == PQHIP_OK);
        CHECK(pqhip_kmeans_iterations_f32_dev(ctx, 7, q.data(), M, K, dsub, pqhip_matrix_device_ptr(mx), 30000, d, 2, loss.data(), nullptr) == PQHIP_OK);
        CHECK(pqhip_opq_train_step_f32_dev(ctx, 7, q.data(), M, K, dsub, P.data(), pqhip_matrix_device_ptr(mx), 30000, d, cross.data(), nullptr) == PQHIP_OK);
        pqhip_matrix_destroy(mx);
    }
    pqhip_ctx_destroy(ctx);
    if (const char* z = std::getenv("PQHIP_HOST_ZERO_COPY")) {
        if (z[0] == '1') CHECK(mock_hip_registrations() > 20);       // the registered leg really ran (every third request is refused)
    }
    CHECK(mock_hip_violations() == 0);
    std::printf("host logic under sanitizers (8 tagged devices): all checks passed\n");
    return 0;
}

// "threads" mode (the ThreadSanitizer build runs only this): many host threads lease, grow and release the scratch
// buffers of ONE OPQ codebook and ONE K > 256 codebook on both device slots at once, with sizes that force the pool to
// grow and buffers to be reallocated while other threads hold leases (ScratchLease::ptr() used to read the
// pool vector without the mutex), and share the per-stream flag table with more streams than it has slots.
static int threads_mode()
{
    pqhip_ctx* ctx = nullptr;
    CHECK(pqhip_ctx_create(nullptr, 0, &ctx) == PQHIP_OK && pqhip_ctx_n_devices(ctx) == 2);
    CHECK(apply_test_options(ctx));
    const int64_t M = 6, K = 64, dsub = 10, d = M * dsub;
    std::vector<float> q((size_t)(M * K * dsub), 0.25f), P((size_t)(d * d), 0.f);
    for (int64_t i = 0; i < d; ++i) P[(size_t)(i * d + i)] = 1.f;
    pqhip_codebook *opq = nullptr, *wide = nullptr;
    CHECK(pqhip_codebook_create(ctx, q.data(), M, K, dsub, P.data(), &opq) == PQHIP_OK);
    std::vector<float> qw((size_t)(3 * 700 * 8), 0.5f);
    CHECK(pqhip_codebook_create(ctx, qw.data(), 3, 700, 8, nullptr, &wide) == PQHIP_OK);
    const int NT = 12;
    std::vector<std::thread> th;
    std::vector<int32_t> rc((size_t)NT, -1);
    for (int t = 0; t < NT; ++t)
        th.emplace_back([&, t] {
            int32_t r = PQHIP_OK;
            for (int it = 0; it < 6 && r == PQHIP_OK; ++it) {
                const int64_t n = 3000 + 4000 * ((t + it) % 5);           // growing and shrinking requests
                std::vector<float> xs((size_t)(n * d), 1.f), out((size_t)(n * d));
                std::vector<uint32_t> c((size_t)(n * M));
                void* stream = (void*)(intptr_t)(0x100 + 16 * t + it);      // 72 distinct "streams" > 64 flag slots
                r = pqhip_quantize_batch_f32_dev(opq, (t + it) & 1, xs.data(), n, d, c.data(), 1, M, stream);
                if (r == PQHIP_OK) r = pqhip_reconstruct_batch_f32_dev(opq, (t + it) & 1, c.data(), 1, n, M, out.data(), d, stream);
                if (r == PQHIP_OK) r = pqhip_check_codes_dev(opq, (t + it) & 1, stream);
                std::vector<float> xw((size_t)(n * 24), 1.f);
                if (r == PQHIP_OK) r = pqhip_quantize_batch_f32_dev(wide, t & 1, xw.data(), n, 24, c.data(), 4, 3, stream);
            }
            rc[(size_t)t] = r;
        });
    for (auto& t : th) t.join();
    for (int32_t r : rc) CHECK(r == PQHIP_OK);
    // host-resident calls from two threads on one context (device-slot mutexes, staging reuse)
    {
        std::vector<std::thread> hs;
        std::vector<int32_t> hr(2, -1);
        for (int t = 0; t < 2; ++t)
            hs.emplace_back([&, t] {
                std::vector<float> xs((size_t)(30000 * d), 1.f);
                std::vector<uint8_t> c((size_t)(30000 * M));
                hr[(size_t)t] = pqhip_quantize_batch_f32(opq, xs.data(), 30000, d, 1, c.data(), 1, M, 1);
            });
        for (auto& t : hs) t.join();
        CHECK(hr[0] == PQHIP_OK && hr[1] == PQHIP_OK);
    }
    pqhip_codebook_destroy(wide);
    pqhip_codebook_destroy(opq);
    pqhip_ctx_destroy(ctx);
    CHECK(mock_hip_violations() == 0);
    std::printf("host logic under sanitizers (threads): all checks passed\n");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::string(argv[1]) == "threads") return threads_mode();
    if (argc > 1 && std::string(argv[1]) == "devices") return devices_mode();
    int32_t nd = 0;
    CHECK(pqhip_device_count(&nd) == PQHIP_OK && nd == 2);
    pqhip_ctx* ctx = nullptr;
    CHECK(pqhip_ctx_create(nullptr, 0, &ctx) == PQHIP_OK && pqhip_ctx_n_devices(ctx) == 2);
    CHECK(apply_test_options(ctx));
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
    CHECK(mock_hip_violations() == 0);
    std::printf("host logic under sanitizers: all checks passed\n");
    return 0;
}
