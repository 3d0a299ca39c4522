// The device-codebook cache policy of the reference-side binding (include/reductive_amd/codebook_cache.hpp,
// mirrored by rust/pqhip_ffi.rs): content-validated hits, replacement of stale images, bounded size.
// CPU part: counting handles.  GPU part (argument "gpu"): real pqhip_codebook handles -- a quantizer
// mutated in place (training) or an allocation reused by a new `Pq` must give the NEW codes.
// Exit 0 = passed, 77 = GPU part skipped (no device).
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#include "reductive_amd/codebook_cache.hpp"
#include "pqhip.h"

using namespace reductive_amd;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

struct Counters { int live = 0, created = 0, destroyed = 0; };
static int* fake_create(void* u, const float*, int64_t, int64_t, int64_t, const float*)
{
    auto* c = static_cast<Counters*>(u);
    ++c->live; ++c->created;
    return new int(c->created);
}
static void fake_destroy(void* u, int* h)
{
    auto* c = static_cast<Counters*>(u);
    --c->live; ++c->destroyed;
    delete h;
}

static pqhip_codebook* real_create(void* u, const float* q, int64_t M, int64_t K, int64_t dsub, const float* p)
{
    pqhip_codebook* cb = nullptr;
    return pqhip_codebook_create(static_cast<pqhip_ctx*>(u), q, M, K, dsub, p, &cb) == PQHIP_OK ? cb : nullptr;
}
static void real_destroy(void*, pqhip_codebook* cb) { pqhip_codebook_destroy(cb); }

int main(int argc, char** argv)
{
    {   // ---- policy, no device ----
        Counters c;
        CodebookCache<int*> cache(3, fake_create, fake_destroy, &c);
        std::vector<float> q(2 * 4 * 3, 1.0f), P(36, 0.5f);
        int* h1 = cache.get(q.data(), 2, 4, 3, nullptr);
        CHECK(h1 && cache.get(q.data(), 2, 4, 3, nullptr) == h1 && cache.hits() == 1 && c.created == 1);
        // the projection pointer is part of the key: PQ and OPQ views of the same quantizers are two images
        int* h2 = cache.get(q.data(), 2, 4, 3, P.data());
        CHECK(h2 && h2 != h1 && c.live == 2);
        // centroids mutated in place (k-means / OPQ training): same address, new contents -> replaced, old destroyed
        q[5] = 2.0f;
        int* h3 = cache.get(q.data(), 2, 4, 3, nullptr);
        CHECK(h3 && cache.replaced() == 1 && c.destroyed == 1 && c.live == 2);
        CHECK(cache.get(q.data(), 2, 4, 3, nullptr) == h3);
        // a projection mutated in place is detected too
        P[7] = -0.5f;
        int* h4 = cache.get(q.data(), 2, 4, 3, P.data());
        CHECK(h4 && cache.replaced() == 2 && c.live == 2);
        // a dropped Pq whose allocation is reused by another quantizer of the same size
        {
            std::unique_ptr<std::vector<float>> a(new std::vector<float>(24, 3.0f));
            const float* addr = a->data();
            int* ha = cache.get(addr, 2, 4, 3, nullptr);
            CHECK(ha && c.live == 3);
            std::fill(a->begin(), a->end(), 4.0f);      // stands for: freed, then reallocated at the same address
            int* hb = cache.get(addr, 2, 4, 3, nullptr);
            CHECK(hb && cache.replaced() == 3 && c.live == 3);
            // same address, other shape -> other key
            CHECK(cache.get(addr, 1, 8, 3, nullptr) && c.live == 3 && cache.evicted() == 1);   // capacity 3: LRU evicted
        }
        // bounded: many distinct quantizers never hold more than `capacity` device images
        std::vector<std::vector<float>> many(20, std::vector<float>(24, 0.f));
        for (size_t i = 0; i < many.size(); ++i) { many[i][0] = (float)i; CHECK(cache.get(many[i].data(), 2, 4, 3, nullptr)); CHECK(c.live <= 3); }
        cache.clear();
        CHECK(c.live == 0 && c.created == c.destroyed);
        // a failing create is reported, not cached
        CodebookCache<int*> failing(2, [](void*, const float*, int64_t, int64_t, int64_t, const float*) -> int* { return nullptr; }, fake_destroy, &c);
        CHECK(failing.get(q.data(), 2, 4, 3, nullptr) == nullptr && failing.size() == 0);
        CHECK(content_hash("abcdefgh1", 9) != content_hash("abcdefgh2", 9));
    }
    std::printf("cache policy checks passed\n");
    if (argc < 2 || std::strcmp(argv[1], "gpu") != 0) return 0;
    int32_t ndev = 0;
    if (pqhip_device_count(&ndev) != PQHIP_OK || ndev == 0) { std::printf("no GPU\n"); return 77; }
    {   // ---- stale images would be silently wrong codes: run real quantizers through the cache ----
        pqhip_ctx* ctx = nullptr;
        CHECK(pqhip_ctx_create(nullptr, 0, &ctx) == PQHIP_OK);
        {
            CodebookCache<pqhip_codebook*> cache(2, real_create, real_destroy, ctx);
            const int64_t M = 2, K = 2, dsub = 3, n = 4;
            std::vector<float> q = {1, 0, 0, 0, 1, 0, 1, -1, 0, 0, 1, 0};                      // pq.rs:400-407
            const std::vector<float> x = {0, 2, 0, -0.5f, 0, 0, 1, -0.2f, 0, 0.5f, 0.5f, 0,
                                          -0.2f, 0.2f, 0, 0, -2, 0, 1, 0.2f, 0, 0, -2, 0};      // pq.rs:378-385
            std::vector<uint8_t> codes(n * M);
            auto encode = [&](pqhip_codebook* cb) {
                return pqhip_quantize_batch_f32(cb, x.data(), n, M * dsub, 1, codes.data(), 1, M, 1) == PQHIP_OK;
            };
            pqhip_codebook* cb = cache.get(q.data(), M, K, dsub, nullptr);
            CHECK(cb && encode(cb));
            const std::vector<uint8_t> want = {1, 1, 0, 1, 1, 0, 0, 0};                         // pq.rs:387-389
            CHECK(codes == want);
            // swap the two centroids of every subquantizer IN PLACE: the codes must flip, not stay
            for (int64_t m = 0; m < M; ++m)
                for (int64_t e = 0; e < dsub; ++e) std::swap(q[(m * K + 0) * dsub + e], q[(m * K + 1) * dsub + e]);
            pqhip_codebook* cb2 = cache.get(q.data(), M, K, dsub, nullptr);
            CHECK(cb2 && cache.replaced() == 1 && encode(cb2));
            for (size_t i = 0; i < want.size(); ++i) CHECK(codes[i] == (uint8_t)(1 - want[i]));
            CHECK(cache.get(q.data(), M, K, dsub, nullptr) == cb2 && cache.hits() == 1);
        }
        pqhip_ctx_destroy(ctx);
    }
    std::printf("all checks passed (GPU)\n");
    return 0;
}
