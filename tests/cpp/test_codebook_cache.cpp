// The device-codebook cache policy of the reference-side binding (include/reductive_amd/codebook_cache.hpp,
// mirrored by rust/pqhip_ffi.rs): content-validated hits, replacement of stale images, bounded size.
// CPU part: counting handles.  GPU part (argument "gpu"): real pqhip_codebook handles -- a quantizer
// mutated in place (training) or an allocation reused by a new `Pq` must give the NEW codes.
// Exit 0 = passed, 77 = GPU part skipped (no device).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include "reductive_amd/codebook_cache.hpp"
#include "pqhip.h"

// the CPU oracle (test infrastructure): checker of the concurrent calls' codes
extern "C" int pqo_quantize_batch(const float* cb, int64_t M, int64_t K, int64_t dsub, const float* P, const float* x, int64_t n,
                                  int64_t x_rs, int64_t x_cs, void* out, int out_bytes, int64_t o_rs, int64_t o_cs, int n_threads);

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
        auto get = [&](const float* qq, int64_t M, int64_t K, int64_t ds, const float* pp) -> int* {
            auto pin = cache.get(qq, M, K, ds, pp);
            return pin ? pin->handle : nullptr;          // (the pin is dropped: the cache still owns the entry)
        };
        int* h1 = get(q.data(), 2, 4, 3, nullptr);
        CHECK(h1 && get(q.data(), 2, 4, 3, nullptr) == h1 && cache.hits() == 1 && c.created == 1);
        // the projection pointer is part of the key: PQ and OPQ views of the same quantizers are two images
        int* h2 = get(q.data(), 2, 4, 3, P.data());
        CHECK(h2 && h2 != h1 && c.live == 2);
        // centroids mutated in place (k-means / OPQ training): same address, new contents -> replaced, old destroyed
        q[5] = 2.0f;
        int* h3 = get(q.data(), 2, 4, 3, nullptr);
        CHECK(h3 && cache.replaced() == 1 && c.destroyed == 1 && c.live == 2);
        CHECK(get(q.data(), 2, 4, 3, nullptr) == h3);
        // a projection mutated in place is detected too
        P[7] = -0.5f;
        int* h4 = get(q.data(), 2, 4, 3, P.data());
        CHECK(h4 && cache.replaced() == 2 && c.live == 2);
        // a dropped Pq whose allocation is reused by another quantizer of the same size
        {
            std::unique_ptr<std::vector<float>> a(new std::vector<float>(24, 3.0f));
            const float* addr = a->data();
            int* ha = get(addr, 2, 4, 3, nullptr);
            CHECK(ha && c.live == 3);
            std::fill(a->begin(), a->end(), 4.0f);      // stands for: freed, then reallocated at the same address
            int* hb = get(addr, 2, 4, 3, nullptr);
            CHECK(hb && cache.replaced() == 3 && c.live == 3);
            // same address, other shape -> other key
            CHECK(get(addr, 1, 8, 3, nullptr) && c.live == 3 && cache.evicted() == 1);   // capacity 3: LRU evicted
        }
        // bounded: many distinct quantizers never hold more than `capacity` device images
        std::vector<std::vector<float>> many(20, std::vector<float>(24, 0.f));
        for (size_t i = 0; i < many.size(); ++i) { many[i][0] = (float)i; CHECK(get(many[i].data(), 2, 4, 3, nullptr)); CHECK(c.live <= 3); }
        // a PIN keeps an evicted / replaced image alive until the call that holds it is over (eviction waits for users)
        {
            std::vector<float> held(24, 9.0f);
            auto pin = cache.get(held.data(), 2, 4, 3, nullptr);
            CHECK(pin && c.live <= 3);
            const int before = c.destroyed;
            for (size_t i = 0; i < 5; ++i) CHECK(get(many[i].data(), 2, 4, 3, nullptr));   // pushes `held` out of the cache
            CHECK(c.live == 4 && *pin->handle > 0);             // 3 cached + the pinned, evicted one -- still usable
            pin.reset();
            CHECK(c.live == 3 && c.destroyed > before);
        }
        // ADVERSARIAL: a `Pq` dropped and another allocated at the SAME address, same shape, that
        // differs in ONE centroid element anywhere -- in particular at offsets the round-3 sampled hash never read
        // (its sample: first / last 4 KiB + 256 words spread over the middle) -- must get a NEW device image.  No
        // invalidate() is called: nothing but the contents says that the quantizer changed.
        {
            std::vector<float> big(2 * 64 * 3000, 1.0f);        // 1.5 MB
            int* g0 = get(big.data(), 2, 64, 3000, nullptr);
            CHECK(g0 && get(big.data(), 2, 64, 3000, nullptr) == g0);
            int last_id = *g0;                                  // (fake handles carry their creation number)
            const size_t n = big.size();
            for (size_t off : {(size_t)100000, (size_t)1025, (size_t)(n / 2 + 1), (size_t)(n - 1030), (size_t)4099, (size_t)(n / 3 + 5), (size_t)(n - 1)}) {
                big[off] = 2.0f + (float)off;                   // one element of one centroid
                int* g = get(big.data(), 2, 64, 3000, nullptr);
                CHECK(g && *g != last_id);                      // replaced, never served from the old image
                last_id = *g;
                CHECK(get(big.data(), 2, 64, 3000, nullptr) == g);   // unchanged contents: a hit
            }
            // a one-bit edit (-0.0f for +0.0f compares equal as floats, differs as contents)
            big[n / 2] = 0.0f;
            int* gz = get(big.data(), 2, 64, 3000, nullptr);
            const int zid = gz ? *gz : -1;                      // (the handle dies with its replacement below)
            big[n / 2] = -0.0f;
            int* gm = get(big.data(), 2, 64, 3000, nullptr);
            CHECK(gz && gm && *gm != zid);
            // the same for the projection
            std::vector<float> bigP(1024 * 1024, 0.25f), qq(2 * 4 * 512, 1.0f);
            int* p0 = get(qq.data(), 2, 4, 512, bigP.data());
            const int pid = p0 ? *p0 : -1;
            bigP[777777] = 0.5f;
            int* p1 = get(qq.data(), 2, 4, 512, bigP.data());
            CHECK(p0 && p1 && *p1 != pid);
        }
        // the training entry points bump the generation: nothing created before is served again even when the
        // contents are byte-for-byte what they were (belt and braces beside the full hash)
        {
            std::vector<float> big(2 * 64 * 3000, 1.0f);
            int* g1 = get(big.data(), 2, 64, 3000, nullptr);
            CHECK(g1 && get(big.data(), 2, 64, 3000, nullptr) == g1);
            const int id1 = *g1;
            cache.invalidate();
            int* g2 = get(big.data(), 2, 64, 3000, nullptr);
            CHECK(g2 && *g2 != id1);
        }
        // cost of a validated hit (reported, not asserted): the headline codebook and a d = 768 OPQ quantizer
        for (size_t bytes : {(size_t)307200, (size_t)(786432 + 2359296)}) {
            std::vector<unsigned char> buf(bytes, 7);
            uint64_t acc = 0;
            auto t0 = std::chrono::steady_clock::now();
            const int reps = 200;
            for (int r = 0; r < reps; ++r) { buf[(size_t)r * 997 % bytes] ^= 1; acc ^= content_hash(buf.data(), bytes); }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            std::printf("content_hash: %zu bytes in %.1f us (%.1f GB/s) [%llx]\n", bytes, us, bytes / us * 1e-3, (unsigned long long)acc);
        }
        cache.clear();
        CHECK(c.live == 0 && c.created == c.destroyed);
        // a failing create is reported, not cached
        CodebookCache<int*> failing(2, [](void*, const float*, int64_t, int64_t, int64_t, const float*) -> int* { return nullptr; }, fake_destroy, &c);
        CHECK(!failing.get(q.data(), 2, 4, 3, nullptr) && failing.size() == 0);
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
            auto pin1 = cache.get(q.data(), M, K, dsub, nullptr);
            pqhip_codebook* cb = pin1 ? pin1->handle : nullptr;
            CHECK(cb && encode(cb));
            const std::vector<uint8_t> want = {1, 1, 0, 1, 1, 0, 0, 0};                         // pq.rs:387-389
            CHECK(codes == want);
            // swap the two centroids of every subquantizer IN PLACE: the codes must flip, not stay
            for (int64_t m = 0; m < M; ++m)
                for (int64_t e = 0; e < dsub; ++e) std::swap(q[(m * K + 0) * dsub + e], q[(m * K + 1) * dsub + e]);
            auto pin2 = cache.get(q.data(), M, K, dsub, nullptr);
            pqhip_codebook* cb2 = pin2 ? pin2->handle : nullptr;
            CHECK(cb2 && cache.replaced() == 1 && encode(cb2));
            for (size_t i = 0; i < want.size(); ++i) CHECK(codes[i] == (uint8_t)(1 - want[i]));
            CHECK(cache.get(q.data(), M, K, dsub, nullptr)->handle == cb2 && cache.hits() == 1);
        }
        // ---- concurrency: 4 host threads x 2 distinct quantizers through ONE cache, 32 calls of
        // 2,048 rows each (a small batch is latency-bound -- launch, PCIe round trip, stream synchronisation -- so
        // concurrent callers CAN overlap; a large one is PCIe-bound and four of them cannot beat the link: at 8,192 rows
        // the transfer is already 200 of a call's 345 us and the ratio moved between 0.6 and 0.86 from box to box; at 2,048
        // rows with two staging sets per device slot it was 0.56-0.63, half of the threads waiting for a set).  The cache
        // mutex covers lookups only and the device slot leases one of its staging sets per call: the wall time of the four
        // threads is reported against the same 128 calls made one after the other; every code equals the CPU oracle's.
        {
            CodebookCache<pqhip_codebook*> cache(4, real_create, real_destroy, ctx);
            const int64_t M = 15, K = 256, dsub = 20, d = M * dsub, n = 2048;   // 2.4 MB per call: 50 us of PCIe beside ~100 us of latencies
            const int CALLS = 32;
            std::vector<std::vector<float>> qs(2, std::vector<float>((size_t)(M * K * dsub)));
            unsigned s = 99;
            auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xffff) / 32768.0f - 1.0f; };
            for (auto& q : qs) for (auto& v : q) v = rnd();
            std::vector<std::vector<float>> xs(4, std::vector<float>((size_t)(n * d)));
            for (auto& x : xs) for (auto& v : x) v = rnd();
            std::vector<std::vector<uint8_t>> got(4, std::vector<uint8_t>((size_t)(n * M))), want = got;
            for (int t = 0; t < 4; ++t)
                CHECK(pqo_quantize_batch(qs[t & 1].data(), M, K, dsub, nullptr, xs[t].data(), n, d, 1, want[t].data(), 1, M, 1, 8) == 0);
            auto call = [&](int t) {
                auto pin = cache.get(qs[t & 1].data(), M, K, dsub, nullptr);
                return pin && pqhip_quantize_batch_f32(pin->handle, xs[t].data(), n, d, 1, got[t].data(), 1, M, 1) == PQHIP_OK;
            };
            for (int t = 0; t < 4; ++t) CHECK(call(t));           // warm: device images, staging buffers
            double serial = 1e30, conc = 1e30;
            for (int rep = 0; rep < 6 && !(rep >= 2 && conc < 0.6 * serial); ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                for (int t = 0; t < 4; ++t) for (int c = 0; c < CALLS; ++c) CHECK(call(t));
                auto t1 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                std::vector<int> ok(4, 0);
                for (int t = 0; t < 4; ++t) th.emplace_back([&, t] { int good = 1; for (int c = 0; c < CALLS; ++c) good &= (int)call(t); ok[(size_t)t] = good; });
                for (auto& x : th) x.join();
                auto t2 = std::chrono::steady_clock::now();
                for (int v : ok) CHECK(v);
                serial = std::min(serial, std::chrono::duration<double>(t1 - t0).count());
                conc = std::min(conc, std::chrono::duration<double>(t2 - t1).count());
            }
            for (int t = 0; t < 4; ++t) CHECK(got[t] == want[t]);
            std::printf("cache concurrency: 128 calls of %lld rows serial %.2f ms, from 4 threads %.2f ms (ratio %.2f), hits %zu, images created %zu\n",
                        (long long)n, serial * 1e3, conc * 1e3, conc / serial, cache.hits(), cache.created());
            CHECK(cache.created() == 2);
            // The ratio is a REPORTED number (0.42-0.45 with four staging sets per device slot); the guard only catches
            // full serialisation (every caller waiting for one slot mutex gives ~1.0).  A tighter wall-clock assertion
            // on a shared 16-core-quota host once stopped `pytest -x` at 0.86.
            CHECK(conc < 0.95 * serial);
        }
        pqhip_ctx_destroy(ctx);
    }
    std::printf("all checks passed (GPU)\n");
    return 0;
}
