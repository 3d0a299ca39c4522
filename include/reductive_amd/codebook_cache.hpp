// reductive_amd/codebook_cache.hpp -- the device-codebook cache of the reference-side binding.
//
// `Pq<A>` must keep `#[derive(Clone, Debug, PartialEq)]` and literal construction (src/pq/pq.rs:28-32,
// opq.rs:95-98, gaussian_opq.rs:64-67), so the binding cannot add a handle field to it; the device image
// of a quantizer lives in an external cache instead (SURVEY.md 8b "Ownership").  This header is the
// cache policy of rust/pqhip_ffi.rs in C++, so that it is EXECUTED by the test-suite
// (tests/cpp/test_codebook_cache.cpp) -- the Rust source cannot be compiled in this image.
//
// Round 3: `Pq<f32>` is `Send + Sync` and callers quantize from many threads, so
//   * the cache mutex is held for lookup / insert / evict ONLY -- never across a GPU call, never across the
//     creation of a device image;
//   * `get()` returns a PIN (shared ownership of the entry, Rust: `Arc<Entry>`): an entry evicted or replaced
//     while calls are running on it is destroyed when its last pin is dropped, so eviction waits for users and
//     users never wait for each other;
//   * a hit is validated by (data pointers, lengths, shape), the FULL 64-bit content hash of quantizers and
//     projection, and a GENERATION counter that every training exit bumps (`invalidate()`).  Round 3 validated a
//     sample of the arrays (head, tail, 256 spread words) to save ~70 us per call; a `Pq` dropped and another one
//     allocated at the same address that differs only outside the sample (one fine-tuned centroid) was then
//     served the OLD device image -- silently wrong codes.  Correctness is not
//     what gets traded: the hash is complete again, and the cost went into the hash instead -- eight independent
//     lanes, one AES round per 16 bytes where the host has AES-NI (multiply-xorshift lanes otherwise), run at the
//     speed of the cache the arrays sit in (measured on the GPU boxes: tests/cpp/test_codebook_cache.cpp prints it),
//     where the word-serial FNV chain of round 2 ran at ~5 GB/s.  Every lane update is a bijection of the lane
//     state, so ANY single changed word changes the hash unless the final 64-bit fold collides (2^-64);
//   * at most `capacity` entries, least recently used evicted -- no unbounded device memory.
// The handle type and its create / destroy functions are template parameters: the product instantiates it
// with pqhip_codebook*, the CPU unit test with counters.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <list>
#include <memory>
#include <mutex>

namespace reductive_amd {

// Full-content checksum for change detection (not a cryptographic hash): EVERY byte is read.
// Portable form: blocks of 64 bytes feed eight independent 64-bit lanes (lane j sees word j of every block),
// h_j = xorshift((h_j ^ w) * prime) -- a bijection of the lane state per step, so a single changed word always
// changes its lane; the eight multiply chains overlap, the loop runs at one 64-bit multiply per cycle (8 B/cycle).
inline uint64_t content_hash_lanes(const void* data, size_t bytes, uint64_t h)
{
    constexpr uint64_t kPrime = 0x100000001b3ull;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t l[8];
    for (int j = 0; j < 8; ++j) l[j] = h ^ (0x9e3779b97f4a7c15ull * (uint64_t)(j + 1));
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        uint64_t w[8];
        __builtin_memcpy(w, p + i, 64);
        for (int j = 0; j < 8; ++j) {
            l[j] = (l[j] ^ w[j]) * kPrime;
            l[j] ^= l[j] >> 29;
        }
    }
    h ^= (uint64_t)bytes * kPrime;
    for (int j = 0; j < 8; ++j) {
        h = (h ^ l[j]) * kPrime;
        h ^= h >> 29;
    }
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        __builtin_memcpy(&w, p + i, 8);
        h = (h ^ w) * kPrime;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) h = (h ^ p[i]) * kPrime;
    return h;
}

#if defined(__x86_64__)
// x86 hosts with AES-NI (every EPYC / Xeon a GPU node is built from): eight 128-bit lanes, one AES round per
// 16 bytes -- state' = AESENC(state ^ block, key) is a permutation of the state for every block and of the block
// for every state, so the single-changed-word argument holds here too -- at 16-32 B/cycle: the hash of a codebook
// runs at the speed of the cache level it sits in.
typedef long long pqhip_v2di __attribute__((vector_size(16), aligned(16)));
typedef long long pqhip_v2di_u __attribute__((vector_size(16), aligned(1)));
__attribute__((target("aes,sse2"))) inline uint64_t content_hash_aes(const void* data, size_t bytes, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    const pqhip_v2di key = {(long long)0x9e3779b97f4a7c15ull, (long long)0xc2b2ae3d27d4eb4full};
    pqhip_v2di s[8];
    for (int j = 0; j < 8; ++j) s[j] = pqhip_v2di{(long long)(h + (uint64_t)j), (long long)(~h ^ ((uint64_t)j << 32))};
    size_t i = 0;
    for (; i + 128 <= bytes; i += 128)
        for (int j = 0; j < 8; ++j) {
            const pqhip_v2di b = *reinterpret_cast<const pqhip_v2di_u*>(p + i + 16 * j);
            s[j] = __builtin_ia32_aesenc128(s[j] ^ b, key);
        }
    pqhip_v2di acc = {(long long)bytes, (long long)h};
    for (int j = 0; j < 8; ++j) acc = __builtin_ia32_aesenc128(acc ^ s[j], key);
    for (; i + 16 <= bytes; i += 16) {
        const pqhip_v2di b = *reinterpret_cast<const pqhip_v2di_u*>(p + i);
        acc = __builtin_ia32_aesenc128(acc ^ b, key);
    }
    if (i < bytes) {
        unsigned char tail[16] = {};
        __builtin_memcpy(tail, p + i, bytes - i);
        pqhip_v2di b;
        __builtin_memcpy(&b, tail, 16);
        acc = __builtin_ia32_aesenc128(acc ^ b, key);
    }
    acc = __builtin_ia32_aesenc128(acc, key);
    acc = __builtin_ia32_aesenc128(acc, key);
    return (uint64_t)acc[0] ^ (uint64_t)acc[1];
}
#endif

inline uint64_t content_hash(const void* data, size_t bytes, uint64_t h = 0xcbf29ce484222325ull)
{
#if defined(__x86_64__)
    static const bool has_aes = __builtin_cpu_supports("aes");
    if (has_aes) return content_hash_aes(data, bytes, h);
#endif
    return content_hash_lanes(data, bytes, h);
}

template <typename Handle>
class CodebookCache {
public:
    struct Key {
        const float* q;
        size_t q_len;
        const float* p;   // nullptr: no projection
        int64_t M, K, dsub;
        bool operator==(const Key& o) const { return q == o.q && q_len == o.q_len && p == o.p && M == o.M && K == o.K && dsub == o.dsub; }
    };
    using Create = Handle (*)(void* user, const float* q, int64_t M, int64_t K, int64_t dsub, const float* p);
    using Destroy = void (*)(void* user, Handle h);

    // One device image.  Destroyed when the cache has dropped it AND the last running call has released its pin.
    struct Slot {
        Handle handle;
        Destroy destroy;
        void* user;
        Slot(Handle h, Destroy d, void* u) : handle(h), destroy(d), user(u) {}
        ~Slot() { destroy(user, handle); }
        Slot(const Slot&) = delete;
        Slot& operator=(const Slot&) = delete;
    };
    using Pin = std::shared_ptr<const Slot>;   // Rust: Arc<Entry>

    CodebookCache(size_t capacity, Create create, Destroy destroy, void* user)
        : cap_(capacity ? capacity : 1), create_(create), destroy_(destroy), user_(user) {}
    ~CodebookCache() { clear(); }
    CodebookCache(const CodebookCache&) = delete;
    CodebookCache& operator=(const CodebookCache&) = delete;

    // Pinned device image of (quantizers [M][K][dsub], projection [d][d] or nullptr); a null pin when creation
    // fails (the caller then stays on its CPU path).  Use `pin->handle` for ONE call and drop the pin.
    Pin get(const float* q, int64_t M, int64_t K, int64_t dsub, const float* p)
    {
        const size_t q_len = (size_t)(M * K * dsub), d = (size_t)(M * dsub);
        const Key key{q, q_len, p, M, K, dsub};
        uint64_t h = content_hash(q, q_len * sizeof(float));             // FULL contents (outside the mutex: reads caller memory only)
        if (p) h = content_hash(p, d * d * sizeof(float), h);
        const uint64_t gen = generation_.load(std::memory_order_acquire);
        Pin stale;                                                       // destroyed after the mutex is released
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto it = entries_.begin(); it != entries_.end(); ++it) {
                if (!(it->key == key)) continue;
                if (it->hash == h && it->gen == gen) {                   // same memory, same contents, nothing trained since
                    entries_.splice(entries_.begin(), entries_, it);
                    ++hits_;
                    return entries_.front().slot;
                }
                stale = std::move(it->slot);                             // same address, other contents: stale image
                entries_.erase(it);
                ++replaced_;
                break;
            }
        }
        stale.reset();
        // the device image is built with the mutex RELEASED (an allocation, a copy and three preparation kernels)
        Handle nh = create_(user_, q, M, K, dsub, p);
        if (!nh) return Pin();
        Pin fresh = std::make_shared<const Slot>(nh, destroy_, user_);
        std::list<Entry> dropped;                                        // evicted entries die outside the mutex too
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto it = entries_.begin(); it != entries_.end(); ++it)
                if (it->key == key) {                                    // another thread built the same image meanwhile
                    if (it->hash == h && it->gen == gen) return it->slot;   // ours (`fresh`) is destroyed on return
                    dropped.splice(dropped.end(), entries_, it);
                    break;
                }
            entries_.push_front(Entry{key, h, gen, fresh});
            ++created_;
            while (entries_.size() > cap_) {
                auto last = std::prev(entries_.end());
                dropped.splice(dropped.end(), entries_, last);
                ++evicted_;
            }
        }
        return fresh;
    }
    // Called by every entry point that rewrites quantizers or projections in place (the k-means / OPQ training
    // steps): no image created before this call is ever served again.
    void invalidate() { generation_.fetch_add(1, std::memory_order_acq_rel); }
    void clear()
    {
        std::list<Entry> dropped;
        {
            std::lock_guard<std::mutex> g(mu_);
            dropped.swap(entries_);
        }
    }
    size_t size() const { std::lock_guard<std::mutex> g(mu_); return entries_.size(); }
    size_t hits() const { return hits_; }
    size_t created() const { return created_; }
    size_t replaced() const { return replaced_; }
    size_t evicted() const { return evicted_; }

private:
    struct Entry { Key key; uint64_t hash; uint64_t gen; Pin slot; };
    std::list<Entry> entries_;
    size_t cap_;
    Create create_;
    Destroy destroy_;
    void* user_;
    mutable std::mutex mu_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<size_t> hits_{0}, created_{0}, replaced_{0}, evicted_{0};
};

}  // namespace reductive_amd
