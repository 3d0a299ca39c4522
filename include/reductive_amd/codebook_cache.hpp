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
//   * a hit is validated by (data pointers, lengths, shape), a SAMPLED 64-bit content hash -- the first and last
//     4 KiB of quantizers and projection plus 256 evenly spaced 8-byte words of each: a few microseconds, where
//     hashing all of a d = 768 OPQ quantizer (3.1 MB) cost about as much as a 4,096-row encode -- and a
//     GENERATION counter that the training entry points bump (`invalidate()`): centroids updated in place by
//     `try_kmeans_iterations` / `train_step` can never be served from a stale image.  A dropped `Pq` whose
//     allocation is reused by another quantizer, or centroids rewritten by the CPU trainer (every centroid moves),
//     change the sample; an edit confined to bytes outside the sample AND outside the binding is the one case the
//     sample misses (the reference has no such code path: `Pq` exposes no `&mut` access to its arrays);
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

// FNV-1a over 8-byte words (tail bytes folded in): a checksum for change detection, not a cryptographic hash.
inline uint64_t content_hash(const void* data, size_t bytes, uint64_t h = 0xcbf29ce484222325ull)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        __builtin_memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

// The sampled form: everything when the array is small, else head + tail + 256 words spread over the middle.
inline uint64_t sampled_hash(const void* data, size_t bytes, uint64_t h = 0xcbf29ce484222325ull)
{
    constexpr size_t kEdge = 4096, kWords = 256;
    if (bytes <= 2 * kEdge + 8 * kWords) return content_hash(data, bytes, h);
    const unsigned char* p = static_cast<const unsigned char*>(data);
    h = content_hash(p, kEdge, h);
    h = content_hash(p + bytes - kEdge, kEdge, h);
    const size_t span = bytes - 2 * kEdge, step = (span / kWords) & ~(size_t)7;
    for (size_t i = 0; i < kWords; ++i) {
        uint64_t w;
        __builtin_memcpy(&w, p + kEdge + i * step, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    return h;
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
        uint64_t h = sampled_hash(q, q_len * sizeof(float));             // (outside the mutex: reads caller memory only)
        if (p) h = sampled_hash(p, d * d * sizeof(float), h);
        const uint64_t gen = generation_.load(std::memory_order_acquire);
        Pin stale;                                                       // destroyed after the mutex is released
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto it = entries_.begin(); it != entries_.end(); ++it) {
                if (!(it->key == key)) continue;
                if (it->hash == h && it->gen == gen) {                   // same memory, same sample, nothing trained since
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
