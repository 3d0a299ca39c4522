// reductive_amd/codebook_cache.hpp -- the device-codebook cache of the reference-side binding.
//
// `Pq<A>` must keep `#[derive(Clone, Debug, PartialEq)]` and literal construction (src/pq/pq.rs:28-32,
// opq.rs:95-98, gaussian_opq.rs:64-67), so the binding cannot add a handle field to it; the device image
// of a quantizer lives in an external cache instead (SURVEY.md 8b "Ownership").  This header is the
// cache policy of rust/pqhip_ffi.rs in C++, so that it is EXECUTED by the test-suite
// (tests/cpp/test_codebook_cache.cpp) -- the Rust source cannot be compiled in this image:
//   * key    = (quantizer data pointer, element count, projection data pointer or 0, M, K, dsub);
//   * a hit is only trusted when a 64-bit content hash of quantizers + projection still matches: a dropped
//     `Pq` whose allocation is reused, or centroids mutated in place during training, REPLACE the entry
//     (the old device image is destroyed) instead of returning stale codes;
//   * at most `capacity` entries, least recently used evicted and destroyed -- no unbounded device memory.
// The handle type and its create / destroy functions are template parameters: the product instantiates it
// with pqhip_codebook*, the CPU unit test with counters.
#pragma once
#include <cstddef>
#include <cstdint>
#include <list>
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

    CodebookCache(size_t capacity, Create create, Destroy destroy, void* user)
        : cap_(capacity ? capacity : 1), create_(create), destroy_(destroy), user_(user) {}
    ~CodebookCache() { clear(); }
    CodebookCache(const CodebookCache&) = delete;
    CodebookCache& operator=(const CodebookCache&) = delete;

    // Device image of (quantizers [M][K][dsub], projection [d][d] or nullptr).  Returns Handle() (null) when
    // creation fails; the caller then stays on its CPU path.  The handle stays valid until it is evicted:
    // hold the returned handle only for the duration of one call, under `lock()`.
    Handle get(const float* q, int64_t M, int64_t K, int64_t dsub, const float* p)
    {
        const size_t q_len = (size_t)(M * K * dsub), d = (size_t)(M * dsub);
        const Key key{q, q_len, p, M, K, dsub};
        uint64_t h = content_hash(q, q_len * sizeof(float));
        if (p) h = content_hash(p, d * d * sizeof(float), h);
        for (auto it = entries_.begin(); it != entries_.end(); ++it) {
            if (!(it->key == key)) continue;
            if (it->hash == h) {                       // same memory, same contents: reuse, mark most recent
                entries_.splice(entries_.begin(), entries_, it);
                ++hits_;
                return entries_.front().handle;
            }
            destroy_(user_, it->handle);               // same address, other contents: stale image
            entries_.erase(it);
            ++replaced_;
            break;
        }
        Handle nh = create_(user_, q, M, K, dsub, p);
        if (!nh) return Handle();
        entries_.push_front(Entry{key, h, nh});
        ++created_;
        while (entries_.size() > cap_) {
            destroy_(user_, entries_.back().handle);
            entries_.pop_back();
            ++evicted_;
        }
        return nh;
    }
    void clear()
    {
        for (auto& e : entries_) destroy_(user_, e.handle);
        entries_.clear();
    }
    std::mutex& lock() { return mu_; }
    size_t size() const { return entries_.size(); }
    size_t hits() const { return hits_; }
    size_t created() const { return created_; }
    size_t replaced() const { return replaced_; }
    size_t evicted() const { return evicted_; }

private:
    struct Entry { Key key; uint64_t hash; Handle handle; };
    std::list<Entry> entries_;
    size_t cap_;
    Create create_;
    Destroy destroy_;
    void* user_;
    std::mutex mu_;
    size_t hits_ = 0, created_ = 0, replaced_ = 0, evicted_ = 0;
};

}  // namespace reductive_amd
