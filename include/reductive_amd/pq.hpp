// reductive_amd/pq.hpp -- C++ host-side mirror of reductive's `Pq<f32>` over the C ABI (pqhip.h).
//
// The reference is compiled code (Rust) and its toolchain is absent from this image, so the host
// side above the C ABI is written in C++ with the reference's names, argument meaning and error
// behaviour:
//   Pq::new                      src/pq/pq.rs:38-61
//   n_quantizer_centroids        src/pq/pq.rs:103-105
//   projection / subquantizers   src/pq/pq.rs:108-110, 191-193
//   QuantizeVector::{quantize_batch, quantize_batch_into, quantize_vector, quantized_len}
//                                src/pq/traits.rs:75-99,  src/pq/pq.rs:252-303
//   Reconstruct::{reconstruct_batch, reconstruct_batch_into, reconstruct, reconstructed_len}
//                                src/pq/traits.rs:102-156, src/pq/pq.rs:305-348
// A Rust `panic!` becomes a thrown reductive_amd::Panic carrying the reference's message.
// The batch methods run on the GPU through libpqhip.so and have no CPU fallback; the
// single-vector methods are the reference's latency path and stay on the host.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../pqhip.h"

namespace reductive_amd {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HipError : std::runtime_error {
    int32_t status;
    HipError(int32_t s, const std::string& what)
        : std::runtime_error(std::string("pqhip: ") + pqhip_strerror(s) + " (" + what + ")" +
                             (s == PQHIP_EHIP ? std::string(" [") + pqhip_last_hip_error() + "]" : "")),
          status(s) {}
};

// ndarray-style 2-D views (strides in elements)
template <typename T>
struct View2 {
    T* ptr;
    int64_t rows, cols, row_stride, col_stride;
    View2(T* p, int64_t r, int64_t c) : ptr(p), rows(r), cols(c), row_stride(c), col_stride(1) {}
    View2(T* p, int64_t r, int64_t c, int64_t rs, int64_t cs)
        : ptr(p), rows(r), cols(c), row_stride(rs), col_stride(cs) {}
};

class Context {
public:
    explicit Context(const std::vector<int32_t>& devices = {})
    {
        int32_t rc = pqhip_ctx_create(devices.empty() ? nullptr : devices.data(), (int32_t)devices.size(), &h_);
        if (rc != PQHIP_OK) throw HipError(rc, "pqhip_ctx_create");
    }
    ~Context() { pqhip_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    pqhip_ctx* handle() const { return h_; }
    int n_devices() const { return pqhip_ctx_n_devices(h_); }

private:
    pqhip_ctx* h_ = nullptr;
};

class Pq {
public:
    // Pq::new(projection, quantizers); quantizers is [M][K][dsub] C order, projection [d][d]
    Pq(std::optional<std::vector<float>> projection, std::vector<float> quantizers, int64_t M, int64_t K,
       int64_t dsub, int64_t proj_rows = -1, int64_t proj_cols = -1)
        : proj_(std::move(projection)), q_(std::move(quantizers)), M_(M), K_(K), dsub_(dsub)
    {
        if (M <= 0 || K <= 0 || dsub <= 0 || q_.empty() || (int64_t)q_.size() != M * K * dsub)
            throw Panic("Attempted to construct a product quantizer without quantizers.");
        const int64_t rl = M * dsub;
        if (proj_) {
            if (proj_rows < 0) proj_rows = proj_cols = (int64_t)std::llround(std::sqrt((double)proj_->size()));
            if (proj_rows != rl || proj_cols != rl || (int64_t)proj_->size() != rl * rl)
                throw Panic("Incorrect projection matrix shape, was: [" + std::to_string(proj_rows) + ", " +
                            std::to_string(proj_cols) + "], should be [" + std::to_string(rl) + ", " +
                            std::to_string(rl) + "]");
        }
    }
    ~Pq() { pqhip_codebook_destroy(cb_); }
    Pq(const Pq& o) : proj_(o.proj_), q_(o.q_), M_(o.M_), K_(o.K_), dsub_(o.dsub_) {}  // Clone: deep copy, handle re-created lazily
    bool operator==(const Pq& o) const { return proj_ == o.proj_ && q_ == o.q_ && M_ == o.M_ && K_ == o.K_; }  // PartialEq

    int64_t n_quantizer_centroids() const { return K_; }
    const std::optional<std::vector<float>>& projection() const { return proj_; }
    const std::vector<float>& subquantizers() const { return q_; }
    int64_t quantized_len() const { return M_; }
    int64_t reconstructed_len() const { return M_ * dsub_; }

    void attach(Context* ctx) { ctx_ = ctx; }

    // ---- QuantizeVector ---------------------------------------------------------------------
    template <typename I>
    std::vector<I> quantize_batch(View2<const float> x)
    {
        std::vector<I> out((size_t)(x.rows * M_), I(0));
        quantize_batch_into<I>(x, View2<I>(out.data(), x.rows, M_));
        return out;
    }

    template <typename I>
    void quantize_batch_into(View2<const float> x, View2<I> quantized)
    {
        if (x.cols != reconstructed_len()) throw Panic("Quantizer and vector length mismatch");
        if (quantized.rows != x.rows || quantized.cols != M_)
            throw Panic("Quantized matrix has incorrect shape, expected: (" + std::to_string(x.rows) + ", " +
                        std::to_string(M_) + "), got: (" + std::to_string(quantized.rows) + ", " +
                        std::to_string(quantized.cols) + ")");
        int32_t rc = pqhip_quantize_batch_f32(cb(), x.ptr, x.rows, x.row_stride, x.col_stride, quantized.ptr,
                                              (int32_t)sizeof(I), quantized.row_stride, quantized.col_stride);
        if (rc == PQHIP_EINDEX_WIDTH) throw Panic("Cannot store centroids in quantizer index type");
        if (rc != PQHIP_OK) throw HipError(rc, "pqhip_quantize_batch_f32");
    }

    // pq.rs:285-298 -> primitives.rs:14-49 -> kmeans.rs:111-126 -> linalg.rs:118-148 (host)
    template <typename I>
    std::vector<I> quantize_vector(const float* x, int64_t len) const
    {
        if (len != reconstructed_len()) throw Panic("Quantizer and vector length mismatch");
        if ((uint64_t)(K_ - 1) > (uint64_t)std::numeric_limits<I>::max())
            throw Panic("Cannot store centroids in quantizer index type");
        const int64_t d = reconstructed_len();
        std::vector<float> rx(x, x + d);
        if (proj_) {  // 1-D x 2-D ndarray dot without BLAS: per column a sequential s = s + x[k]*P[k][c]
            for (int64_t c = 0; c < d; ++c) {
                float s = 0.f;
                for (int64_t k = 0; k < d; ++k) { const float p = x[k] * (*proj_)[k * d + c]; s = s + p; }
                rx[c] = s;
            }
        }
        std::vector<I> out((size_t)M_);
        std::vector<float> dist((size_t)K_);
        for (int64_t m = 0; m < M_; ++m) {
            const float* xs = rx.data() + m * dsub_;
            const float xx = unrolled_dot(xs, xs, dsub_);
            for (int64_t j = 0; j < K_; ++j) {
                const float* c = q_.data() + (m * K_ + j) * dsub_;
                const float cc = unrolled_dot(c, c, dsub_), dp = unrolled_dot(c, xs, dsub_);
                const float t = xx + cc, u = dp + dp;
                dist[j] = t - u;
            }
            int64_t best = 0;
            for (int64_t j = 1; j < K_; ++j)
                if (of_less(dist[j], dist[best])) best = j;
            out[m] = (I)best;
        }
        return out;
    }

    // ---- Reconstruct --------------------------------------------------------------------------
    template <typename I>
    std::vector<float> reconstruct_batch(View2<const I> quantized)
    {
        std::vector<float> out((size_t)(quantized.rows * reconstructed_len()), 0.f);
        reconstruct_batch_into<I>(quantized, View2<float>(out.data(), quantized.rows, reconstructed_len()));
        return out;
    }

    template <typename I>
    void reconstruct_batch_into(View2<const I> quantized, View2<float> rec)
    {
        if (rec.rows != quantized.rows || rec.cols != reconstructed_len())
            throw Panic("Reconstructions matrix has incorrect shape, expected: (" + std::to_string(quantized.rows) +
                        ", " + std::to_string(reconstructed_len()) + "), got: (" + std::to_string(rec.rows) + ", " +
                        std::to_string(rec.cols) + ")");
        if (quantized.cols != M_) throw Panic("Quantization length does not match number of subquantizers");
        int32_t rc = pqhip_reconstruct_batch_f32(cb(), quantized.ptr, (int32_t)sizeof(I), quantized.rows,
                                                 quantized.row_stride, quantized.col_stride, rec.ptr,
                                                 rec.row_stride, rec.col_stride);
        if (rc == PQHIP_ECODE_RANGE) throw Panic("ndarray: index out of bounds");
        if (rc != PQHIP_OK) throw HipError(rc, "pqhip_reconstruct_batch_f32");
    }

    template <typename I>
    std::vector<float> reconstruct(const I* quantized, int64_t len) const
    {
        if (len != M_) throw Panic("Quantization length does not match number of subquantizers");
        const int64_t d = reconstructed_len();
        std::vector<float> rec((size_t)d);
        for (int64_t m = 0; m < M_; ++m) {
            if ((uint64_t)quantized[m] >= (uint64_t)K_) throw Panic("ndarray: index out of bounds");
            std::memcpy(rec.data() + m * dsub_, q_.data() + (m * K_ + (int64_t)quantized[m]) * dsub_,
                        sizeof(float) * (size_t)dsub_);
        }
        if (proj_) {  // reconstruction.dot(&projection.t()): per output k a contiguous dot with P[k,:]
            std::vector<float> out((size_t)d);
            for (int64_t k = 0; k < d; ++k) out[k] = unrolled_dot(proj_->data() + k * d, rec.data(), d);
            return out;
        }
        return rec;
    }

private:
    static float unrolled_dot(const float* x, const float* y, int64_t n)  // ndarray numeric_util::unrolled_dot
    {
        float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t i = 0;
        for (; n - i >= 8; i += 8)
            for (int l = 0; l < 8; ++l) { const float pr = x[i + l] * y[i + l]; p[l] = p[l] + pr; }
        float s = 0.f;
        s = s + (p[0] + p[4]); s = s + (p[1] + p[5]); s = s + (p[2] + p[6]); s = s + (p[3] + p[7]);
        for (; i < n; ++i) { const float pr = x[i] * y[i]; s = s + pr; }
        return s;
    }
    static bool of_less(float a, float b)  // ordered-float: NaN greatest
    {
        if (std::isnan(a)) return false;
        if (std::isnan(b)) return true;
        return a < b;
    }
    pqhip_codebook* cb()
    {
        if (!cb_) {
            if (!ctx_) { own_ctx_.reset(new Context()); ctx_ = own_ctx_.get(); }
            int32_t rc = pqhip_codebook_create(ctx_->handle(), q_.data(), M_, K_, dsub_,
                                               proj_ ? proj_->data() : nullptr, &cb_);
            if (rc != PQHIP_OK) throw HipError(rc, "pqhip_codebook_create");
        }
        return cb_;
    }

    std::optional<std::vector<float>> proj_;
    std::vector<float> q_;
    int64_t M_, K_, dsub_;
    Context* ctx_ = nullptr;
    std::unique_ptr<Context> own_ctx_;
    pqhip_codebook* cb_ = nullptr;
};

// ---- "next" row: the k-means step of training ------------------------------------------------------
// `n_iterations` x kmeans_iteration (src/kmeans.rs:308-327) on every subquantizer's column block:
// the body of `kmeans_with_centroids(.., NIterationsCondition(n))` at pq.rs:176 for all
// subquantizers, and of `Opq::update_subquantizers` (opq.rs:227-245) for n_iterations == 1.
// `quantizers` ([M][K][dsub], C order) is updated in place; returns the last iteration's mean
// squared error of every subquantizer (empty when want_loss is false).
inline std::vector<float> kmeans_iterations(Context& ctx, std::vector<float>& quantizers, int64_t M, int64_t K,
                                            int64_t dsub, View2<const float> instances, int n_iterations,
                                            bool want_loss = true)
{
    if (K == 0 || quantizers.empty())
        throw Panic("Cannot cluster instances with zero centroids.");                     // kmeans.rs:260-263
    if ((int64_t)quantizers.size() != M * K * dsub || instances.cols != M * dsub)
        throw Panic("Centroid and instance lengths differ.");                             // kmeans.rs:264-268
    std::vector<float> loss(want_loss ? (size_t)M : 0);
    const int32_t rc = pqhip_kmeans_iterations_f32(ctx.handle(), quantizers.data(), M, K, dsub, instances.ptr,
                                                   instances.rows, instances.row_stride, instances.col_stride,
                                                   n_iterations, want_loss ? loss.data() : nullptr);
    if (rc != PQHIP_OK) throw HipError(rc, "pqhip_kmeans_iterations_f32");
    return loss;
}

// Instances kept in HBM for a training run (pqhip_matrix_*): upload once, iterate many times.
class ResidentMatrix {
public:
    ResidentMatrix(Context& ctx, View2<const float> x, int32_t device_slot = 0) : ctx_(ctx), slot_(device_slot), cols_(x.cols)
    {
        const int32_t rc = pqhip_matrix_upload_f32(ctx.handle(), device_slot, x.ptr, x.rows, x.cols, x.row_stride,
                                                   x.col_stride, &m_);
        if (rc != PQHIP_OK) throw HipError(rc, "pqhip_matrix_upload_f32");
    }
    ~ResidentMatrix() { pqhip_matrix_destroy(m_); }
    ResidentMatrix(const ResidentMatrix&) = delete;
    ResidentMatrix& operator=(const ResidentMatrix&) = delete;
    const float* device_ptr() const { return pqhip_matrix_device_ptr(m_); }
    int64_t rows() const { return pqhip_matrix_rows(m_); }
    int64_t cols() const { return cols_; }
    Context& context() const { return ctx_; }
    int32_t slot() const { return slot_; }

private:
    Context& ctx_;
    int32_t slot_;
    int64_t cols_;
    pqhip_matrix* m_ = nullptr;
};

// The device part of `Opq::train_iteration` (src/pq/opq.rs:156-195): rotation, k-means update,
// quantize -> reconstruct round trip, `instances.t().dot(&reconstructed)`.  `quantizers` is updated in
// place; returns cross [d][d]; the caller finishes with `svd(cross)` and `projection = u.dot(vt)`.
inline std::vector<float> opq_train_step(const ResidentMatrix& instances, std::vector<float>& quantizers, int64_t M,
                                         int64_t K, int64_t dsub, const std::vector<float>& projection)
{
    const int64_t d = M * dsub;
    if ((int64_t)projection.size() != d * d)
        throw Panic("Incorrect projection matrix shape, was: [" + std::to_string(projection.size()) + " elements], should be [" +
                    std::to_string(d) + ", " + std::to_string(d) + "]");
    if ((int64_t)quantizers.size() != M * K * dsub || instances.cols() != d)
        throw Panic("Centroid and instance lengths differ.");
    std::vector<float> cross((size_t)(d * d));
    const int32_t rc = pqhip_opq_train_step_f32_dev(instances.context().handle(), instances.slot(), quantizers.data(), M, K,
                                                    dsub, projection.data(), instances.device_ptr(), instances.rows(), d,
                                                    cross.data(), nullptr);
    if (rc != PQHIP_OK) throw HipError(rc, "pqhip_opq_train_step_f32_dev");
    return cross;
}

}  // namespace reductive_amd
