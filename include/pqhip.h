/*
 * pqhip.h -- C ABI of libpqhip.so: MI355X (gfx950) product-quantizer encode / reconstruct.
 *
 * This is the drop-in boundary for ONE hot path of finalfusion/reductive v0.9.0: the bodies of
 *   QuantizeVector::quantize_batch_into   src/pq/pq.rs:268-283  (-> src/pq/primitives.rs:64-104,
 *                                          src/kmeans.rs:133-159, src/linalg.rs:150-180)
 *   Reconstruct::reconstruct_batch_into   src/pq/pq.rs:309-327  (-> src/pq/primitives.rs:110-173)
 * for A = f32.  The reference has no FFI of its own (it is a pure-Rust crate); the entry points
 * below are exactly what a `extern "C"` block inside `impl QuantizeVector<f32> for Pq<f32>` binds
 * (binding shown in INTEGRATION.md and rust/pqhip_ffi.rs).  Plain pointers and sizes only.
 *
 * Conventions
 *  - every function returns a pqhip_status (0 = OK); nothing throws, nothing aborts;
 *  - strides are in ELEMENTS (ndarray convention), may be any non-negative value for host entry
 *    points; device entry points need unit column stride;
 *  - host pointers are only read/written during the call; handles own their device memory;
 *  - all entry points are re-entrant and may be called concurrently from many threads
 *    (`Pq<f32>` is `Send + Sync`; reference hot path is `&self`);
 *  - results: u8/u16/u32 codes are bit-identical to the CANON-F32 arithmetic declared in
 *    DESIGN.md (first index wins ties, NaN ordered last as ordered-float does); PQ
 *    reconstructions are exact copies of codebook rows; OPQ reconstructions follow the same
 *    chain rule and agree with the reference within 1e-5 relative.
 *  - there is NO CPU fallback in this library: without a usable gfx950 device every compute
 *    entry point returns PQHIP_ENODEV.
 */
#ifndef PQHIP_H
#define PQHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQHIP_VERSION 100 /* 0.1.0 */

typedef enum pqhip_status {
    PQHIP_OK = 0,
    PQHIP_EINVAL = 1,       /* null pointer / negative size / bad argument                     */
    PQHIP_ESHAPE = 2,       /* "Quantizer and vector length mismatch" (primitives.rs:74-78),
                               output-shape asserts (primitives.rs:80-87, 159-167),
                               projection shape (pq.rs:46-55), empty quantizers (pq.rs:39-42)  */
    PQHIP_ECODE_RANGE = 3,  /* a code >= K in reconstruct (reference: ndarray index_axis panic,
                               primitives.rs:146)                                              */
    PQHIP_EINDEX_WIDTH = 4, /* K-1 does not fit the index type (primitives.rs:31-34)           */
    PQHIP_ENODEV = 5,       /* no gfx950 device / device index out of range                    */
    PQHIP_EHIP = 6,         /* a HIP runtime call failed (pqhip_last_hip_error() has the text) */
    PQHIP_ENOMEM = 7,       /* host or device allocation failed                                */
    PQHIP_EUNSUPPORTED = 8  /* valid request this build has no kernel for                      */
} pqhip_status;

typedef struct pqhip_ctx pqhip_ctx;           /* a set of devices + per-device staging/streams */
typedef struct pqhip_codebook pqhip_codebook; /* device-resident image of one `Pq<f32>`         */

int32_t pqhip_version(void);
const char *pqhip_strerror(int32_t status);
/* text of the last failing HIP call on this thread ("" if none) */
const char *pqhip_last_hip_error(void);

/* number of visible HIP devices (0 and PQHIP_ENODEV if none) */
int32_t pqhip_device_count(int32_t *out_count);

/* devices == NULL / n_devices == 0  ->  all visible devices. */
int32_t pqhip_ctx_create(const int32_t *devices, int32_t n_devices, pqhip_ctx **out);
void pqhip_ctx_destroy(pqhip_ctx *ctx);
int32_t pqhip_ctx_n_devices(const pqhip_ctx *ctx);

/*
 * Upload one product quantizer (replaces `Pq::new`, pq.rs:38-61, on the device side).
 *   quantizers : [M][K][dsub] f32, C order          (Pq.quantizers, pq.rs:31)
 *   projection : [d][d] f32 row-major, d = M*dsub, applied as x.dot(P) (pq.rs:276); NULL = plain PQ
 * The codebook is replicated on every device of the ctx (<= 3 MB), no collective involved.
 */
int32_t pqhip_codebook_create(pqhip_ctx *ctx, const float *quantizers, int64_t n_subquantizers,
                              int64_t n_centroids, int64_t sub_dim, const float *projection,
                              pqhip_codebook **out);
void pqhip_codebook_destroy(pqhip_codebook *cb);
int64_t pqhip_codebook_quantized_len(const pqhip_codebook *cb);     /* M      pq.rs:300-302 */
int64_t pqhip_codebook_reconstructed_len(const pqhip_codebook *cb); /* M*dsub pq.rs:345-347 */
int64_t pqhip_codebook_n_centroids(const pqhip_codebook *cb);       /* K      pq.rs:103-105 */
int32_t pqhip_codebook_has_projection(const pqhip_codebook *cb);

/*
 * HOST-buffer entry points: what `Pq::quantize_batch_into` / `reconstruct_batch_into` call.
 * Rows are sharded contiguously over the ctx's devices, streamed through
 * pinned staging buffers, results land in the caller's strided buffer.  code_bytes is
 * sizeof(I) for the Rust index type I: 1 (u8), 2 (u16), 4 (u32) or 8 (usize/u64).
 */
int32_t pqhip_quantize_batch_f32(pqhip_codebook *cb, const float *x, int64_t n_rows,
                                 int64_t x_row_stride, int64_t x_col_stride, void *codes,
                                 int32_t code_bytes, int64_t codes_row_stride,
                                 int64_t codes_col_stride);

int32_t pqhip_reconstruct_batch_f32(pqhip_codebook *cb, const void *codes, int32_t code_bytes,
                                    int64_t n_rows, int64_t codes_row_stride,
                                    int64_t codes_col_stride, float *out, int64_t out_row_stride,
                                    int64_t out_col_stride);

/*
 * DEVICE-resident entry points (inputs/outputs already in the HBM of device `device_slot` of the
 * ctx; unit column stride; row strides in elements).  Asynchronous on `stream` (a hipStream_t
 * passed as void*, NULL = the device's default stream); the caller synchronises.  code_bytes in
 * {1, 2, 4, 8} as for the host entry points (traits.rs:77-88 is generic over the index type): the
 * kernels produce / consume u8 and u32 codes, 2- and 8-byte matrices are converted on the device
 * through a leased scratch matrix (the row-lookup, ADC and k-means entry points take 1- and 4-byte
 * codes only).  Scratch for the OPQ variants is managed inside the codebook handle.
 */
int32_t pqhip_quantize_batch_f32_dev(pqhip_codebook *cb, int32_t device_slot, const float *d_x,
                                     int64_t n_rows, int64_t x_row_stride, void *d_codes,
                                     int32_t code_bytes, int64_t codes_row_stride, void *stream);

int32_t pqhip_reconstruct_batch_f32_dev(pqhip_codebook *cb, int32_t device_slot,
                                        const void *d_codes, int32_t code_bytes, int64_t n_rows,
                                        int64_t codes_row_stride, float *d_out,
                                        int64_t out_row_stride, void *stream);

/*
 * "Next" row: the lookup path of a consumer that keeps a quantized
 * matrix resident (finalfusion's quantized embedding storage is the out-of-tree example): row
 * select, `Reconstruct::reconstruct_batch` (src/pq/traits.rs:109-117 -> src/pq/pq.rs:309-327) of the
 * selected code rows, and an optional per-row rescale, fused into one pass over HBM:
 *     out[i][:] = reconstruct(codes[rows[i]][:]) * scales[rows[i]]        (scales == NULL: no rescale)
 * d_codes [n_codes][M] (1- or 4-byte codes, row stride in elements), d_rows [n] int64 and d_scales
 * [n_codes] f32 live in HBM; out [n][d].  With a projection the un-rotation (pq.rs:323-326) runs
 * before the rescale.  A row index outside [0, n_codes) is ndarray's `select` panic: it raises the
 * same asynchronous flag as a code >= K (query with pqhip_check_codes_dev -> PQHIP_ECODE_RANGE).
 * The multiply is one rounded f32 multiply per element, so results equal
 * `reconstruct_batch(codes.select(Axis(0), rows)) * scales.select(rows)` bit for bit.
 */
int32_t pqhip_reconstruct_rows_f32_dev(pqhip_codebook *cb, int32_t device_slot, const void *d_codes,
                                       int32_t code_bytes, int64_t n_codes, int64_t codes_row_stride,
                                       const int64_t *d_rows, int64_t n_rows, const float *d_scales,
                                       float *d_out, int64_t out_row_stride, void *stream);

/*
 * The same lookup over INTERLEAVED records: a resident matrix of n_codes records of `record_bytes` bytes, each
 * holding the M codes of a row at offset 0 and its f32 scale at `scale_offset_bytes` (both multiples of 4 bytes;
 * `record_bytes` also a multiple of `code_bytes`).  With 32-byte records at M = 15 (15 code bytes, 1 pad, the scale
 * at offset 16, 12 pad) a lookup touches ONE 128-byte line where the split layout above touches one for the 15-byte
 * code row (12 % of which straddle two) and one for the scale -- the layout `reductive_amd.Pq.interleave_records`
 * builds and `Pq.reconstruct_records_device` consumes.  Semantics and results are those of
 * pqhip_reconstruct_rows_f32_dev with d_scales given.
 */
int32_t pqhip_reconstruct_rows_records_f32_dev(pqhip_codebook *cb, int32_t device_slot, const void *d_records,
                                               int32_t code_bytes, int64_t n_codes, int64_t record_bytes,
                                               int64_t scale_offset_bytes, const int64_t *d_rows, int64_t n_rows,
                                               float *d_out, int64_t out_row_stride, void *stream);

/*
 * "Next" row: asymmetric distance computation -- the scan that follows
 * encode in a PQ pipeline, over a code matrix kept resident in HBM.  Not a function of reductive; it
 * is defined from the reference's own vector-to-matrix distance so that it has an exact meaning:
 *   tables[q][m][j] = y_q[m*dsub .. (m+1)*dsub).squared_euclidean_distance(quantizers[m])[j]
 *                     (src/linalg.rs:118-148 -- the distances `kmeans::cluster_assignment`,
 *                     kmeans.rs:111-126, minimises inside `Pq::quantize_vector`, pq.rs:285-298), with
 *                     y_q = query_q.dot(projection) first for an OPQ quantizer (pq.rs:293);
 *                     hence argmin_j tables[q][m][j] == quantize_vector(query_q)[m];
 *   out[q][i]       = sum over m = 0 .. M-1, in that order, from +0, of tables[q][m][codes[i][m]]
 *                     (f32 adds; the estimate of |query_q - reconstruct(codes[i])|^2 for plain PQ).
 * d_queries [n_queries][q_row_stride] f32 and d_tables [n_queries][M][K] f32 live in HBM; d_codes
 * [n_codes][codes_row_stride] are 1- or 4-byte codes; d_out [n_queries][out_row_stride] f32.
 * A code >= K raises the stream's range flag (pqhip_check_codes_dev -> PQHIP_ECODE_RANGE).
 * Both calls are asynchronous on `stream`.
 */
int32_t pqhip_adc_tables_f32_dev(pqhip_codebook *cb, int32_t device_slot, const float *d_queries,
                                 int64_t n_queries, int64_t q_row_stride, float *d_tables, void *stream);
int32_t pqhip_adc_scan_f32_dev(pqhip_codebook *cb, int32_t device_slot, const float *d_tables,
                               int64_t n_queries, const void *d_codes, int32_t code_bytes, int64_t n_codes,
                               int64_t codes_row_stride, float *d_out, int64_t out_row_stride, void *stream);

/* Reconstruct's range check is asynchronous on the device path: returns PQHIP_ECODE_RANGE if any
 * device call since the last query saw a code >= K (synchronises `stream`). */
int32_t pqhip_check_codes_dev(pqhip_codebook *cb, int32_t device_slot, void *stream);

/*
 * "Next" row of the hot path: the assignment step of k-means training,
 * `kmeans::cluster_assignments(centroids, instances, Axis(0))` (src/kmeans.rs:133-159, called from
 * kmeans.rs:319 and through primitives::quantize_batch::<_, usize, _> at opq.rs:180).  Same kernels
 * as PQ encode with one subquantizer; out[i] = index of the nearest of the K centroids [K][dim] for
 * row i of x, as out_bytes-wide unsigned integers (8 = usize).  Host buffers, element strides.
 */
int32_t pqhip_cluster_assignments_f32(pqhip_ctx *ctx, const float *centroids, int64_t n_centroids,
                                      int64_t dim, const float *x, int64_t n_rows,
                                      int64_t x_row_stride, int64_t x_col_stride, void *out,
                                      int32_t out_bytes);

/*
 * The whole k-means step of PQ/OPQ training, for all M subquantizers at once: `n_iterations` times
 * `kmeans_iteration` (src/kmeans.rs:308-327 = cluster_assignments :133-159, update_centroids
 * :166-198, mean_squared_error :329-360) on every subquantizer's column block of x.  It replaces
 *   - `sq_instances.kmeans_with_centroids(Axis(0), quantizer, NIterationsCondition(n_iterations))`
 *     (src/pq/pq.rs:176; kmeans.rs:270-279) for every subquantizer of `train_pq_using`
 *     (pq.rs:214-241), with the initial centroids of pq.rs:166-172 passed in;
 *   - `Opq::update_subquantizers` (src/pq/opq.rs:227-245) with n_iterations = 1 and loss = NULL
 *     (x = the rotated instances);
 *   - a plain `kmeans_iteration` / `kmeans_with_centroids` with M = 1, dsub = dim.
 * quantizers [M][K][dsub] (host, C order) holds the initial centroids and receives the updated
 * ones; loss (host, [M], may be NULL) receives the LAST iteration's mean squared error of every
 * subquantizer.  Results are bit-identical to the reference's sequential f32 arithmetic (sums in
 * row order, f32 counts, IEEE division, one sequential fold for the loss); empty clusters become
 * zero vectors as in kmeans.rs:180-197.  The instances stay resident on one device for all
 * iterations (host entry point: the first device of the context).  Limits: K <= 16384,
 * n_rows <= 2^31; sub-vectors wider than 256 floats use the slow anchor kernel for the assignment
 * step.  Both calls return synchronised.
 */
int32_t pqhip_kmeans_iterations_f32(pqhip_ctx *ctx, float *quantizers, int64_t n_subquantizers,
                                    int64_t n_centroids, int64_t sub_dim, const float *x,
                                    int64_t n_rows, int64_t x_row_stride, int64_t x_col_stride,
                                    int32_t n_iterations, float *loss);

/* same, instances already in HBM on `device_slot` (unit column stride) */
int32_t pqhip_kmeans_iterations_f32_dev(pqhip_ctx *ctx, int32_t device_slot, float *quantizers,
                                        int64_t n_subquantizers, int64_t n_centroids,
                                        int64_t sub_dim, const float *d_x, int64_t n_rows,
                                        int64_t x_row_stride, int32_t n_iterations, float *loss,
                                        void *stream);

/*
 * Resident instance matrices for the training entry points (which iterate over the same rows many
 * times): a row-major copy [n_rows][n_cols] of a host matrix (any element strides) in the HBM of
 * `device_slot`.  pqhip_matrix_device_ptr() is what the *_dev entry points take as d_x (row stride
 * n_cols).  A caller that has no device-memory management of its own (the Rust binding) uploads
 * once per training run.
 */
typedef struct pqhip_matrix pqhip_matrix;
int32_t pqhip_matrix_upload_f32(pqhip_ctx *ctx, int32_t device_slot, const float *x, int64_t n_rows,
                                int64_t n_cols, int64_t x_row_stride, int64_t x_col_stride,
                                pqhip_matrix **out);
const float *pqhip_matrix_device_ptr(const pqhip_matrix *m);
int64_t pqhip_matrix_rows(const pqhip_matrix *m);
void pqhip_matrix_destroy(pqhip_matrix *m);

/*
 * The device part of `Opq::train_iteration` (src/pq/opq.rs:156-195), i.e. everything of an OPQ
 * training iteration except its LAPACK call:
 *   rx = instances.dot(&projection)                                   (opq.rs:167)
 *   update_subquantizers(centroids, rx)   -- one kmeans_iteration per subquantizer  (:168, :227-245)
 *   quantized = quantize_batch::<usize>(centroids, rx); reconstruct_batch_into(centroids, quantized, rx)
 *                                                                     (:176-182, no projection)
 *   cross = instances.t().dot(&reconstructed)                         (first half of :191)
 * quantizers [M][K][dsub] (host, in/out), projection [d][d] (host, in), instances d_x [n][d] resident
 * in HBM on `device_slot`; cross [d][d] (host, out).  The caller finishes the iteration with
 * `(u, _, vt) = cross.svd(); projection = u.dot(vt)` (opq.rs:191-192).  Every step follows the
 * arithmetic rules of DESIGN.md section 3 (the cross product is rule 2 with k over the rows: chains
 * restart every 256 rows, block results added in row order), so quantizers and cross are
 * bit-identical to the reference's default build for the same projection.  Returns synchronised.
 */
int32_t pqhip_opq_train_step_f32_dev(pqhip_ctx *ctx, int32_t device_slot, float *quantizers,
                                     int64_t n_subquantizers, int64_t n_centroids, int64_t sub_dim,
                                     const float *projection, const float *d_x, int64_t n_rows,
                                     int64_t x_row_stride, float *cross, void *stream);

/* d_out [n][d] = d_x [n][d] . projection [d][d] (host) on the device with rule-2 arithmetic:
 * `instances.dot(&projection)` of the training paths (opq.rs:62, gaussian_opq.rs:55).  Returns synchronised. */
int32_t pqhip_rotate_f32_dev(pqhip_ctx *ctx, int32_t device_slot, const float *d_x, int64_t n_rows,
                             int64_t x_row_stride, int64_t d, const float *projection, float *d_out,
                             int64_t out_row_stride, void *stream);

/* out [da][db] (host) = a^T . b for device-resident a [n][da], b [n][db] (unit column strides) with
 * the same rule-2 arithmetic: `a.t().dot(&b)` of ndarray (opq.rs:191). */
int32_t pqhip_at_dot_b_f32_dev(pqhip_ctx *ctx, int32_t device_slot, const float *d_a,
                               int64_t a_row_stride, int64_t da, const float *d_b,
                               int64_t b_row_stride, int64_t db, int64_t n_rows, float *out,
                               void *stream);

/* ---- knobs used by the test-suite and the bench (not part of the reference surface) --------
 * The library reads three environment variables (PQHIP_PACK_THREADS, PQHIP_HOST_ZERO_COPY, PQHIP_FUSED2_OPQ: all
 * about the host path / deployment); every other switch is an explicit call below.
 *
 * pqhip_set_encode_variant: force an encode kernel family on one codebook (PQHIP_EUNSUPPORTED from the next
 * quantize call when the family has no instantiation for the shape):
 *   0  auto
 *   1  scalar anchor kernel (exact by construction, any shape)
 *   2  MFMA 32x32x2 with a lane-local (VALU) argmin          auto for sub-vectors of <= 2 floats
 *   4  MFMA 32x32x2, LDS-atomic argmin, fragments in LDS     auto wherever 9 is not taken; K > 256; k-means
 *   6  VALU kernel for small codebooks (K <= 64, u8 codes)   auto for K <= 16, sub-vectors of <= 8 floats where 10 does not fit
 *   7  two subquantizers per matrix tile (K <= 16)           auto for 2 floats, and 4 floats from 48 subquantizers on
 *   8  OPQ only: rotation + encode in one kernel             auto where instantiated (opq_fused2_launch.h)
 *   9  MFMA 16x16x4, LDS-atomic argmin, four waves per SIMD  auto for K > 128 and sub-vectors of 12 .. 24 floats
 *  10  MFMA 16x16x4 for small codebooks (K <= 32, 4 / 8 / 12 / 16 / 20 / 24 / 32 floats, u8 codes, 16-byte aligned rows)   auto wherever it fits
 *  11  1- and 2-float sub-vectors, K <= 256: per-cell candidate lists (Pq handles with finite, in-range centroids)
 *      auto for K > 16, and for every K at 1 float
 *   (3 and 5 named kernels that rounds 1-2 shipped; PQHIP_EINVAL since round 4)                                  */
int32_t pqhip_set_encode_variant(pqhip_codebook *cb, int32_t variant);
/* process-wide: the P-block rotation kernel where both exist (16-byte aligned rows, d % 4 == 0):
 *   0  auto: k_rotate_pblock9 (16x16x4) for the gather form (OPQ reconstruct), for d > 640, and for plain rotation
 *      where 64-column blocks would execute >= 10 % more columns than 16-column tiles (d = 96, 144, 272, 400 ..);
 *      k_rotate_pblock8 (32x32x2) otherwise
 *   8 / 9  force one of them                                                                                    */
int32_t pqhip_set_rotation_variant(int32_t variant);
/* per-context options (value >= 0; PQHIP_EINVAL for an unknown name):
 *   "kmeans_window_rows"   rows per window of the k-means iteration (0 = default, 512 K)
 *   "kmeans_lane_form"     1 = lane-per-chain update walk for every shape
 *   "kmeans_no_graph"      1 = never replay small training sets as a captured hipGraph
 *   "opq_scratch_rows"     rows per chunk of the two-kernel OPQ paths (0 = whole rounds of the rotation grid)
 *   "opq_fused"            0 = OPQ encode as rotation -> scratch -> encode (default 1; PQHIP_FUSED2_OPQ=0 presets 0)
 *   "opq_gather_rotation"  0 = OPQ reconstruct as gather -> scratch -> rotation (default 1)
 *   "adc_single_query"     1 = one scan pass per query (default 0: 8 / 4 queries share a pass)
 *   "cross_product_exact"  0 = X^T.R of the OPQ training step / pqhip_at_dot_b_f32_dev as a plain split-K product:
 *                          within 1e-5 relative of the exact rule-2 result, no per-block partial matrices (default 1)
 *   "cross_product_group_bytes"  workspace of partial matrices per launch group (0 = 4 GiB)
 *   "lookup_two_pass"      row lookups (pqhip_reconstruct_rows*): 0 = one kernel, 1 = select the code rows into a compact
 *                          staging area first, 2 (default) = two passes when the resident matrix exceeds 256 MB
 *   "candidate_tables"     1 (default): pqhip_codebook_create builds the per-cell candidate tables of the 1- / 2-float sub-vector
 *                          encode kernel on the host (M = 150, K = 256: about 2.4 s on 8 cores, 34 ms at M = 10, K = 128 -- it pays
 *                          for itself after some 5e8 / 2.6e8 encoded rows); 0: handles created from now on get no tables and
 *                          encode on the kernels that evaluate every centroid (same codes)                                  */
int32_t pqhip_ctx_set_option(pqhip_ctx *ctx, const char *name, int64_t value);
/* Launch log of the calling thread: every kernel the library launches is noted by name; pqhip_launch_log() renders
 * "k_a + k_b x3 + ..." (distinct names in first-launch order with counts; valid until the thread's next call of
 * it), pqhip_launch_log_reset() clears it.  bench.py reports roofline.kernel from it.                            */
const char *pqhip_launch_log(void);
void pqhip_launch_log_reset(void);
/* Host only (no GPU needed): the candidate tables k_encode_vor2 uses for a codebook of 1- or 2-float sub-vectors, as built at
 * codebook creation (layout and the argument why the winner is always on a cell's list: reductive_amd/csrc/vor2_prep.h).
 * quantizers [M][K][dsub], dsub 1 or 2.  *n_words receives the number of 32-bit words; words_out (capacity words_cap, may be NULL to ask
 * for the size) the regions back to back, region_off_out [M + 1] their word offsets.  PQHIP_EUNSUPPORTED when the codebook
 * is not eligible (K > 256, non-finite or extreme centroids).  The CPU test-suite checks the tables against the oracle. */
int32_t pqhip_vor2_tables_host(const float *quantizers, int64_t M, int64_t K, int64_t dsub, uint32_t *words_out,
                               int64_t words_cap, uint32_t *region_off_out, int64_t *n_words);
/* name of the encode kernel the last device call on this codebook launched ("" if none)        */
const char *pqhip_last_encode_kernel(const pqhip_codebook *cb);

/* Self-test of the hardware property the MFMA path rests on: v_mfma_f32_32x32x2_f32 must equal
 * a k-ordered fmaf chain bit for bit.  Runs n_trials random 32x32xk tiles on device_slot;
 * *out_mismatches receives the number of differing elements. */
int32_t pqhip_selftest_mfma_chain(pqhip_ctx *ctx, int32_t device_slot, int32_t k, int32_t n_trials,
                                  uint64_t seed, int64_t *out_mismatches);

#ifdef __cplusplus
}
#endif
#endif /* PQHIP_H */
