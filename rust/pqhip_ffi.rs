//! pqhip_ffi.rs -- the reference-side binding of libpqhip.so (include/pqhip.h).
//!
//! SOURCE ONLY: this image has no rustc/cargo, so this file is never compiled in this
//! repository's pipeline; every call below is mirrored by the C++ (include/reductive_amd/pq.hpp)
//! and Python (reductive_amd/pq.py) host layers, which ARE executed by the test-suite.
//!
//! Drop into `reductive/src/pq/` behind a cargo feature `hip`, add `mod pqhip_ffi;` to
//! `src/pq/mod.rs`, and route the two hot-path methods of `impl QuantizeVector<A> for Pq<A>` /
//! `impl Reconstruct<A> for Pq<A>` (src/pq/pq.rs:268-283, 309-327) through `try_quantize_batch`
//! / `try_reconstruct_batch` (see INTEGRATION.md for the five-line patch of pq.rs).
#![cfg(feature = "hip")]
#![allow(non_camel_case_types)]

use std::any::TypeId;
use std::os::raw::{c_char, c_void};
use std::sync::atomic::{AtomicU64, Ordering};
use std::sync::{Arc, Mutex, OnceLock};

use ndarray::{ArrayBase, ArrayView2, ArrayView3, ArrayViewMut2, ArrayViewMut3, Data, Ix2, ShapeBuilder};

#[repr(C)]
pub struct pqhip_ctx { _p: [u8; 0] }
#[repr(C)]
pub struct pqhip_codebook { _p: [u8; 0] }
#[repr(C)]
pub struct pqhip_matrix { _p: [u8; 0] }

pub const PQHIP_OK: i32 = 0;
pub const PQHIP_ECODE_RANGE: i32 = 3;
pub const PQHIP_EINDEX_WIDTH: i32 = 4;

#[link(name = "pqhip")]
extern "C" {
    pub fn pqhip_version() -> i32;
    pub fn pqhip_strerror(status: i32) -> *const c_char;
    pub fn pqhip_device_count(out_count: *mut i32) -> i32;
    pub fn pqhip_ctx_create(devices: *const i32, n_devices: i32, out: *mut *mut pqhip_ctx) -> i32;
    pub fn pqhip_ctx_destroy(ctx: *mut pqhip_ctx);
    pub fn pqhip_codebook_create(ctx: *mut pqhip_ctx, quantizers: *const f32, n_subquantizers: i64,
        n_centroids: i64, sub_dim: i64, projection: *const f32, out: *mut *mut pqhip_codebook) -> i32;
    pub fn pqhip_codebook_destroy(cb: *mut pqhip_codebook);
    pub fn pqhip_quantize_batch_f32(cb: *mut pqhip_codebook, x: *const f32, n_rows: i64,
        x_row_stride: i64, x_col_stride: i64, codes: *mut c_void, code_bytes: i32,
        codes_row_stride: i64, codes_col_stride: i64) -> i32;
    pub fn pqhip_reconstruct_batch_f32(cb: *mut pqhip_codebook, codes: *const c_void, code_bytes: i32,
        n_rows: i64, codes_row_stride: i64, codes_col_stride: i64, out: *mut f32,
        out_row_stride: i64, out_col_stride: i64) -> i32;
    pub fn pqhip_cluster_assignments_f32(ctx: *mut pqhip_ctx, centroids: *const f32, n_centroids: i64,
        dim: i64, x: *const f32, n_rows: i64, x_row_stride: i64, x_col_stride: i64, out: *mut c_void,
        out_bytes: i32) -> i32;
    pub fn pqhip_kmeans_iterations_f32(ctx: *mut pqhip_ctx, quantizers: *mut f32, n_subquantizers: i64,
        n_centroids: i64, sub_dim: i64, x: *const f32, n_rows: i64, x_row_stride: i64,
        x_col_stride: i64, n_iterations: i32, loss: *mut f32) -> i32;
    pub fn pqhip_matrix_upload_f32(ctx: *mut pqhip_ctx, device_slot: i32, x: *const f32, n_rows: i64,
        n_cols: i64, x_row_stride: i64, x_col_stride: i64, out: *mut *mut pqhip_matrix) -> i32;
    pub fn pqhip_matrix_device_ptr(m: *const pqhip_matrix) -> *const f32;
    pub fn pqhip_matrix_destroy(m: *mut pqhip_matrix);
    pub fn pqhip_ctx_set_option(ctx: *mut pqhip_ctx, name: *const c_char, value: i64) -> i32;
    pub fn pqhip_opq_train_step_f32_dev(ctx: *mut pqhip_ctx, device_slot: i32, quantizers: *mut f32,
        n_subquantizers: i64, n_centroids: i64, sub_dim: i64, projection: *const f32, d_x: *const f32,
        n_rows: i64, x_row_stride: i64, cross: *mut f32, stream: *mut c_void) -> i32;
}

/// Batches smaller than this stay on the CPU path (a launch + PCIe round trip is pointless).
const MIN_GPU_ROWS: usize = 4096;

/// Device images of `Pq<f32>` values, kept OUTSIDE the struct so that `Pq` keeps
/// `#[derive(Clone, Debug, PartialEq)]` and literal construction (pq.rs:28-32, opq.rs:95-98).
/// Same policy as include/reductive_amd/codebook_cache.hpp (which the test-suite executes,
/// tests/cpp/test_codebook_cache.cpp, including 4 threads x 2 quantizers overlapping on one GPU):
///  * `Pq<f32>` is `Send + Sync`: the cache mutex is held for lookup / insert / evict ONLY, never across
///    a GPU call or the creation of a device image;
///  * a lookup returns an `Arc<Image>` PIN: an entry evicted or replaced while calls run on it is
///    destroyed when the last pin drops (eviction waits for users, users never wait for each other);
///  * key = (quantizer data pointer, element count, ORIGINAL projection data pointer or 0, M, K, dsub)
///    -- never the address of a temporary copy;
///  * a hit is trusted only when the FULL 64-bit content hash of quantizers and projection (every byte is read:
///    eight independent lanes, one AES round per 16 bytes with AES-NI, multiply-xorshift lanes otherwise --
///    ~12 us for the 307 KB headline codebook) and the GENERATION counter still match.  Round 3 hashed a sample
///    only; a dropped `Pq` whose allocation is reused by a quantizer that differs outside the sample was then
///    served a stale image.  Every training exit bumps the generation (`TrainingExit` guard in
///    `try_kmeans_iterations` -- CPU fall-through included -- and `train_step`; call `training_finished()` after
///    any other in-place rewrite of centroids or projection), so a stale image can never be served after them;
///  * at most `CACHE_CAP` entries, least recently used dropped: device memory stays bounded.
const CACHE_CAP: usize = 8;
#[derive(PartialEq, Clone, Copy)]
struct Key { q: usize, q_len: usize, p: usize, m: usize, k: usize, dsub: usize }
struct Image(*mut pqhip_codebook);
unsafe impl Send for Image {}
unsafe impl Sync for Image {}            // the C ABI's entry points are re-entrant on one codebook handle
impl Drop for Image { fn drop(&mut self) { unsafe { pqhip_codebook_destroy(self.0) } } }
struct Entry { key: Key, hash: u64, gen: u64, image: Arc<Image> }
struct Ctx(*mut pqhip_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
struct Handles { ctx: Ctx, books: Mutex<Vec<Entry>> /* most recently used first */, generation: AtomicU64 }
static HANDLES: OnceLock<Option<Handles>> = OnceLock::new();

fn handles() -> Option<&'static Handles> {
    HANDLES.get_or_init(|| {
        let mut ctx = std::ptr::null_mut();
        if unsafe { pqhip_ctx_create(std::ptr::null(), 0, &mut ctx) } == PQHIP_OK {
            Some(Handles { ctx: Ctx(ctx), books: Mutex::new(Vec::new()), generation: AtomicU64::new(0) })
        } else { None }
    }).as_ref()
}

/// content_hash of codebook_cache.hpp (same two forms; the values are process-local, so the forms need not agree).
/// Portable form: eight 64-bit lanes, `h_j = xorshift((h_j ^ w) * prime)` per 64-byte block -- a bijection of the
/// lane per step, so one changed word always changes its lane.
fn content_hash_lanes(bytes: &[u8], mut h: u64) -> u64 {
    const P: u64 = 0x100000001b3;
    let mut l = [0u64; 8];
    for j in 0..8 { l[j] = h ^ 0x9e3779b97f4a7c15u64.wrapping_mul(j as u64 + 1); }
    let mut blocks = bytes.chunks_exact(64);
    for b in &mut blocks {
        for j in 0..8 {
            let w = u64::from_ne_bytes(b[8 * j..8 * j + 8].try_into().unwrap());
            l[j] = (l[j] ^ w).wrapping_mul(P);
            l[j] ^= l[j] >> 29;
        }
    }
    h ^= (bytes.len() as u64).wrapping_mul(P);
    for j in 0..8 { h = (h ^ l[j]).wrapping_mul(P); h ^= h >> 29; }
    let mut words = blocks.remainder().chunks_exact(8);
    for c in &mut words { h = (h ^ u64::from_ne_bytes(c.try_into().unwrap())).wrapping_mul(P); h ^= h >> 29; }
    for &b in words.remainder() { h = (h ^ b as u64).wrapping_mul(P); }
    h
}
/// AES-NI form: eight 128-bit lanes, `state = AESENC(state ^ block, key)` (a permutation of state and of block).
#[cfg(target_arch = "x86_64")]
#[target_feature(enable = "aes,sse2")]
unsafe fn content_hash_aes(bytes: &[u8], h: u64) -> u64 {
    use std::arch::x86_64::*;
    let key = _mm_set_epi64x(0xc2b2ae3d27d4eb4fu64 as i64, 0x9e3779b97f4a7c15u64 as i64);
    let mut s = [_mm_setzero_si128(); 8];
    for j in 0..8 { s[j] = _mm_set_epi64x((!h ^ ((j as u64) << 32)) as i64, h.wrapping_add(j as u64) as i64); }
    let mut blocks = bytes.chunks_exact(128);
    for b in &mut blocks {
        for j in 0..8 {
            let v = _mm_loadu_si128(b.as_ptr().add(16 * j) as *const __m128i);
            s[j] = _mm_aesenc_si128(_mm_xor_si128(s[j], v), key);
        }
    }
    let mut acc = _mm_set_epi64x(h as i64, bytes.len() as i64);
    for j in 0..8 { acc = _mm_aesenc_si128(_mm_xor_si128(acc, s[j]), key); }
    let mut rest = blocks.remainder().chunks_exact(16);
    for c in &mut rest { acc = _mm_aesenc_si128(_mm_xor_si128(acc, _mm_loadu_si128(c.as_ptr() as *const __m128i)), key); }
    let tail = rest.remainder();
    if !tail.is_empty() {
        let mut t = [0u8; 16];
        t[..tail.len()].copy_from_slice(tail);
        acc = _mm_aesenc_si128(_mm_xor_si128(acc, _mm_loadu_si128(t.as_ptr() as *const __m128i)), key);
    }
    acc = _mm_aesenc_si128(_mm_aesenc_si128(acc, key), key);
    (_mm_cvtsi128_si64(acc) ^ _mm_extract_epi64(acc, 1)) as u64
}
/// FULL content hash of an f32 array (every byte is read).
fn content_hash(data: &[f32], h: u64) -> u64 {
    let bytes = unsafe { std::slice::from_raw_parts(data.as_ptr() as *const u8, data.len() * 4) };
    #[cfg(target_arch = "x86_64")]
    { if std::is_x86_feature_detected!("aes") { return unsafe { content_hash_aes(bytes, h) }; } }
    content_hash_lanes(bytes, h)
}

/// Pinned device image of (quantizers, projection).  `None` => no device / creation failed => CPU path.
/// The cache lock is NOT held when this returns: run the GPU call on `pin.0`, then drop the pin.
fn codebook(q: ArrayView3<f32>, p: Option<ArrayView2<f32>>) -> Option<Arc<Image>> {
    // contiguous copies only when the views are not in standard layout; the KEY uses the caller's pointers
    let key_q = q.as_ptr() as usize;
    let key_p = p.as_ref().map_or(0, |p| p.as_ptr() as usize);
    let qs = q.as_standard_layout();
    let ps = p.as_ref().map(|p| p.as_standard_layout());
    let (m, k, dsub) = qs.dim();
    let key = Key { q: key_q, q_len: qs.len(), p: key_p, m, k, dsub };
    let mut hash = content_hash(qs.as_slice()?, 0xcbf29ce484222325);      // FULL contents
    if let Some(ps) = &ps { hash = content_hash(ps.as_slice()?, hash); }
    let h = handles()?;
    let gen = h.generation.load(Ordering::Acquire);
    let stale;                                           // dropped (=> destroyed) after the lock is released
    {
        let mut books = h.books.lock().ok()?;
        match books.iter().position(|e| e.key == key) {
            Some(i) if books[i].hash == hash && books[i].gen == gen => {
                let e = books.remove(i);
                let pin = e.image.clone();
                books.insert(0, e);
                return Some(pin);
            }
            Some(i) => stale = Some(books.remove(i)),     // same address, other contents or trained since
            None => stale = None,
        }
    }
    drop(stale);
    // the device image is built with the lock RELEASED
    let mut cb = std::ptr::null_mut();
    let rc = unsafe { pqhip_codebook_create(h.ctx.0, qs.as_ptr(), m as i64, k as i64, dsub as i64,
        ps.as_ref().map_or(std::ptr::null(), |p| p.as_ptr()), &mut cb) };
    if rc != PQHIP_OK { return None; }
    let fresh = Arc::new(Image(cb));
    let mut dropped = Vec::new();                        // evicted entries die outside the lock too
    {
        let mut books = h.books.lock().ok()?;
        if let Some(i) = books.iter().position(|e| e.key == key) {
            if books[i].hash == hash && books[i].gen == gen { return Some(books[i].image.clone()); }   // built twice: keep theirs
            dropped.push(books.remove(i));
        }
        books.insert(0, Entry { key, hash, gen, image: fresh.clone() });
        while books.len() > CACHE_CAP { dropped.push(books.pop().unwrap()); }
    }
    drop(dropped);
    Some(fresh)
}

/// Runs `f` on the pinned device image; no lock is held while `f` (the GPU call) runs.
fn with_codebook<R>(q: ArrayView3<f32>, p: Option<ArrayView2<f32>>, f: impl FnOnce(*mut pqhip_codebook) -> R) -> Option<R> {
    let pin = codebook(q, p)?;
    Some(f(pin.0))
}

/// Every entry point that rewrites quantizers or projections in place calls this.
fn invalidate_images() { if let Some(h) = handles() { h.generation.fetch_add(1, Ordering::AcqRel); } }
/// Public form for the patched trainers: call at EVERY exit of a training loop that rewrote centroids or the
/// projection in place on the CPU (the fall-through after `try_kmeans_iterations` returned false, `Opq::train_iteration`'s
/// `projection.assign(..)`): no device image created before it is served again.
pub fn training_finished() { invalidate_images(); }
/// Bumps the generation when it goes out of scope -- on every return path of a training entry point, including the
/// early `return false` exits after which the caller's CPU loop rewrites the centroids in place.
struct TrainingExit;
impl Drop for TrainingExit { fn drop(&mut self) { invalidate_images(); } }

fn same<A: 'static, B: 'static>() -> bool { TypeId::of::<A>() == TypeId::of::<B>() }

/// Reinterpret an `ArrayView<A>` as `ArrayView<f32>` once `A == f32` has been established with `TypeId`.
/// (A `transmute` between the two view types is rejected for a generic `A` -- E0512, the sizes are not
/// known to be equal -- so the view is rebuilt from its raw parts: same pointer, shape and strides.)
unsafe fn view3_as_f32<'a, A: 'static>(v: ArrayView3<'a, A>) -> ArrayView3<'a, f32> {
    debug_assert!(same::<A, f32>());
    let (d0, d1, d2) = v.dim();
    let s = v.strides();
    ArrayView3::from_shape_ptr((d0, d1, d2).strides((s[0] as usize, s[1] as usize, s[2] as usize)), v.as_ptr() as *const f32)
}
unsafe fn view2_as_f32<'a, A: 'static>(v: ArrayView2<'a, A>) -> ArrayView2<'a, f32> {
    debug_assert!(same::<A, f32>());
    let (d0, d1) = v.dim();
    let s = v.strides();
    ArrayView2::from_shape_ptr((d0, d1).strides((s[0] as usize, s[1] as usize)), v.as_ptr() as *const f32)
}
fn non_negative(strides: &[isize]) -> bool { strides.iter().all(|&s| s >= 0) }

/// Returns true when the GPU produced the result; false => caller runs `primitives::*` unchanged.
/// The shape asserts of primitives.rs:74-87 are performed by the caller BEFORE this call, so panic
/// messages are unchanged.
pub fn try_quantize_batch<A, I, S>(quantizers: ArrayView3<A>, projection: Option<ArrayView2<A>>,
    x: &ArrayBase<S, Ix2>, mut quantized: ArrayViewMut2<I>) -> bool
where A: 'static + Copy, I: 'static + Copy, S: Data<Elem = A>,
{
    if !same::<A, f32>() || x.nrows() < MIN_GPU_ROWS { return false; }
    let code_bytes = std::mem::size_of::<I>() as i32;
    if !(same::<I, u8>() || same::<I, u16>() || same::<I, u32>() || same::<I, u64>() || same::<I, usize>()) { return false; }
    if !non_negative(quantizers.strides()) || projection.as_ref().map_or(false, |p| !non_negative(p.strides())) { return false; }
    let (xs, qs) = (x.strides(), quantized.strides());
    if !non_negative(xs) || !non_negative(qs) { return false; }
    // SAFETY: A == f32 was checked above; pointer, shape and strides are carried over unchanged.
    let (q, p) = unsafe { (view3_as_f32(quantizers), projection.map(|p| view2_as_f32(p))) };
    let (x_ptr, n) = (x.as_ptr() as *const f32, x.nrows() as i64);
    let out = quantized.as_mut_ptr() as *mut c_void;
    let rc = with_codebook(q, p, |cb| unsafe {
        pqhip_quantize_batch_f32(cb, x_ptr, n, xs[0] as i64, xs[1] as i64, out, code_bytes, qs[0] as i64, qs[1] as i64)
    });
    let rc = match rc { Some(rc) => rc, None => return false };
    // EINDEX_WIDTH: K-1 > I::MAX. The reference's batch path wraps silently (primitives.rs:98-100);
    // fall back so that (odd) behaviour is preserved bit for bit.
    rc == PQHIP_OK
}

pub fn try_reconstruct_batch<A, I, S>(quantizers: ArrayView3<A>, projection: Option<ArrayView2<A>>,
    quantized: &ArrayBase<S, Ix2>, mut reconstructions: ArrayViewMut2<A>) -> bool
where A: 'static + Copy, I: 'static + Copy, S: Data<Elem = I>,
{
    if !same::<A, f32>() || quantized.nrows() < MIN_GPU_ROWS { return false; }
    if !(same::<I, u8>() || same::<I, u16>() || same::<I, u32>() || same::<I, u64>() || same::<I, usize>()) { return false; }
    if !non_negative(quantizers.strides()) || projection.as_ref().map_or(false, |p| !non_negative(p.strides())) { return false; }
    let (cs, os) = (quantized.strides(), reconstructions.strides());
    if !non_negative(cs) || !non_negative(os) { return false; }
    // SAFETY: A == f32 was checked above.
    let (q, p) = unsafe { (view3_as_f32(quantizers), projection.map(|p| view2_as_f32(p))) };
    let (c_ptr, n) = (quantized.as_ptr() as *const c_void, quantized.nrows() as i64);
    let out = reconstructions.as_mut_ptr() as *mut f32;
    let rc = with_codebook(q, p, |cb| unsafe {
        pqhip_reconstruct_batch_f32(cb, c_ptr, std::mem::size_of::<I>() as i32, n, cs[0] as i64, cs[1] as i64,
                                    out, os[0] as i64, os[1] as i64)
    });
    let rc = match rc { Some(rc) => rc, None => return false };
    if rc == PQHIP_ECODE_RANGE { panic!("ndarray: index out of bounds"); }   // primitives.rs:146
    rc == PQHIP_OK
}

/// The k-means step of training for ALL subquantizers at once ("next" row of the hot path).
/// Call sites in the reference:
///  * `Pq::train_pq_using` (pq.rs:214-241): draw the initial centroids of every subquantizer with
///    the per-subquantizer XorShift rngs exactly as `subquantizer_initial_centroids` does
///    (pq.rs:105-123), stack them into one [M, K, dsub] array, then ONE call here with
///    `n_iterations` replaces the M x `kmeans_with_centroids(.., NIterationsCondition(n))` of
///    pq.rs:176; `losses` feeds the `min_by_key` over attempts (pq.rs:182-187).
///  * `Opq::update_subquantizers` (opq.rs:227-245): n_iterations = 1, losses = None, x = rx.
/// Returns false (caller keeps the CPU path) for A != f32, small inputs, negative strides, a
/// missing device or any non-OK status.  Results are bit-identical to the sequential f32 arithmetic
/// of kmeans.rs:166-198 and :329-360 under the crate's default (matrixmultiply) backend.
pub fn try_kmeans_iterations<A, S>(mut quantizers: ArrayViewMut3<A>, instances: &ArrayBase<S, Ix2>,
    n_iterations: usize, losses: Option<&mut [A]>) -> bool
where A: 'static + Copy, S: Data<Elem = A>,
{
    let _exit = TrainingExit;                              // generation bump on EVERY exit, CPU fall-through included
    if !same::<A, f32>() || instances.nrows() < MIN_GPU_ROWS || !quantizers.is_standard_layout() { return false; }
    let h = match handles() { Some(h) => h, None => return false };      // (no cache lock: serving calls keep running)
    let (m, k, dsub) = quantizers.dim();
    if instances.ncols() != m * dsub { return false; }          // the caller's asserts fire on the CPU path
    let xs = instances.strides();
    if xs.iter().any(|&s| s < 0) { return false; }
    let loss_ptr = match losses { Some(l) if l.len() == m => l.as_mut_ptr() as *mut f32, Some(_) => return false, None => std::ptr::null_mut() };
    let rc = unsafe { pqhip_kmeans_iterations_f32(h.ctx.0, quantizers.as_mut_ptr() as *mut f32, m as i64, k as i64,
        dsub as i64, instances.as_ptr() as *const f32, instances.nrows() as i64, xs[0] as i64, xs[1] as i64,
        n_iterations as i32, loss_ptr) };
    rc == PQHIP_OK                                         // (`_exit` bumps the generation: the centroids were rewritten in place)
}


/// `instances.t().dot(&reconstructed)` of `Opq::train_iteration` (opq.rs:191) on the device: `true` (default) keeps
/// matrixmultiply's order -- one fmaf chain per 256-row block, block results added in row order -- bit for bit;
/// `false` is a plain split-K product within 1e-5 relative of it (no per-block partial matrices: 16 ms instead of 20 ms
/// per 10 M x 300 rows).  Builds of the crate that can train OPQ link a BLAS (Cargo.toml:37-42), whose summation order
/// is its own; choose per deployment.
pub fn set_cross_product_exact(exact: bool) -> bool {
    match handles() {
        Some(h) => unsafe { pqhip_ctx_set_option(h.ctx.0, b"cross_product_exact\0".as_ptr() as *const c_char, exact as i64) == PQHIP_OK },
        None => false,
    }
}

/// Candidate tables of the 1- / 2-float sub-vector encode kernel (include/pqhip.h, option "candidate_tables"): built on the host
/// when the device image of a `Pq` is created -- about 0.9 s for M = 150, K = 256 on a GPU box's host, 8 ms for M = 10, K = 128 --
/// and repaid only by some 1e8 encoded rows per codebook.  `false` makes images created from now on go without them (same codes
/// from the kernels that evaluate every centroid): the setting for a caller that quantizes one vocabulary per codebook.
pub fn set_candidate_tables(build: bool) -> bool {
    match handles() {
        Some(h) => unsafe { pqhip_ctx_set_option(h.ctx.0, b"candidate_tables\0".as_ptr() as *const c_char, build as i64) == PQHIP_OK },
        None => false,
    }
}

/// Instances kept in HBM for the length of an OPQ training run (`Opq::train_pq_using`, opq.rs:44-99):
/// upload once, then one `train_step` per iteration replaces opq.rs:167-182 and the GEMM of :191.
pub struct ResidentInstances { m: *mut pqhip_matrix, rows: usize, cols: usize }
impl ResidentInstances {
    pub fn upload<S: Data<Elem = f32>>(x: &ArrayBase<S, Ix2>) -> Option<Self> {
        let h = handles()?;
        let (xs, mut m) = (x.strides(), std::ptr::null_mut());
        if xs.iter().any(|&s| s < 0) { return None; }
        let rc = unsafe { pqhip_matrix_upload_f32(h.ctx.0, 0, x.as_ptr(), x.nrows() as i64, x.ncols() as i64,
            xs[0] as i64, xs[1] as i64, &mut m) };
        if rc == PQHIP_OK { Some(Self { m, rows: x.nrows(), cols: x.ncols() }) } else { None }
    }
    /// In `Opq::train_iteration`: `if let Some(cross) = resident.train_step(projection.view(), centroids.view_mut())
    /// { let (u, _, vt) = cross.svd(true, true).unwrap(); projection.assign(&u.unwrap().dot(&vt.unwrap())); return; }`
    pub fn train_step(&self, projection: ArrayView2<f32>, mut quantizers: ArrayViewMut3<f32>) -> Option<ndarray::Array2<f32>> {
        let _exit = TrainingExit;                          // generation bump on every exit of this step
        let h = handles()?;                                // (no cache lock: a training step never blocks serving calls)
        let (m, k, dsub) = quantizers.dim();
        if m * dsub != self.cols || !quantizers.is_standard_layout() { return None; }
        let p = projection.as_standard_layout();
        let mut cross = ndarray::Array2::<f32>::zeros((self.cols, self.cols));
        let rc = unsafe { pqhip_opq_train_step_f32_dev(h.ctx.0, 0, quantizers.as_mut_ptr(), m as i64, k as i64, dsub as i64,
            p.as_ptr(), pqhip_matrix_device_ptr(self.m), self.rows as i64, self.cols as i64, cross.as_mut_ptr(),
            std::ptr::null_mut()) };
        if rc == PQHIP_OK { Some(cross) } else { None }
    }
}
impl Drop for ResidentInstances { fn drop(&mut self) { unsafe { pqhip_matrix_destroy(self.m) } } }
