"""Host-side mirror of reductive's `Pq<f32>` (src/pq/pq.rs:28-348, src/pq/traits.rs:75-156).

Same names, argument meaning and error behaviour as the reference's `QuantizeVector` /
`Reconstruct` traits; the batch methods -- the hot path -- dispatch to libpqhip.so through the
C ABI (include/pqhip.h).  The reference panics on shape errors; here a panic is `PanicError`
carrying the reference's message.  There is no CPU fallback for the batch methods: without the
HIP library or a gfx950 device they raise.

The single-vector methods (`quantize_vector`, `reconstruct`) are the reference's latency path
(pq.rs:285-298, 329-343) and stay on the host, as in the Rust integration.
"""
import ctypes
import threading

import numpy as np

from . import _lib

_INDEX_TYPES = (np.uint8, np.uint16, np.uint32, np.uint64)


class PanicError(AssertionError):
    """A Rust `panic!` / failed `assert!` of the reference surfaced as an exception."""


class ReductiveError(ValueError):
    """`ReductiveError` (src/error.rs:6-41): invalid training hyper-parameters."""


_default_ctx = None
_ctx_lock = threading.Lock()


class _Ctx:
    def __init__(self, devices=None):
        L = _lib.lib()
        h = ctypes.c_void_p()
        if devices:
            arr = (ctypes.c_int32 * len(devices))(*devices)
            rc = L.pqhip_ctx_create(arr, len(devices), ctypes.byref(h))
        else:
            rc = L.pqhip_ctx_create(None, 0, ctypes.byref(h))
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_ctx_create")
        self.handle = h
        self.devices = list(devices) if devices else None
        self.n_devices = L.pqhip_ctx_n_devices(h)

    def close(self):
        if self.handle:
            _lib.lib().pqhip_ctx_destroy(self.handle)
            self.handle = None

    def set_option(self, name, value):
        """test / A-B knob (include/pqhip.h: pqhip_ctx_set_option), e.g. ("opq_fused", 0), ("kmeans_window_rows", 64)."""
        rc = _lib.lib().pqhip_ctx_set_option(self.handle, name.encode(), int(value))
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_ctx_set_option(%s)" % name)


def set_option(name, value, ctx=None):
    """pqhip_ctx_set_option on `ctx` (default: the process-wide context)."""
    (ctx or default_ctx()).set_option(name, value)


def launch_log(reset=False):
    """Kernels the library launched from this thread since the last reset ("k_a + k_b x3"; include/pqhip.h)."""
    L = _lib.lib()
    text = L.pqhip_launch_log().decode()
    if reset:
        L.pqhip_launch_log_reset()
    return text


def vor2_tables(quantizers):
    """The candidate tables of the 2-float sub-vector encode kernel for `quantizers` [M][K][2] (or [M][K][1]), built on the host exactly as
    at codebook creation (include/pqhip.h: pqhip_vor2_tables_host; layout: csrc/vor2_prep.h).  Returns (words uint32[],
    region_off uint32[M + 1]) or None when the codebook is not eligible.  Needs no GPU."""
    q = np.ascontiguousarray(quantizers, dtype=np.float32)
    if q.ndim != 3 or q.shape[2] not in (1, 2):
        raise ReductiveError("vor2_tables: quantizers must be [M][K][1] or [M][K][2]")
    L = _lib.lib()
    M, K, dsub = q.shape
    n = ctypes.c_int64(0)
    off = np.zeros(M + 1, dtype=np.uint32)
    rc = L.pqhip_vor2_tables_host(q.ctypes.data, M, K, dsub, None, 0, None, ctypes.byref(n))
    if rc == _lib.EUNSUPPORTED:
        return None
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_vor2_tables_host")
    words = np.zeros(n.value, dtype=np.uint32)
    rc = L.pqhip_vor2_tables_host(q.ctypes.data, M, K, dsub, words.ctypes.data, words.size, off.ctypes.data, ctypes.byref(n))
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_vor2_tables_host")
    return words, off


def default_ctx():
    global _default_ctx
    with _ctx_lock:
        if _default_ctx is None:
            _default_ctx = _Ctx()
        return _default_ctx


def _estrides(a):
    return [s // a.itemsize for s in a.strides]


def _unrolled_dot_rows(a, b):
    """ndarray numeric_util::unrolled_dot of every row of a [r, n] with b ([n] or [r, n]);
    float32 with separately rounded multiply and add (numpy never fuses)."""
    prod = (a * (b if b.ndim == 2 else b[None, :])).astype(np.float32)
    n = a.shape[1]
    nf = (n // 8) * 8
    p = np.zeros((a.shape[0], 8), np.float32)
    for i in range(0, nf, 8):
        p = p + prod[:, i:i + 8]
    s = np.zeros(a.shape[0], np.float32)
    s = s + (p[:, 0] + p[:, 4])
    s = s + (p[:, 1] + p[:, 5])
    s = s + (p[:, 2] + p[:, 6])
    s = s + (p[:, 3] + p[:, 7])
    for i in range(nf, n):
        s = s + prod[:, i]
    return s


def _first_min(d):
    """kmeans.rs:119-125: first minimum under ordered-float (NaN greatest, -0 == +0)."""
    ok = ~np.isnan(d)
    if not ok.any():
        return 0
    return int(np.flatnonzero(ok & (d == d[ok].min()))[0])


def cluster_assignments(centroids, instances, dtype=np.uint64, ctx=None):
    """`kmeans::cluster_assignments(centroids, instances, Axis(0))` (src/kmeans.rs:133-159) on the
    GPU: index of the nearest centroid [K, dim] for every row of `instances` [n, dim]."""
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    x = np.asarray(instances, dtype=np.float32)
    if centroids.ndim != 2 or x.ndim != 2 or x.shape[1] != centroids.shape[1]:
        raise PanicError("Cannot compute (squared) euclidean distance of matrices with different "
                         "numbers of columns.")                                  # linalg.rs:161-165
    out = np.zeros(x.shape[0], dtype=dtype)
    if x.shape[0] == 0:
        return out
    if any(s < 0 for s in x.strides) or any(s % 4 for s in x.strides):
        x = np.ascontiguousarray(x)
    rs, cs = _estrides(x)
    ctx = ctx or default_ctx()
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _lib.lib().pqhip_cluster_assignments_f32(ctx.handle, centroids.ctypes.data_as(fp),
                                                  centroids.shape[0], centroids.shape[1],
                                                  x.ctypes.data, x.shape[0], rs, cs,
                                                  out.ctypes.data, out.itemsize)
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_cluster_assignments_f32")
    return out


def kmeans_iterations(quantizers, instances, n_iterations=1, want_loss=True, ctx=None):
    """`n_iterations` x `kmeans_iteration` (src/kmeans.rs:308-327) on every subquantizer's column
    block, on the GPU: the body of `kmeans_with_centroids(.., NIterationsCondition(n))` at
    pq.rs:176 for all subquantizers of `train_pq_using`, and of `Opq::update_subquantizers`
    (opq.rs:227-245) when n_iterations == 1.  quantizers [M, K, dsub] are the initial centroids;
    returns (updated quantizers, last mean squared error per subquantizer or None).
    `instances` may be a numpy array [n, d] or a CUDA float32 torch tensor (kept in HBM)."""
    q = np.array(quantizers, dtype=np.float32, order="C", copy=True)
    if q.ndim == 2:
        q = q[None]
    if not hasattr(instances, "is_cuda"):
        instances = np.asarray(instances, dtype=np.float32)
    if q.ndim != 3 or q.shape[1] == 0:
        raise PanicError("Cannot cluster instances with zero centroids.")        # kmeans.rs:260-263
    M, K, dsub = q.shape
    if instances.ndim != 2 or instances.shape[1] != M * dsub:
        raise PanicError("Centroid and instance lengths differ.")                # kmeans.rs:264-268
    ctx = ctx or default_ctx()
    loss = np.zeros(M, np.float32) if want_loss else None
    fp = ctypes.POINTER(ctypes.c_float)
    lp = loss.ctypes.data_as(fp) if want_loss else None
    L = _lib.lib()
    if hasattr(instances, "is_cuda"):
        import torch
        x = instances
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
        if x.shape[1] != M * dsub:
            raise PanicError("Centroid and instance lengths differ.")            # kmeans.rs:264-268
        if x.stride(1) != 1:
            x = x.contiguous()
        dev = x.device.index or 0
        slot = dev if ctx.devices is None else ctx.devices.index(dev)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rs = x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])
        rc = L.pqhip_kmeans_iterations_f32_dev(ctx.handle, slot, q.ctypes.data_as(fp), M, K, dsub,
                                               x.data_ptr(), x.shape[0], rs, n_iterations, lp,
                                               ctypes.c_void_p(stream))
    else:
        x = np.asarray(instances, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != M * dsub:
            raise PanicError("Centroid and instance lengths differ.")            # kmeans.rs:264-268
        if any(s < 0 for s in x.strides) or any(s % 4 for s in x.strides):
            x = np.ascontiguousarray(x)
        rs, cs = _estrides(x) if x.size else (x.shape[1], 1)
        rc = L.pqhip_kmeans_iterations_f32(ctx.handle, q.ctypes.data_as(fp), M, K, dsub, x.ctypes.data,
                                           x.shape[0], rs, cs, n_iterations, lp)
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_kmeans_iterations_f32")
    return q, loss


def train_pq(n_subquantizers, n_subquantizer_bits, n_iterations, n_attempts, instances, rng=None, ctx=None):
    """`Pq::train_pq_using` (src/pq/pq.rs:214-241 -> train_subquantizer pq.rs:144-188) with the
    k-means iterations on the GPU.  Initial centroids are K distinct random instances per
    subquantizer (RandomInstanceCentroids, pq.rs:166-172); the draw uses numpy's generator, not the
    reference's XorShift stream, so trained codebooks agree with the reference statistically, not
    bit for bit.  The best of `n_attempts` (lowest final loss, first on ties) is kept per subquantizer."""
    x = np.asarray(instances, dtype=np.float32)
    n, d = x.shape
    K = 1 << n_subquantizer_bits
    # check_quantizer_invariants (pq.rs:63-100), same order, messages of error.rs:6-41
    if n_subquantizers == 0 or n_subquantizers > d:
        raise ReductiveError("The number of subquantizers must be between 1 and %d, was %d" % (d, n_subquantizers))
    max_bits = int(np.trunc(np.log2(float(n)))) if n > 0 else 0
    if n_subquantizer_bits == 0 or n_subquantizer_bits > max_bits:
        raise ReductiveError("The number of subquantizers bits must be between 1 and %d" % max_bits)
    if d % n_subquantizers != 0:
        # (the reference's format string swaps the two numbers, error.rs:19-23; kept as is)
        raise ReductiveError("The number of columns (%d) is not exactly dividable by the number of "
                             "subquantizers (%d)" % (n_subquantizers, d))
    if n_iterations == 0:
        raise ReductiveError("The number of quantization iterations must be >= 1")
    if n_attempts == 0:
        raise ReductiveError("The number of quantization attempts per iteration must be >= 1")
    rng = rng or np.random.default_rng(0)
    M, dsub = n_subquantizers, d // n_subquantizers
    best_q, best_loss = None, None
    for _ in range(n_attempts):
        init = np.stack([x[rng.choice(n, K, replace=False), m * dsub:(m + 1) * dsub] for m in range(M)])
        q, loss = kmeans_iterations(init, x, n_iterations, want_loss=True, ctx=ctx)
        if best_q is None:
            best_q, best_loss = q, loss
        else:
            better = np.array([(not np.isnan(l)) and (np.isnan(b) or l < b) for l, b in zip(loss, best_loss)])
            best_q[better] = q[better]
            best_loss[better] = loss[better]
    return Pq(None, best_q, ctx=ctx)


def at_dot_b(a, b, ctx=None):
    """`a.t().dot(&b)` (opq.rs:191) for CUDA float32 tensors a [n, da], b [n, db] on the GPU with the
    reference's summation order over the rows; returns a numpy [da, db] array."""
    import torch
    assert a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.stride(1) == 1 and b.stride(1) == 1
    ctx = ctx or default_ctx()
    dev = a.device.index or 0
    slot = dev if ctx.devices is None else ctx.devices.index(dev)
    out = np.zeros((a.shape[1], b.shape[1]), np.float32)
    rs = lambda t: t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    rc = _lib.lib().pqhip_at_dot_b_f32_dev(ctx.handle, slot, a.data_ptr(), rs(a), a.shape[1], b.data_ptr(), rs(b),
                                           b.shape[1], a.shape[0], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_at_dot_b_f32_dev")
    return out


def opq_train_step(quantizers, projection, instances, ctx=None):
    """The device part of `Opq::train_iteration` (opq.rs:156-195): rotation, one k-means iteration
    per subquantizer, the quantize -> reconstruct round trip and `instances.t().dot(&reconstructed)`.
    instances: CUDA float32 tensor [n, d].  Returns (updated quantizers, cross [d, d]); the caller
    finishes with `u, _, vt = svd(cross); projection = u @ vt` (opq.rs:191-192)."""
    import torch
    q = np.array(quantizers, dtype=np.float32, order="C", copy=True)
    if q.ndim != 3 or q.size == 0:
        raise PanicError("Cannot cluster instances with zero centroids.")
    M, K, dsub = q.shape
    d = M * dsub
    P = np.ascontiguousarray(projection, dtype=np.float32)
    if list(P.shape) != [d, d]:
        raise PanicError("Incorrect projection matrix shape, was: %s, should be [%d, %d]" % (list(P.shape), d, d))
    x = instances
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    if x.shape[1] != d:
        raise PanicError("Centroid and instance lengths differ.")
    if x.stride(1) != 1:
        x = x.contiguous()
    ctx = ctx or default_ctx()
    dev = x.device.index or 0
    slot = dev if ctx.devices is None else ctx.devices.index(dev)
    cross = np.zeros((d, d), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _lib.lib().pqhip_opq_train_step_f32_dev(
        ctx.handle, slot, q.ctypes.data_as(fp), M, K, dsub, P.ctypes.data_as(fp), x.data_ptr(), x.shape[0],
        x.stride(0) if x.shape[0] > 1 else max(x.stride(0), d), cross.ctypes.data_as(fp),
        ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_opq_train_step_f32_dev")
    return q, cross


def bucket_eigenvalues(eigenvalues, n_buckets):
    """`bucket_eigenvalues` (opq.rs:198-273): eigenvalue allocation of Ge et al. 2013 -- largest
    first, each to the non-full bucket whose product (sum of shifted logs) is smallest."""
    ev = np.asarray(eigenvalues, dtype=np.float32)
    if n_buckets <= 0:
        raise PanicError("Cannot distribute eigenvalues over zero buckets.")
    if ev.shape[0] < n_buckets:
        raise PanicError("At least one eigenvalue is required per bucket")
    if ev.shape[0] % n_buckets != 0:
        raise PanicError("The number of eigenvalues should be a multiple of the number of buckets.")
    order = sorted(range(ev.shape[0]), key=lambda i: (np.isnan(ev[i]), ev[i]))     # ascending, NaN last
    eps = np.finfo(np.float32).eps
    if not ev[order[0]] >= -eps:
        raise PanicError("Bucketing is only supported for positive eigenvalues.")
    logs = np.log(ev + eps).astype(np.float32)
    logs = (logs - logs.min()).astype(np.float32)
    assignments = [[] for _ in range(n_buckets)]
    products = [np.float32(0)] * n_buckets
    cap = ev.shape[0] // n_buckets
    while order:
        i = order.pop()
        open_buckets = [b for b in range(n_buckets) if len(assignments[b]) < cap]
        b = min(open_buckets, key=lambda k: (products[k], k))                          # first minimum
        assignments[b].append(i)
        products[b] = np.float32(products[b] + logs[i])
    return assignments


def create_projection_matrix(instances, n_subquantizers):
    """`Opq::create_projection_matrix` (opq.rs:101-137): eigenvectors of the covariance matrix
    (linalg.rs:23-44, LAPACK `eigh`, upper triangle), columns ordered by `bucket_eigenvalues`."""
    x = np.asarray(instances, dtype=np.float32)
    n, d = x.shape
    if n == 0:
        raise PanicError("Cannot compute a covariance from zero observations")
    centered = x - x.mean(axis=0, dtype=np.float32)
    cov = (centered.T @ (centered / np.float32(n - 1))).astype(np.float32)
    evals, evecs = np.linalg.eigh(cov, UPLO="U")
    P = np.zeros((d, d), np.float32)
    for col, direction in enumerate(i for b in bucket_eigenvalues(evals, n_subquantizers) for i in b):
        P[:, col] = evecs[:, direction]
    return P


def set_rotation_variant(variant):
    """test knob (include/pqhip.h: pqhip_set_rotation_variant), process-wide: 0 auto, 8 / 9 force that rotation kernel."""
    rc = _lib.lib().pqhip_set_rotation_variant(variant)
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_set_rotation_variant")


def rotate(instances, projection, ctx=None):
    """`instances.dot(&projection)` (opq.rs:62, gaussian_opq.rs:55) on the GPU with the reference's
    summation order; CUDA float32 tensor [n, d] in, CUDA tensor out."""
    import torch
    x = instances
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    if x.stride(1) != 1:
        x = x.contiguous()
    d = x.shape[1]
    P = np.ascontiguousarray(projection, dtype=np.float32)
    if list(P.shape) != [d, d]:
        raise PanicError("Incorrect projection matrix shape, was: %s, should be [%d, %d]" % (list(P.shape), d, d))
    ctx = ctx or default_ctx()
    dev = x.device.index or 0
    slot = dev if ctx.devices is None else ctx.devices.index(dev)
    out = torch.empty((x.shape[0], d), dtype=torch.float32, device=x.device)
    rc = _lib.lib().pqhip_rotate_f32_dev(ctx.handle, slot, x.data_ptr(), x.shape[0],
                                         x.stride(0) if x.shape[0] > 1 else max(x.stride(0), d), d,
                                         P.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), out.data_ptr(), d,
                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != _lib.OK:
        raise _lib.PqHipError(rc, "pqhip_rotate_f32_dev")
    return out


def train_gaussian_opq(n_subquantizers, n_subquantizer_bits, n_iterations, n_attempts, instances, rng=None, ctx=None):
    """`GaussianOpq::train_pq_using` (gaussian_opq.rs:33-68): the PCA / eigenvalue-allocation
    projection of `create_projection_matrix`, then a plain PQ trained on the rotated instances
    (rotation and k-means on the GPU)."""
    import torch
    x = np.asarray(instances, dtype=np.float32)
    if x.ndim != 2 or n_subquantizers == 0 or n_subquantizers > x.shape[1]:
        raise ReductiveError("The number of subquantizers must be between 1 and %d, was %d"
                             % (x.shape[1] if x.ndim == 2 else 0, n_subquantizers))
    if x.shape[1] % n_subquantizers != 0:
        raise ReductiveError("The number of columns (%d) is not exactly dividable by the number of "
                             "subquantizers (%d)" % (n_subquantizers, x.shape[1]))
    P = create_projection_matrix(x, n_subquantizers)
    ctx = ctx or default_ctx()
    dev = torch.device("cuda", (ctx.devices[0] if ctx.devices else 0))
    rx = rotate(torch.from_numpy(np.ascontiguousarray(x)).to(dev), P, ctx=ctx).cpu().numpy()
    pq = train_pq(n_subquantizers, n_subquantizer_bits, n_iterations, n_attempts, rx, rng=rng, ctx=ctx)
    return Pq(P, pq.subquantizers(), ctx=ctx)


def train_opq(n_subquantizers, n_subquantizer_bits, n_iterations, n_attempts, instances, rng=None, ctx=None):
    """`Opq::train_pq_using` (opq.rs:44-99) with every data-sized step on the GPU: the iteration's
    rotation, k-means update, quantize -> reconstruct round trip and cross product run in
    `opq_train_step`; what stays on the host is LAPACK (covariance eigen-decomposition for the
    initial projection, opq.rs:101-137; one d x d SVD per iteration, opq.rs:191-192) and the
    random draw of the initial centroids (numpy's generator, not the reference's stream).
    `n_attempts` has no effect, as in the reference (opq.rs:34-36)."""
    import torch
    x = np.asarray(instances, dtype=np.float32)
    n, d = x.shape
    M, K = n_subquantizers, 1 << n_subquantizer_bits
    if M == 0 or M > d:
        raise ReductiveError("The number of subquantizers must be between 1 and %d, was %d" % (d, M))
    max_bits = int(np.trunc(np.log2(float(n)))) if n > 0 else 0
    if n_subquantizer_bits == 0 or n_subquantizer_bits > max_bits:
        raise ReductiveError("The number of subquantizers bits must be between 1 and %d" % max_bits)
    if d % M != 0:
        raise ReductiveError("The number of columns (%d) is not exactly dividable by the number of "
                             "subquantizers (%d)" % (M, d))
    if n_iterations == 0:
        raise ReductiveError("The number of quantization iterations must be >= 1")
    rng = rng or np.random.default_rng(0)
    dsub = d // M
    P = create_projection_matrix(x, M)
    # initial centroids: K distinct rows of rx per subquantizer (opq.rs:139-158)
    q = np.stack([(x[rng.choice(n, K, replace=False)] @ P)[:, m * dsub:(m + 1) * dsub] for m in range(M)]).astype(np.float32)
    ctx = ctx or default_ctx()
    dev = torch.device("cuda", (ctx.devices[0] if ctx.devices else 0))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for _ in range(n_iterations):
        q, cross = opq_train_step(q, P, xd, ctx=ctx)
        u, _, vt = np.linalg.svd(cross)                                                   # opq.rs:191
        P = (u @ vt).astype(np.float32)                                                   # opq.rs:192
    return Pq(P, q, ctx=ctx)


class Pq:
    """Product quantizer (Jegou et al., 2011) -- mirror of `reductive::pq::Pq<f32>`."""

    def __init__(self, projection, quantizers, ctx=None):
        """`Pq::new(projection, quantizers)` (pq.rs:38-61)."""
        quantizers = np.ascontiguousarray(quantizers, dtype=np.float32)
        if quantizers.ndim != 3 or quantizers.size == 0:
            raise PanicError("Attempted to construct a product quantizer without quantizers.")
        rl = quantizers.shape[0] * quantizers.shape[2]
        if projection is not None:
            projection = np.ascontiguousarray(projection, dtype=np.float32)
            if list(projection.shape) != [rl, rl]:
                raise PanicError("Incorrect projection matrix shape, was: %s, should be [%d, %d]"
                                 % (list(projection.shape), rl, rl))
        self._projection = projection
        self._quantizers = quantizers
        self._ctx = ctx
        self._handle = None
        self._lock = threading.Lock()

    # ---- accessors (pq.rs:103-110, 191-193) -------------------------------------------------
    def n_quantizer_centroids(self):
        return self._quantizers.shape[1]

    def projection(self):
        return self._projection

    def subquantizers(self):
        return self._quantizers

    def quantized_len(self):
        return self._quantizers.shape[0]

    def reconstructed_len(self):
        return self._quantizers.shape[0] * self._quantizers.shape[2]

    def __eq__(self, other):  # #[derive(PartialEq)] pq.rs:28 -- value equality, handle excluded
        if not isinstance(other, Pq):
            return NotImplemented
        pe = (self._projection is None) == (other._projection is None)
        if pe and self._projection is not None:
            pe = np.array_equal(self._projection, other._projection)
        return pe and np.array_equal(self._quantizers, other._quantizers)

    # ---- device handle (created lazily, cached; SURVEY.md section 8b "Ownership") -------------
    def _cb(self):
        with self._lock:
            if self._handle is None:
                L = _lib.lib()
                ctx = self._ctx or default_ctx()
                M, K, dsub = self._quantizers.shape
                fp = ctypes.POINTER(ctypes.c_float)
                h = ctypes.c_void_p()
                proj = self._projection.ctypes.data_as(fp) if self._projection is not None else None
                rc = L.pqhip_codebook_create(ctx.handle, self._quantizers.ctypes.data_as(fp), M, K,
                                             dsub, proj, ctypes.byref(h))
                if rc != _lib.OK:
                    raise _lib.PqHipError(rc, "pqhip_codebook_create")
                self._handle = h
                self._ctx = ctx
            return self._handle

    def close(self):
        with self._lock:
            if self._handle is not None:
                _lib.lib().pqhip_codebook_destroy(self._handle)
                self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_encode_variant(self, variant):
        """test/bench knob (include/pqhip.h: pqhip_set_encode_variant): 0 auto, 1 scalar anchor kernel, 2 / 4 MFMA kernels,
        6 small-codebook VALU kernel, 7 pair kernel (K <= 16), 8 fused OPQ kernel, 9 16x16x4 MFMA kernel."""
        rc = _lib.lib().pqhip_set_encode_variant(self._cb(), variant)
        if rc != _lib.OK:
            raise _lib.PqHipError(rc)

    def last_encode_kernel(self):
        return _lib.lib().pqhip_last_encode_kernel(self._cb()).decode()

    # ---- QuantizeVector (traits.rs:75-99) -----------------------------------------------------
    def quantize_batch(self, x, dtype=np.uint8):
        """`quantize_batch::<I, _>(x)` (pq.rs:256-265); `dtype` plays the role of `I`."""
        x = np.asarray(x, dtype=np.float32)
        if x.ndim != 2:
            raise PanicError("quantize_batch expects a matrix")
        quantized = np.zeros((x.shape[0], self.quantized_len()), dtype=dtype)
        self.quantize_batch_into(x, quantized)
        return quantized

    def quantize_batch_into(self, x, quantized):
        """`quantize_batch_into` (pq.rs:268-283 -> primitives.rs:64-104) on the GPU."""
        x = np.asarray(x, dtype=np.float32)
        if quantized.dtype.type not in _INDEX_TYPES:
            raise TypeError("index type must be one of u8/u16/u32/u64")
        if x.ndim != 2 or x.shape[1] != self.reconstructed_len():
            raise PanicError("Quantizer and vector length mismatch")            # primitives.rs:74-78
        if quantized.shape != (x.shape[0], self.quantized_len()):
            raise PanicError("Quantized matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (x.shape[0], self.quantized_len(), quantized.shape[0],
                                quantized.shape[1] if quantized.ndim > 1 else 0))  # primitives.rs:80-87
        if x.shape[0] == 0:
            return
        if any(s < 0 for s in x.strides) or any(s % 4 for s in x.strides):
            x = np.ascontiguousarray(x)
        if any(s < 0 for s in quantized.strides):
            raise ValueError("negative output strides are not supported")
        rs, cs = _estrides(x)
        ors, ocs = _estrides(quantized)
        rc = _lib.lib().pqhip_quantize_batch_f32(self._cb(), x.ctypes.data, x.shape[0], rs, cs,
                                                quantized.ctypes.data, quantized.itemsize, ors, ocs)
        if rc == _lib.EINDEX_WIDTH:
            # the batch path of the reference has no such assert (it silently wraps,
            # primitives.rs:98-100); the GPU entry point refuses instead -- see DESIGN.md
            raise PanicError("Cannot store centroids in quantizer index type")
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_quantize_batch_f32")

    def quantize_vector(self, x, dtype=np.uint8):
        """`quantize_vector` (pq.rs:285-298 -> primitives.rs:14-49 -> linalg.rs:118-148); host."""
        x = np.asarray(x, dtype=np.float32)
        if x.ndim != 1 or x.shape[0] != self.reconstructed_len():
            raise PanicError("Quantizer and vector length mismatch")            # primitives.rs:25-29
        K = self.n_quantizer_centroids()
        if K - 1 > np.iinfo(dtype).max:
            raise PanicError("Cannot store centroids in quantizer index type")  # primitives.rs:31-34
        if self._projection is not None:
            # 1-D x 2-D ndarray dot without BLAS: sequential  s = s + x[k]*P[k,c]  per column
            rx = np.zeros(x.shape[0], np.float32)
            for k in range(x.shape[0]):
                rx = rx + x[k] * self._projection[k]
            x = rx
        M, _, dsub = self._quantizers.shape
        out = np.zeros(M, dtype=dtype)
        with np.errstate(invalid="ignore", over="ignore"):
            for m in range(M):
                xs = x[m * dsub:(m + 1) * dsub]
                c = self._quantizers[m]
                xx = _unrolled_dot_rows(xs[None, :], xs)[0]
                cc = _unrolled_dot_rows(c, c)
                dp = _unrolled_dot_rows(c, xs)
                d = (xx + cc) - (dp + dp)
                out[m] = _first_min(d.astype(np.float32))
        return out

    # ---- Reconstruct (traits.rs:102-156) ------------------------------------------------------
    def reconstruct_batch(self, quantized):
        """`reconstruct_batch` default method (traits.rs:109-117)."""
        quantized = np.asarray(quantized)
        if quantized.ndim != 2:
            raise PanicError("reconstruct_batch expects a matrix")
        out = np.zeros((quantized.shape[0], self.reconstructed_len()), np.float32)
        self.reconstruct_batch_into(quantized, out)
        return out

    def reconstruct_batch_into(self, quantized, reconstructions):
        """`reconstruct_batch_into` (pq.rs:309-327 -> primitives.rs:150-173) on the GPU."""
        quantized = np.asarray(quantized)
        if quantized.dtype.kind == "i":
            if (quantized < 0).any():
                raise PanicError("negative code")
            quantized = quantized.astype(np.uint64)
        if quantized.dtype.type not in _INDEX_TYPES:
            raise TypeError("index type must be one of u8/u16/u32/u64")
        if reconstructions.dtype != np.float32:
            raise TypeError("reconstructions must be float32")
        if (quantized.ndim != 2 or reconstructions.ndim != 2
                or reconstructions.shape[0] != quantized.shape[0]
                or reconstructions.shape[1] != self.reconstructed_len()):
            raise PanicError("Reconstructions matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (quantized.shape[0], self.reconstructed_len(),
                                reconstructions.shape[0], reconstructions.shape[-1]))  # primitives.rs:159-167
        if quantized.shape[1] != self.quantized_len():
            raise PanicError("Quantization length does not match number of subquantizers")  # primitives.rs:123-127
        if quantized.shape[0] == 0:
            return
        if any(s < 0 for s in quantized.strides):
            quantized = np.ascontiguousarray(quantized)
        if any(s < 0 for s in reconstructions.strides):
            raise ValueError("negative output strides are not supported")
        crs, ccs = _estrides(quantized)
        ors, ocs = _estrides(reconstructions)
        rc = _lib.lib().pqhip_reconstruct_batch_f32(self._cb(), quantized.ctypes.data,
                                                   quantized.itemsize, quantized.shape[0], crs, ccs,
                                                   reconstructions.ctypes.data, ors, ocs)
        if rc == _lib.ECODE_RANGE:
            raise PanicError("ndarray: index out of bounds")                    # primitives.rs:146
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_reconstruct_batch_f32")

    def reconstruct(self, quantized):
        """`reconstruct` (traits.rs:133-141 -> pq.rs:329-343); single vector, host."""
        quantized = np.asarray(quantized)
        if quantized.ndim != 1 or quantized.shape[0] != self.quantized_len():
            raise PanicError("Quantization length does not match number of subquantizers")
        K = self.n_quantizer_centroids()
        if (quantized.astype(np.int64) < 0).any() or (quantized.astype(np.uint64) >= K).any():
            raise PanicError("ndarray: index out of bounds")
        rec = np.concatenate([self._quantizers[m, int(c)] for m, c in enumerate(quantized)])
        if self._projection is not None:
            # reconstruction.dot(&projection.t()): per output k a contiguous dot with row P[k,:]
            rec = _unrolled_dot_rows(self._projection, rec.astype(np.float32))
        return rec.astype(np.float32)

    # ---- device-resident variants (torch tensors in HBM; used by bench.py and the GPU tests) ----
    def _slot_for(self, tensor):
        ctx = self._ctx or default_ctx()
        dev = tensor.device.index or 0
        if ctx.devices is None:
            return dev
        return ctx.devices.index(dev)

    def quantize_batch_device(self, x, out=None, stream=None):
        """x: CUDA float32 tensor [n, d] (unit column stride) -> codes tensor [n, M] (uint8; int32 when K > 256).
        Asynchronous on torch's current stream unless `stream` (a raw hipStream_t int) is given."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
        if x.shape[1] != self.reconstructed_len():
            raise PanicError("Quantizer and vector length mismatch")
        if x.stride(1) != 1:
            x = x.contiguous()
        if out is None:
            # u8 codes, or 32-bit codes (int32 tensor, values < 2^31) when K > 256
            dt = torch.uint8 if self.n_quantizer_centroids() <= 256 else torch.int32
            out = torch.empty((x.shape[0], self.quantized_len()), dtype=dt, device=x.device)
        if tuple(out.shape) != (x.shape[0], self.quantized_len()):
            raise PanicError("Quantized matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (x.shape[0], self.quantized_len(), out.shape[0], out.shape[1]))
        # index type I of quantize_batch::<I, _>: uint8, or int16 / int32 / int64 tensors as 2- / 4- / 8-byte containers
        # (u16 / u32 / u64 bit patterns)
        assert out.is_cuda and out.dtype in (torch.uint8, torch.int16, torch.int32, torch.int64) and out.stride(1) == 1
        code_bytes = out.element_size()
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = _lib.lib().pqhip_quantize_batch_f32_dev(cb, self._slot_for(x), x.data_ptr(), x.shape[0],
                                                    x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1]),
                                                    out.data_ptr(), code_bytes,
                                                    out.stride(0) if out.shape[0] > 1 else max(out.stride(0), out.shape[1]),
                                                    ctypes.c_void_p(stream))
        if rc == _lib.EINDEX_WIDTH:
            raise PanicError("Cannot store centroids in quantizer index type")
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_quantize_batch_f32_dev")
        return out

    def reconstruct_batch_device(self, codes, out=None, stream=None, check=True):
        """codes: CUDA uint8 (or int16 / int32 / int64 = u16 / u32 / u64 containers) tensor [n, M] -> float32 tensor [n, d].
        check=True (default) synchronises the stream and raises the reference's index panic
        (primitives.rs:146) when a code >= K was met; check=False leaves the call asynchronous and the
        violation pending on the stream's flag (a later check=True call on that stream reports it)."""
        import torch
        assert codes.is_cuda and codes.dtype in (torch.uint8, torch.int16, torch.int32, torch.int64) and codes.dim() == 2
        if codes.shape[1] != self.quantized_len():
            raise PanicError("Quantization length does not match number of subquantizers")
        if codes.stride(1) != 1:
            codes = codes.contiguous()
        if out is None:
            out = torch.empty((codes.shape[0], self.reconstructed_len()), dtype=torch.float32,
                              device=codes.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.stride(1) == 1
        if tuple(out.shape) != (codes.shape[0], self.reconstructed_len()):
            raise PanicError("Reconstructions matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (codes.shape[0], self.reconstructed_len(), out.shape[0], out.shape[1]))
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(codes.device).cuda_stream
        slot = self._slot_for(codes)
        rc = _lib.lib().pqhip_reconstruct_batch_f32_dev(
            cb, slot, codes.data_ptr(), codes.element_size(), codes.shape[0],
            codes.stride(0) if codes.shape[0] > 1 else max(codes.stride(0), codes.shape[1]),
            out.data_ptr(), out.stride(0) if out.shape[0] > 1 else max(out.stride(0), out.shape[1]),
            ctypes.c_void_p(stream))
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_reconstruct_batch_f32_dev")
        if check:
            rc = _lib.lib().pqhip_check_codes_dev(cb, slot, ctypes.c_void_p(stream))
            if rc == _lib.ECODE_RANGE:
                raise PanicError("ndarray: index out of bounds")
            if rc != _lib.OK:
                raise _lib.PqHipError(rc, "pqhip_check_codes_dev")
        return out

    def reconstruct_rows_device(self, codes, rows, scales=None, out=None, stream=None, check=True):
        """Lookup path of a resident quantized matrix ("next" row):
        `reconstruct_batch(codes.select(Axis(0), rows)) * scales.select(rows)` in one pass.
        codes: CUDA uint8 [N, M]; rows: CUDA int64 [n]; scales: None or CUDA float32 [N]."""
        import torch
        assert codes.is_cuda and codes.dtype == torch.uint8 and codes.dim() == 2
        assert rows.is_cuda and rows.dtype == torch.int64 and rows.dim() == 1 and rows.is_contiguous()
        if codes.shape[1] != self.quantized_len():
            raise PanicError("Quantization length does not match number of subquantizers")
        if codes.stride(1) != 1:
            codes = codes.contiguous()
        if scales is not None:
            assert scales.is_cuda and scales.dtype == torch.float32 and scales.is_contiguous()
            if scales.shape != (codes.shape[0],):
                raise PanicError("scales must hold one value per code row")
        n = rows.shape[0]
        if out is None:
            out = torch.empty((n, self.reconstructed_len()), dtype=torch.float32, device=codes.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.stride(1) == 1
        if tuple(out.shape) != (n, self.reconstructed_len()):
            raise PanicError("Reconstructions matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (n, self.reconstructed_len(), out.shape[0], out.shape[1]))
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(codes.device).cuda_stream
        slot = self._slot_for(codes)
        rc = _lib.lib().pqhip_reconstruct_rows_f32_dev(
            cb, slot, codes.data_ptr(), 1, codes.shape[0],
            codes.stride(0) if codes.shape[0] > 1 else max(codes.stride(0), codes.shape[1]),
            rows.data_ptr(), n, scales.data_ptr() if scales is not None else None,
            out.data_ptr(), out.stride(0) if n > 1 else max(out.stride(0), out.shape[1]),
            ctypes.c_void_p(stream))
        if rc == _lib.ECODE_RANGE:
            raise PanicError("ndarray: index out of bounds")
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_reconstruct_rows_f32_dev")
        if check:
            rc = _lib.lib().pqhip_check_codes_dev(cb, slot, ctypes.c_void_p(stream))
            if rc == _lib.ECODE_RANGE:
                raise PanicError("ndarray: index out of bounds")
            if rc != _lib.OK:
                raise _lib.PqHipError(rc, "pqhip_check_codes_dev")
        return out

    @staticmethod
    def interleave_records(codes, scales, record_bytes=None):
        """Resident-matrix layout with ONE cache line per lookup: CUDA uint8 [N, record_bytes] records holding the M code
        bytes of a row at offset 0 and its f32 scale at the next multiple of 16 bytes (32-byte records at M = 15).
        Returns (records, scale_offset_bytes)."""
        import torch
        assert codes.is_cuda and codes.dtype == torch.uint8 and codes.dim() == 2
        assert scales.is_cuda and scales.dtype == torch.float32 and scales.shape == (codes.shape[0],)
        m = codes.shape[1]
        off = (m + 15) // 16 * 16
        rb = record_bytes or max(32, 1 << (off + 4 - 1).bit_length())
        assert rb % 4 == 0 and off + 4 <= rb
        rec = torch.zeros((codes.shape[0], rb), dtype=torch.uint8, device=codes.device)
        rec[:, :m] = codes
        rec[:, off:off + 4] = scales.contiguous().view(torch.uint8).view(-1, 4)
        return rec, off

    def reconstruct_records_device(self, records, scale_offset, rows, out=None, stream=None, check=True):
        """reconstruct_rows_device over interleaved records (see interleave_records): same results, one 128-byte
        line per lookup instead of two (pqhip_reconstruct_rows_records_f32_dev)."""
        import torch
        assert records.is_cuda and records.dtype == torch.uint8 and records.dim() == 2 and records.is_contiguous()
        assert rows.is_cuda and rows.dtype == torch.int64 and rows.dim() == 1 and rows.is_contiguous()
        n = rows.shape[0]
        if out is None:
            out = torch.empty((n, self.reconstructed_len()), dtype=torch.float32, device=records.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.stride(1) == 1
        if tuple(out.shape) != (n, self.reconstructed_len()):
            raise PanicError("Reconstructions matrix has incorrect shape, expected: (%d, %d), got: (%d, %d)"
                             % (n, self.reconstructed_len(), out.shape[0], out.shape[1]))
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(records.device).cuda_stream
        slot = self._slot_for(records)
        rc = _lib.lib().pqhip_reconstruct_rows_records_f32_dev(
            cb, slot, records.data_ptr(), 1, records.shape[0], records.shape[1], scale_offset, rows.data_ptr(), n,
            out.data_ptr(), out.stride(0) if n > 1 else max(out.stride(0), out.shape[1]), ctypes.c_void_p(stream))
        if rc == _lib.ECODE_RANGE:
            raise PanicError("ndarray: index out of bounds")
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_reconstruct_rows_records_f32_dev")
        if check:
            rc = _lib.lib().pqhip_check_codes_dev(cb, slot, ctypes.c_void_p(stream))
            if rc == _lib.ECODE_RANGE:
                raise PanicError("ndarray: index out of bounds")
            if rc != _lib.OK:
                raise _lib.PqHipError(rc, "pqhip_check_codes_dev")
        return out

    # ---- "next" row: asymmetric distance computation over resident codes -----
    def adc_tables_device(self, queries, stream=None):
        """queries: CUDA float32 [d] or [nq, d] -> tables [M, K] or [nq, M, K] with
        tables[q, m, j] = `y_q[m].squared_euclidean_distance(quantizers[m])[j]` (linalg.rs:118-148;
        y = query.dot(projection) first for OPQ, pq.rs:293)."""
        import torch
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.dim() in (1, 2)
        single = queries.dim() == 1
        q2 = queries[None] if single else queries
        if q2.shape[1] != self.reconstructed_len():
            raise PanicError("Quantizer and vector length mismatch")
        if q2.stride(1) != 1:
            q2 = q2.contiguous()
        M, K = self.quantized_len(), self.n_quantizer_centroids()
        out = torch.empty((q2.shape[0], M, K), dtype=torch.float32, device=queries.device)
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(queries.device).cuda_stream
        rc = _lib.lib().pqhip_adc_tables_f32_dev(cb, self._slot_for(queries), q2.data_ptr(), q2.shape[0],
                                                q2.stride(0) if q2.shape[0] > 1 else max(q2.stride(0), q2.shape[1]),
                                                out.data_ptr(), ctypes.c_void_p(stream))
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_adc_tables_f32_dev")
        return out[0] if single else out

    def adc_scan_device(self, codes, tables, out=None, stream=None, check=False):
        """codes: CUDA uint8 (or int32 when K > 256) [n, M]; tables: [M, K] or [nq, M, K] from
        adc_tables_device -> distances [n] or [nq, n]: out[q, i] = sum_m tables[q, m, codes[i, m]]
        (sequential f32 sum over m)."""
        import torch
        assert codes.is_cuda and codes.dtype in (torch.uint8, torch.int32) and codes.dim() == 2
        assert tables.is_cuda and tables.dtype == torch.float32 and tables.is_contiguous()
        M, K = self.quantized_len(), self.n_quantizer_centroids()
        if codes.shape[1] != M:
            raise PanicError("Quantization length does not match number of subquantizers")
        single = tables.dim() == 2
        if tuple(tables.shape[-2:]) != (M, K):
            raise PanicError("lookup tables must be [.., %d, %d]" % (M, K))
        nq = 1 if single else tables.shape[0]
        if codes.stride(1) != 1:
            codes = codes.contiguous()
        n = codes.shape[0]
        if out is None:
            out = torch.empty((n,) if single else (nq, n), dtype=torch.float32, device=codes.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == nq * n
        cb = self._cb()
        if stream is None:
            stream = torch.cuda.current_stream(codes.device).cuda_stream
        slot = self._slot_for(codes)
        rc = _lib.lib().pqhip_adc_scan_f32_dev(cb, slot, tables.data_ptr(), nq, codes.data_ptr(), codes.element_size(), n,
                                              codes.stride(0) if n > 1 else max(codes.stride(0), M),
                                              out.data_ptr(), n, ctypes.c_void_p(stream))
        if rc != _lib.OK:
            raise _lib.PqHipError(rc, "pqhip_adc_scan_f32_dev")
        if check:
            rc = _lib.lib().pqhip_check_codes_dev(cb, slot, ctypes.c_void_p(stream))
            if rc == _lib.ECODE_RANGE:
                raise PanicError("ndarray: index out of bounds")
            if rc != _lib.OK:
                raise _lib.PqHipError(rc, "pqhip_check_codes_dev")
        return out
