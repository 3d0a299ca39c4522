"""ctypes loader for the CPU oracle (oracle/pq_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under reductive_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PQO_SANITIZED=1: load the ASan/UBSan build instead (the process must have libasan preloaded)
_SAN = os.environ.get("PQO_SANITIZED") == "1"
_SO = os.path.join(_HERE, "libpq_oracle_san.so" if _SAN else "libpq_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p


def build(force=False):
    src = os.path.join(_HERE, "pq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"] + (["libpq_oracle_san.so"] if _SAN else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.pqo_dot_unrolled.restype = ctypes.c_float
        L.pqo_dot_unrolled.argtypes = [_fp, _fp, _i64]
        L.pqo_first_min.restype = _i64
        L.pqo_first_min.argtypes = [_fp, _i64]
        L.pqo_sqdist_mm.restype = None
        L.pqo_sqdist_mm.argtypes = [_fp, _i64, _i64, _fp, _i64, _i64, _fp]
        L.pqo_cluster_assignments.restype = None
        L.pqo_cluster_assignments.argtypes = [_fp, _i64, _i64, _fp, _i64, _i64,
                                              ctypes.POINTER(_i64)]
        L.pqo_rotate.restype = None
        L.pqo_rotate.argtypes = [_fp, _i64, _i64, _i64, _i64, _fp, _fp]
        L.pqo_quantize_batch.restype = ctypes.c_int
        L.pqo_quantize_batch.argtypes = [_fp, _i64, _i64, _i64, _fp, _fp, _i64, _i64, _i64, _vp,
                                         ctypes.c_int, _i64, _i64, ctypes.c_int]
        L.pqo_reconstruct_batch.restype = ctypes.c_int
        L.pqo_reconstruct_batch.argtypes = [_fp, _i64, _i64, _i64, _fp, _vp, ctypes.c_int, _i64,
                                            _i64, _i64, _fp, _i64, _i64]
        L.pqo_quantize_vector.restype = ctypes.c_int
        L.pqo_quantize_vector.argtypes = [_fp, _i64, _i64, _i64, _fp, _fp, ctypes.POINTER(_i64)]
        L.pqo_update_centroids.restype = None
        L.pqo_update_centroids.argtypes = [_fp, _i64, _i64, _fp, _i64, _i64, _i64,
                                           ctypes.POINTER(_i64)]
        L.pqo_mean_squared_error.restype = ctypes.c_float
        L.pqo_mean_squared_error.argtypes = [_fp, _i64, _fp, _i64, _i64, _i64, ctypes.POINTER(_i64)]
        L.pqo_kmeans_iterations.restype = ctypes.c_int
        L.pqo_kmeans_iterations.argtypes = [_fp, _i64, _i64, _i64, _fp, _i64, _i64, _i64,
                                            ctypes.c_int, _fp, ctypes.c_int]
        L.pqo_at_dot_b.restype = None
        L.pqo_at_dot_b.argtypes = [_fp, _i64, _i64, _i64, _fp, _i64, _i64, _fp, ctypes.c_int]
        L.pqo_opq_train_step.restype = ctypes.c_int
        L.pqo_opq_train_step.argtypes = [_fp, _i64, _i64, _i64, _fp, _fp, _i64, _i64, _i64, _fp, ctypes.c_int]
        L.pqo_adc_tables.restype = ctypes.c_int
        L.pqo_adc_tables.argtypes = [_fp, _i64, _i64, _i64, _fp, _fp, _fp]
        L.pqo_adc_scan.restype = ctypes.c_int
        L.pqo_adc_scan.argtypes = [_fp, _i64, _i64, _vp, ctypes.c_int, _i64, _i64, _i64, _fp]
        _lib = L
    return _lib


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def _estrides(a):
    assert all(s % a.itemsize == 0 for s in a.strides)
    return [s // a.itemsize for s in a.strides]


def dot_unrolled(x, y):
    x, y = _f32c(x), _f32c(y)
    return np.float32(lib().pqo_dot_unrolled(_p(x), _p(y), x.size))


def first_min(d):
    d = _f32c(d)
    return int(lib().pqo_first_min(_p(d), d.size))


def sqdist(x, c):
    """linalg.rs:150-180 for row-contiguous x [n,dd] and c [k,dd] -> [n,k]."""
    x, c = _f32c(x), _f32c(c)
    out = np.empty((x.shape[0], c.shape[0]), np.float32)
    lib().pqo_sqdist_mm(_p(x), x.shape[0], x.shape[1], _p(c), c.shape[0], c.shape[1], _p(out))
    return out


def cluster_assignments(centroids, x):
    centroids, x = _f32c(centroids), _f32c(x)
    out = np.empty(x.shape[0], np.int64)
    lib().pqo_cluster_assignments(_p(centroids), centroids.shape[0], centroids.shape[1], _p(x),
                                  x.shape[0], x.shape[1], out.ctypes.data_as(ctypes.POINTER(_i64)))
    return out


def rotate(x, P):
    x = np.asarray(x, dtype=np.float32)
    P = _f32c(P)
    rs, cs = _estrides(x)
    out = np.empty(x.shape, np.float32)
    lib().pqo_rotate(_p(x), x.shape[0], x.shape[1], rs, cs, _p(P), _p(out))
    return out


def quantize_batch(quantizers, x, projection=None, dtype=np.uint8, n_threads=1, out=None):
    """Pq::quantize_batch (pq.rs:256-283).  quantizers [M,K,dsub]; x [n,d] any strides."""
    q = _f32c(quantizers)
    M, K, dsub = q.shape
    x = np.asarray(x, dtype=np.float32)
    assert x.ndim == 2 and x.shape[1] == M * dsub, "Quantizer and vector length mismatch"
    P = _f32c(projection) if projection is not None else None
    if out is None:
        out = np.zeros((x.shape[0], M), dtype=dtype)
    assert out.shape == (x.shape[0], M)
    rs, cs = _estrides(x) if x.size else (x.shape[1], 1)
    ors, ocs = _estrides(out) if out.size else (M, 1)
    rc = lib().pqo_quantize_batch(_p(q), M, K, dsub, _p(P), _p(x), x.shape[0], rs, cs,
                                  out.ctypes.data_as(_vp), out.itemsize, ors, ocs, n_threads)
    assert rc == 0, rc
    return out


def reconstruct_batch(quantizers, codes, projection=None):
    """Reconstruct::reconstruct_batch (traits.rs:109-117 -> pq.rs:309-327)."""
    q = _f32c(quantizers)
    M, K, dsub = q.shape
    codes = np.asarray(codes)
    assert codes.ndim == 2 and codes.shape[1] == M
    assert codes.dtype.kind == "u" or codes.dtype.kind == "i"
    P = _f32c(projection) if projection is not None else None
    out = np.zeros((codes.shape[0], M * dsub), np.float32)
    crs, ccs = _estrides(codes) if codes.size else (M, 1)
    rc = lib().pqo_reconstruct_batch(_p(q), M, K, dsub, _p(P), codes.ctypes.data_as(_vp),
                                     codes.itemsize, codes.shape[0], crs, ccs, _p(out),
                                     M * dsub, 1)
    if rc == 2:
        raise IndexError("code >= K (reference: ndarray index_axis panic, primitives.rs:146)")
    assert rc == 0, rc
    return out


def quantize_vector(quantizers, x, projection=None):
    q = _f32c(quantizers)
    M, K, dsub = q.shape
    x = _f32c(x)
    assert x.shape == (M * dsub,), "Quantizer and vector length mismatch"
    P = _f32c(projection) if projection is not None else None
    out = np.empty(M, np.int64)
    lib().pqo_quantize_vector(_p(q), M, K, dsub, _p(P), _p(x),
                              out.ctypes.data_as(ctypes.POINTER(_i64)))
    return out


# ---- "next" row: the k-means step of training (kmeans.rs:166-198, 308-360) ---------------------
def update_centroids(centroids_shape, x, assignments):
    """kmeans.rs:166-198 for instances along axis 0; returns the new [K, dim] centroids."""
    K, dim = centroids_shape
    x = np.asarray(x, dtype=np.float32)
    a = np.ascontiguousarray(assignments, dtype=np.int64)
    assert x.shape == (a.shape[0], dim), "The number of assignments should be equal to the number of instances."
    out = np.empty((K, dim), np.float32)
    rs, cs = _estrides(x) if x.size else (dim, 1)
    lib().pqo_update_centroids(_p(out), K, dim, _p(x), x.shape[0], rs, cs,
                               a.ctypes.data_as(ctypes.POINTER(_i64)))
    return out


def mean_squared_error(centroids, x, assignments):
    """kmeans.rs:329-360 for instances along axis 0."""
    c = _f32c(centroids)
    x = np.asarray(x, dtype=np.float32)
    a = np.ascontiguousarray(assignments, dtype=np.int64)
    rs, cs = _estrides(x)
    return np.float32(lib().pqo_mean_squared_error(_p(c), c.shape[1], _p(x), x.shape[0], rs, cs,
                                                   a.ctypes.data_as(ctypes.POINTER(_i64))))


def kmeans_iterations(quantizers, x, n_iterations=1, n_threads=1):
    """`n_iterations` x kmeans_iteration (kmeans.rs:308-327) on every subquantizer's columns
    (pq.rs:176 / opq.rs:227-245).  Returns (new quantizers [M,K,dsub], last loss [M])."""
    q = _f32c(quantizers).copy()
    M, K, dsub = q.shape
    x = np.asarray(x, dtype=np.float32)
    assert x.ndim == 2 and x.shape[1] == M * dsub, "Centroid and instance lengths differ."
    rs, cs = _estrides(x)
    loss = np.zeros(M, np.float32)
    rc = lib().pqo_kmeans_iterations(_p(q), M, K, dsub, _p(x), x.shape[0], rs, cs, n_iterations,
                                     _p(loss), n_threads)
    assert rc == 0, rc
    return q, loss


# ---- OPQ training iteration without LAPACK (opq.rs:156-195) --------------------------------------
def at_dot_b(a, b, n_threads=1):
    """`a.t().dot(&b)` for row-major a [n, da], b [n, db] (rule 2 with k over the rows)."""
    a, b = _f32c(a), _f32c(b)
    assert a.shape[0] == b.shape[0]
    out = np.zeros((a.shape[1], b.shape[1]), np.float32)
    lib().pqo_at_dot_b(_p(a), a.shape[0], a.shape[1], a.shape[1], _p(b), b.shape[1], b.shape[1], _p(out), n_threads)
    return out


def opq_train_step(quantizers, projection, x, n_threads=1):
    """Device part of Opq::train_iteration: returns (updated quantizers, cross = x^T . reconstructed)."""
    q = _f32c(quantizers).copy()
    M, K, dsub = q.shape
    P = _f32c(projection)
    x = np.asarray(x, dtype=np.float32)
    rs, cs = _estrides(x)
    cross = np.zeros((M * dsub, M * dsub), np.float32)
    rc = lib().pqo_opq_train_step(_p(q), M, K, dsub, _p(P), _p(x), x.shape[0], rs, cs, _p(cross), n_threads)
    assert rc == 0, rc
    return q, cross


# ---- "next" row rank 4: asymmetric distance computation over codes -------------------------------
def adc_tables(quantizers, query, projection=None):
    """tables [M, K]: tables[m, j] = squared distance of the query's m-th sub-vector to centroid j,
    evaluated as `instance.squared_euclidean_distance(centroids)` (linalg.rs:118-148); the query is
    rotated first (pq.rs:293) when a projection is given.  A [nq, d] query matrix gives [nq, M, K]."""
    q = _f32c(quantizers)
    M, K, dsub = q.shape
    y = _f32c(query)
    P = _f32c(projection) if projection is not None else None
    single = y.ndim == 1
    y2 = y[None] if single else y
    assert y2.shape[1] == M * dsub, "Quantizer and vector length mismatch"
    out = np.empty((y2.shape[0], M, K), np.float32)
    for i in range(y2.shape[0]):
        yi = np.ascontiguousarray(y2[i])
        lib().pqo_adc_tables(_p(q), M, K, dsub, _p(P), _p(yi), _p(out[i]))
    return out[0] if single else out


def adc_scan(tables, codes):
    """dist[i] = sum_m tables[m, codes[i, m]] (sequential f32 sum over m from +0); tables [M, K] -> [n],
    tables [nq, M, K] -> [nq, n]."""
    t = _f32c(tables)
    single = t.ndim == 2
    t3 = t[None] if single else t
    codes = np.asarray(codes)
    assert codes.ndim == 2 and codes.shape[1] == t3.shape[1]
    crs, ccs = _estrides(codes) if codes.size else (codes.shape[1], 1)
    out = np.empty((t3.shape[0], codes.shape[0]), np.float32)
    for i in range(t3.shape[0]):
        rc = lib().pqo_adc_scan(_p(np.ascontiguousarray(t3[i])), t3.shape[1], t3.shape[2], codes.ctypes.data_as(_vp),
                                codes.itemsize, codes.shape[0], crs, ccs, _p(out[i]))
        if rc == 2:
            raise IndexError("code >= K")
        assert rc == 0, rc
    return out[0] if single else out
