"""Independent, exact (rational-arithmetic) float32 model of CANON-F32 for SMALL cases.

Used only to validate oracle/pq_oracle.c: every f32 operation is performed exactly on
fractions.Fraction and rounded ONCE to binary32 (round-to-nearest-even), so there is no
dependence on the host's FMA unit, the C compiler's contraction rules or numpy's summation
order.  Pure-Python loops: keep inputs tiny.
"""
from fractions import Fraction
import math

import numpy as np

KC = 256


def _round_f32(fr):
    """Correctly rounded Fraction -> binary32 (returned as a Python float holding that value)."""
    if fr == 0:
        return 0.0
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    # exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    if Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)                       # subnormal range shares the scale of 2^-126
    scale = Fraction(2) ** (e - 23)
    q = a / scale
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    val = Fraction(n) * scale
    if val >= Fraction(2) ** 128:
        return sign * math.inf
    return sign * float(val)               # exact: val is a binary32 value


def _F(v):
    return Fraction(float(v))


def fadd(a, b):
    if not (math.isfinite(a) and math.isfinite(b)):
        return float(np.float32(a) + np.float32(b))
    return _round_f32(_F(a) + _F(b))


def fsub(a, b):
    if not (math.isfinite(a) and math.isfinite(b)):
        return float(np.float32(a) - np.float32(b))
    return _round_f32(_F(a) - _F(b))


def fmul(a, b):
    if not (math.isfinite(a) and math.isfinite(b)):
        return float(np.float32(a) * np.float32(b))
    return _round_f32(_F(a) * _F(b))


def fma(a, b, c):
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return float(np.float32(np.float64(a) * np.float64(b) + np.float64(c)))
    return _round_f32(_F(a) * _F(b) + _F(c))


def dot_unrolled(x, y):
    """ndarray numeric_util::unrolled_dot (CANON-F32 rule 1)."""
    n = len(x)
    p = [0.0] * 8
    i = 0
    while n - i >= 8:
        for l in range(8):
            p[l] = fadd(p[l], fmul(x[i + l], y[i + l]))
        i += 8
    s = 0.0
    s = fadd(s, fadd(p[0], p[4]))
    s = fadd(s, fadd(p[1], p[5]))
    s = fadd(s, fadd(p[2], p[6]))
    s = fadd(s, fadd(p[3], p[7]))
    while i < n:
        s = fadd(s, fmul(x[i], y[i]))
        i += 1
    return s


def gemm_dot(x, y):
    """CANON-F32 rule 2: fmaf chain from +0, restarted every KC, blocks summed left to right."""
    n = len(x)
    total = None
    for kb in range(0, n, KC):
        ab = 0.0
        for k in range(kb, min(kb + KC, n)):
            ab = fma(x[k], y[k], ab)
        total = ab if total is None else fadd(total, ab)
    return 0.0 if total is None else total


def of_less(a, b):
    if math.isnan(a):
        return False
    if math.isnan(b):
        return True
    return a < b


def first_min(d):
    best = 0
    for j in range(1, len(d)):
        if of_less(d[j], d[best]):
            best = j
    return best


def sqdist(x, c):
    x = np.asarray(x, np.float32)
    c = np.asarray(c, np.float32)
    xx = [dot_unrolled(list(r), list(r)) for r in x]
    cc = [dot_unrolled(list(r), list(r)) for r in c]
    out = np.empty((len(x), len(c)), np.float32)
    for i in range(len(x)):
        for j in range(len(c)):
            dp = gemm_dot(list(x[i]), list(c[j]))
            out[i, j] = np.float32(fsub(fadd(xx[i], cc[j]), fadd(dp, dp)))
    return out


def rotate(x, P):
    x = np.asarray(x, np.float32)
    P = np.asarray(P, np.float32)
    out = np.empty_like(x)
    for i in range(x.shape[0]):
        for c in range(P.shape[1]):
            out[i, c] = np.float32(gemm_dot(list(x[i]), list(P[:, c])))
    return out


def quantize_batch(quantizers, x, projection=None):
    q = np.asarray(quantizers, np.float32)
    M, K, dsub = q.shape
    x = np.asarray(x, np.float32)
    if projection is not None:
        x = rotate(x, projection)
    codes = np.zeros((x.shape[0], M), np.int64)
    for m in range(M):
        d = sqdist(x[:, m * dsub:(m + 1) * dsub], q[m])
        for i in range(x.shape[0]):
            codes[i, m] = first_min([float(v) for v in d[i]])
    return codes


def reconstruct_batch(quantizers, codes, projection=None):
    q = np.asarray(quantizers, np.float32)
    M, K, dsub = q.shape
    codes = np.asarray(codes)
    out = np.zeros((codes.shape[0], M * dsub), np.float32)
    for i in range(codes.shape[0]):
        for m in range(M):
            out[i, m * dsub:(m + 1) * dsub] = q[m, int(codes[i, m])]
    if projection is not None:
        out = rotate(out, np.asarray(projection, np.float32).T)
    return out


def fdiv(a, b):
    if not (math.isfinite(a) and math.isfinite(b)) or b == 0:
        with np.errstate(all="ignore"):
            return float(np.float32(a) / np.float32(b))
    return _round_f32(_F(a) / _F(b))


def update_centroids(K, x, assignments):
    """kmeans.rs:166-198: zero fill, sequential row-order adds, f32 counts, IEEE division."""
    n, dim = x.shape
    c = [[0.0] * dim for _ in range(K)]
    cnt = [0.0] * K
    for i in range(n):
        a = int(assignments[i])
        for e in range(dim):
            c[a][e] = fadd(c[a][e], float(x[i, e]))
        cnt[a] = fadd(cnt[a], 1.0)
    for k in range(K):
        if cnt[k] > 0:
            c[k] = [fdiv(v, cnt[k]) for v in c[k]]
    return np.array(c, np.float32)


def mean_squared_error(centroids, x, assignments):
    """kmeans.rs:329-360: one sequential fold of (c - x)^2 over all elements, / (n*dim)."""
    n, dim = x.shape
    sse = 0.0
    for i in range(n):
        a = int(assignments[i])
        for e in range(dim):
            err = fsub(float(centroids[a, e]), float(x[i, e]))
            sse = fadd(sse, fmul(err, err))
    return np.float32(fdiv(sse, float(np.float32(n * dim))))
