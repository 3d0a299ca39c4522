"""Deterministic synthetic data for parity tests (integer-only hashing -> identical bits on any host).

normalish(): approximately N(0,1) values built from four 16-bit uniform fields of a SplitMix64
hash of the element index (Irwin-Hall), so no libm call is involved and the float32 bits do not
depend on the platform.  Matches the *shape* of the reference's bench data (benches/pq.rs:9 uses
N(0,1) f32); exact distribution is irrelevant to parity.
"""
import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _hash(seed, n):
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) + np.uint64(1))
        return splitmix64(np.arange(n, dtype=np.uint64) + base)


def normalish(seed, shape):
    n = int(np.prod(shape))
    h = _hash(seed, n)
    s = np.zeros(n, np.int64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    s -= 131070
    return (s.astype(np.float32) / np.float32(37837.0)).reshape(shape)


def uniform01(seed, shape):
    n = int(np.prod(shape))
    h = _hash(seed, n)
    return ((h >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24)).reshape(shape)


def codes_u8(seed, shape, k):
    n = int(np.prod(shape))
    h = _hash(seed, n)
    return ((h >> np.uint64(33)) % np.uint64(k)).astype(np.uint8).reshape(shape)


def orthonormal(seed, d):
    """Random orthonormal [d,d] (float64 QR of a normalish matrix, cast to f32).

    QR goes through LAPACK, so bits may differ across hosts: fixtures that use a projection
    store the matrix itself rather than regenerating it."""
    a = normalish(seed, (d, d)).astype(np.float64)
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    return np.ascontiguousarray(q.astype(np.float32))


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()
