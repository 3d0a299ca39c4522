"""Candidate tables of the 2-float sub-vector encode kernel (reductive_amd/csrc/vor2_prep.h, kernels_vor2.hip.h), checked on
the CPU: the tables are built by the library's host code (no GPU needed), the kernel's walk -- cell from the float operations
fl(fl(x - lo) * inv), the cell's list, strict `<` over the listed centroids in ascending order with the CANON-F32 distance --
is restated in numpy, and the result must be the oracle's code (oracle/pq_oracle: every centroid evaluated) for every point,
i.e. the winner is always on the list.  Reference shape: pq.rs:431-440 (d = 20, M = 10, K = 128)."""
import numpy as np
import pytest

import synth
from oracle import pq_oracle as orc


@pytest.fixture(scope="module")
def ra():
    import os
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    return reductive_amd


def walk(words, off, q, x):
    """The kernel's walk, vectorised over rows: returns codes [n][M] with -1 where the row takes the exact path."""
    M, K, dsub = q.shape
    n = x.shape[0]
    out = np.full((n, M), -1, dtype=np.int64)
    lists_seen = []
    f32 = np.float32
    for m in range(M):
        r = words[off[m]:off[m + 1]]
        hf = r.view(np.float32)
        x0 = x[:, dsub * m].astype(f32)
        x1 = x[:, 2 * m + 1].astype(f32) if dsub == 2 else np.zeros(n, dtype=f32)   # 1 float: the second coordinate is 0
        with np.errstate(all="ignore"):
            t0 = (x0 - hf[0]) * hf[1]
            t1 = (x1 - hf[2]) * hf[3]
            u0 = (x0 - hf[5]) * hf[6]
            u1 = (x1 - hf[7]) * hf[8]
            in_f = (t0 >= 0) & (t0 < hf[4]) & (t1 >= 0) & (t1 < hf[16])
            in_c = (u0 >= 0) & (u0 < hf[9]) & (u1 >= 0) & (u1 < hf[17])
        G, CG = int(r[13]), int(r[14])
        s0 = np.where(in_f, t0, np.where(in_c, u0, 0)).astype(np.int64)
        s1 = np.where(in_f, t1, np.where(in_c, u1, 0)).astype(np.int64)
        ci = np.where(in_f, int(r[10]) + s0 * G + s1, int(r[11]) + s0 * CG + s1)
        c16 = r.view(np.uint16)
        cw = c16[ci].astype(np.int64)                         # 16-bit cell entries: list offset in words << 4 | words - 1
        split = in_f & ((cw & 15) == 15)                      # a subdivided fine cell: the entry of the point's half cell
        with np.errstate(all="ignore"):
            f0 = (t0 - np.floor(np.where(in_f, t0, 0)).astype(f32)).astype(f32)
            f1 = (t1 - np.floor(np.where(in_f, t1, 0)).astype(f32)).astype(f32)
        si = int(r[15]) + 4 * (cw >> 4) + 2 * (f0 >= np.float32(0.5)) + (f1 >= np.float32(0.5))
        cw = np.where(split, c16[np.where(split, si, 0)].astype(np.int64), cw)
        ok = in_f | in_c
        nwords = np.where(ok, (cw & 15) + 1, 0)
        lofs = cw >> 4
        lw = r.view(np.uint8)[int(r[12]):]                    # the lists (whole words, padded with the last index)
        xx = (x0 * x0 + x1 * x1).astype(f32)
        best = np.full(n, np.inf, dtype=f32)
        bj = np.full(n, -1, dtype=np.int64)
        c = q[m].astype(f32)
        if dsub == 1:
            c = np.concatenate([c, np.zeros_like(c)], axis=1)
        cc = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]).astype(f32)
        for i in range(int(nwords.max()) if n else 0):
            act = i < nwords
            for e in range(4):                                # the kernel evaluates all four entries of a word
                j = lw[np.minimum(4 * (lofs + i) + e, lw.size - 1)].astype(np.int64)
                # dp = fma(x1, c1, fl(x0 c0)); d = fma(dp, -2, fl(xx + cc)): exact in float64 (24-bit products), then one rounding
                p0 = (x0 * c[j, 0]).astype(f32)
                dp = (x1.astype(np.float64) * c[j, 1].astype(np.float64) + p0.astype(np.float64)).astype(f32)
                t = (xx + cc[j]).astype(f32)
                d = (t.astype(np.float64) - 2.0 * dp.astype(np.float64)).astype(f32)
                take = act & (d < best)
                best = np.where(take, d, best)
                bj = np.where(take, j, bj)
        out[:, m] = bj
        lists_seen.append(4.0 * nwords[ok].mean() if ok.any() else 0.0)
    return out, lists_seen


def check(ra, q, x, min_fast=0.0):
    t = ra.vor2_tables(q)
    assert t is not None
    words, off = t
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8).astype(np.int64)
        got, mean_list = walk(words, off, q, x)
    fast = got >= 0
    assert (got[fast] == want[fast]).all(), np.argwhere(fast & (got != want))[:5]
    assert fast.mean() >= min_fast, fast.mean()
    return mean_list


def test_reference_test_shape_gaussian_data(ra):
    M, K = 10, 128                                           # pq.rs:431-440
    x = synth.normalish(9100, (200_000, 2 * M))
    q = np.stack([x[np.arange(K) * 131 + 7 * m, 2 * m:2 * m + 2] for m in range(M)])   # data points as centroids
    mean_list = check(ra, q, x, min_fast=1.0)
    assert max(mean_list) < 16, mean_list                    # a handful of the 128 centroids per sub-vector (whole words counted)


@pytest.mark.parametrize("K", [1, 2, 3, 17, 47, 48, 128, 255, 256])
def test_every_k_uniform_data_over_both_grids(ra, K):
    M = 3
    rng = np.random.default_rng(9200 + K)
    q = rng.standard_normal((M, K, 2)).astype(np.float32)
    # uniform over 20 x the centroids' range: fine cells, coarse cells and rows outside both
    x = (rng.random((60_000, 2 * M)).astype(np.float32) - np.float32(0.5)) * np.float32(60.0)
    x[:20_000] = rng.standard_normal((20_000, 2 * M)).astype(np.float32)
    check(ra, q, x)


def test_points_on_cell_boundaries_and_on_centroids(ra):
    M, K = 2, 128
    rng = np.random.default_rng(9300)
    q = rng.standard_normal((M, K, 2)).astype(np.float32)
    words, off = ra.vor2_tables(q)
    rows = []
    for m in range(M):
        hf = words[off[m]:off[m + 1]].view(np.float32)
        G = int(words[off[m] + 13])
        for lo, inv, g in ((hf[0], hf[1], G), (hf[5], hf[6], 16)):
            edges = (np.float64(lo) + np.arange(g + 1) / np.float64(inv)).astype(np.float32)
            for e in edges:                                  # the edge, and its float neighbours
                for v in (e, np.nextafter(e, np.float32(-np.inf)), np.nextafter(e, np.float32(np.inf))):
                    rows.append(v)
    vals = np.array(rows, dtype=np.float32)
    a, b = np.meshgrid(vals, vals)
    x = np.zeros((a.size, 2 * M), dtype=np.float32)
    for m in range(M):
        x[:, 2 * m] = a.ravel()
        x[:, 2 * m + 1] = np.roll(b.ravel(), 17 * m)
    check(ra, q, x[:400_000])
    # rows that ARE centroids, midpoints of centroid pairs (exact ties up to rounding), duplicated centroids
    q2 = q.copy()
    q2[0, 100] = q2[0, 3]
    q2[1, 5] = q2[1, 77]
    mid = ((q2[:, :64] + q2[:, 64:]) * np.float32(0.5))
    x2 = np.concatenate([q2.transpose(1, 0, 2).reshape(K, 2 * M), mid.transpose(1, 0, 2).reshape(64, 2 * M)])
    check(ra, q2, x2, min_fast=1.0)


@pytest.mark.parametrize("K,scale", [(128, 1.0), (16, 1.0), (256, 1.0), (128, 1e-6), (128, 4e5)])
def test_points_on_bisectors_of_centroid_pairs(ra, K, scale):
    """Where list membership decides: points ON the perpendicular bisector of two centroids (the two distances agree to the
    last bits, either may win after rounding, and the tie rule picks the lower index) and a few ulps to either side of it, for
    pairs that are Voronoi neighbours and pairs that are not, at positions spread over both grids."""
    M = 2
    rng = np.random.default_rng(9350 + K)
    q = (rng.standard_normal((M, K, 2)) * scale).astype(np.float32)
    n = 150_000
    x = np.zeros((n, 2 * M), dtype=np.float32)
    for m in range(M):
        i = rng.integers(0, K, n)
        j = (i + 1 + rng.integers(0, max(K - 1, 1), n)) % K
        a, b = q[m, i].astype(np.float64), q[m, j].astype(np.float64)
        mid, dirv = (a + b) / 2, (b - a)[:, ::-1] * np.array([1.0, -1.0])
        t = rng.standard_normal(n)[:, None] * rng.choice([0.0, 0.1, 1.0, 5.0], n)[:, None]
        p = (mid + t * dirv).astype(np.float32)
        k = rng.integers(-3, 4, n)                                            # 0 .. 3 ulps off the bisector, either side
        step = np.where(k >= 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32)
        for _ in range(3):
            move = np.abs(k) > 0
            p[move, 0] = np.nextafter(p[move, 0], step[move])
            k = k - np.sign(k)
        x[:, 2 * m:2 * m + 2] = p
    check(ra, q, x)


@pytest.mark.parametrize("kind", ["identical", "collinear", "outlier", "tiny", "large", "two_clusters", "lattice"])
def test_adversarial_codebooks(ra, kind):
    M, K = 2, 64
    rng = np.random.default_rng(9400)
    q = rng.standard_normal((M, K, 2)).astype(np.float32)
    scale = np.float32(1.0)
    if kind == "identical":
        q[:] = q[:, :1]
    elif kind == "collinear":
        q[:, :, 1] = q[:, :, 0] * np.float32(0.5)
    elif kind == "outlier":
        q[:, 0] = np.float32(1e4)
    elif kind == "tiny":
        scale = np.float32(1e-18)
    elif kind == "large":
        scale = np.float32(3e9)
    elif kind == "two_clusters":
        q[:, :32] += np.float32(1000.0)
    elif kind == "lattice":                                  # many exact ties
        g = np.arange(8, dtype=np.float32)
        q[:] = np.stack(np.meshgrid(g, g), -1).reshape(64, 2)
    q = q * scale
    x = rng.standard_normal((40_000, 2 * M)).astype(np.float32) * np.float32(2.0) * scale
    if kind == "two_clusters":
        x[:20_000] += np.float32(1000.0)
    if kind == "lattice":
        x = (rng.integers(-2, 20, (40_000, 2 * M)).astype(np.float32) * np.float32(0.5))
    if kind == "outlier":
        x[:1000] = np.float32(1e4) + rng.standard_normal((1000, 2 * M)).astype(np.float32)
    if kind in ("identical", "tiny") and ra.vor2_tables(q) is None:
        return                                               # lists beyond 64 candidates (every cell lists every centroid; distances of 1e-36 against the 2^-120 slack): refused
    check(ra, q, x)


def test_ineligible_codebooks(ra):
    q = synth.normalish(9500, (2, 16, 2))
    for bad in (np.nan, np.inf, np.float32(3e12)):
        b = q.copy()
        b[1, 3, 0] = bad
        assert ra.vor2_tables(b) is None
    assert ra.vor2_tables(synth.normalish(9501, (1, 257, 2))) is None


@pytest.mark.parametrize("K", [1, 2, 16, 100, 256])
def test_one_float_sub_vectors(ra, K):
    """A codebook per dimension (dsub = 1): the same tables with every centroid on the x axis; the walk's 2-float formulas with a
    zero second coordinate are CANON-F32's 1-float distances bit for bit, so the result must again be the oracle's code --
    gaussian rows, rows over both grids and beyond, every cell edge with its float neighbours, midpoints of neighbouring
    centroids, duplicated centroids."""
    M = 6
    rng = np.random.default_rng(9800 + K)
    q = rng.standard_normal((M, K, 1)).astype(np.float32)
    if K > 4:
        q[1, K - 1] = q[1, 0]
    x = rng.standard_normal((60_000, M)).astype(np.float32)
    x[20_000:40_000] = (rng.random((20_000, M)).astype(np.float32) - np.float32(0.5)) * np.float32(80.0)
    words, off = ra.vor2_tables(q)
    for m in range(M):
        hf = words[off[m]:off[m + 1]].view(np.float32)
        edges = (np.float64(hf[0]) + np.arange(int(hf[4]) + 1) / np.float64(hf[1])).astype(np.float32)
        vals = np.concatenate([edges, np.nextafter(edges, np.float32(-np.inf)), np.nextafter(edges, np.float32(np.inf))])
        srt = np.sort(q[m, :, 0])
        vals = np.concatenate([vals, ((srt[:-1].astype(np.float64) + srt[1:]) / 2).astype(np.float32), q[m, :, 0]])
        x[40_000:40_000 + vals.size, m] = vals[:20_000]
    check(ra, q, x)
