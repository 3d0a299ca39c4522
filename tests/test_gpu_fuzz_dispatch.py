"""Randomised differential test of the encode dispatch: for seeded random shapes (M, K, dsub, n), row strides, column offsets
(rows that are 4-, 8- or 16-byte aligned), code-matrix strides and a sprinkling of special values, whatever kernel the library
picks must return the oracle's codes (`cluster_assignment` per sub-vector, kmeans.rs:141-156 over linalg.rs:167-176) -- and so
must the scalar anchor kernel.  Guards the dispatch conditions of the shape-specific kernels (alignment, whole words of codes,
padding centroids, tables that are refused)."""
import numpy as np
import pytest

from oracle import pq_oracle as orc

pytestmark = pytest.mark.gpu

DSUBS = [1, 2, 2, 2, 3, 4, 4, 5, 8, 8, 8, 12, 16, 20, 31, 32, 33]
KS = [1, 2, 5, 16, 16, 17, 31, 32, 33, 64, 100, 128, 255, 256, 257, 300]
SEEN = set()                                                  # kernels the auto dispatch picked over all seeds


@pytest.fixture(scope="module")
def ra():
    import os
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    reductive_amd.lib()
    return reductive_amd


@pytest.mark.parametrize("seed", range(48))
def test_random_shape_matches_oracle(ra, seed):
    import torch
    rng = np.random.default_rng(31_000 + seed)
    dsub = DSUBS[rng.integers(len(DSUBS))]
    K = KS[rng.integers(len(KS))]
    M = int(rng.integers(1, 41 if dsub <= 8 else 13))
    n = int(rng.integers(1, 6000))
    d = M * dsub
    q = rng.standard_normal((M, K, dsub)).astype(np.float32)
    if rng.random() < 0.3 and K > 1:
        q[:, K - 1] = q[:, 0]                                 # duplicated centroids: the lower index wins
    x = (rng.standard_normal((n, d)) * rng.choice([0.5, 1.0, 3.0])).astype(np.float32)
    if n > 8 and rng.random() < 0.7:
        x[1, rng.integers(d)] = np.nan
        x[2, rng.integers(d)] = np.inf
        x[3] *= np.float32(1e19)
        x[4] = 0.0
        x[5, :dsub] = q[0, K // 2]
        x[n - 1] = -np.inf
    pad_l, pad_r = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    wide = torch.zeros((n, pad_l + d + pad_r), device="cuda")
    wide[:, pad_l:pad_l + d] = torch.from_numpy(x).cuda()
    view = wide[:, pad_l:pad_l + d]
    opad = int(rng.integers(0, 6))
    dt = torch.uint8 if K <= 256 else torch.int32
    out = torch.full((n, M + opad), 7, device="cuda", dtype=dt)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=4, dtype=np.uint8 if K <= 256 else np.uint32)
    pq = ra.Pq(None, q)
    pq.quantize_batch_device(view, out=out[:, :M])
    got = out[:, :M].cpu().numpy()
    assert (got.astype(np.int64) == want.astype(np.int64)).all(), (pq.last_encode_kernel(), M, K, dsub, n, pad_l, pad_r, opad)
    assert int((out[:, M:] != 7).sum()) == 0
    SEEN.add(pq.last_encode_kernel().split("<")[0])
    anchor = ra.Pq(None, q)
    anchor.set_encode_variant(1)
    got1 = anchor.quantize_batch_device(view).cpu().numpy()
    assert (got1.astype(np.int64) == want.astype(np.int64)).all(), ("anchor", M, K, dsub, n)


OPQ_SHAPES = [(15, 20), (16, 16), (20, 16), (24, 16), (48, 16), (32, 16), (13, 24), (3, 5), (6, 10), (12, 4), (2, 32), (5, 7),
              (40, 8), (9, 33), (1, 12), (56, 16)]


@pytest.mark.parametrize("seed", range(24))
def test_random_opq_shape_encode_and_reconstruct_match_oracle(ra, seed):
    """The same for `Pq` with a projection (pq.rs:276-282 encode, pq.rs:296-312 reconstruct): fused rotation + encode where
    instantiated, rotation -> scratch -> encode elsewhere, gather inside the inverse rotation."""
    import torch
    import synth
    rng = np.random.default_rng(32_000 + seed)
    M, dsub = OPQ_SHAPES[rng.integers(len(OPQ_SHAPES))]
    K = [16, 100, 200, 250, 256, 256, 256][rng.integers(7)]
    n = int(rng.integers(1, 9000))
    d = M * dsub
    q = rng.standard_normal((M, K, dsub)).astype(np.float32)
    P = synth.orthonormal(32_100 + seed, d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if n > 8 and rng.random() < 0.5:
        x[1, rng.integers(d)] = np.nan
        x[2] *= np.float32(1e19)
        x[3] = 0.0
    pad = 4 * int(rng.integers(0, 3))
    wide = torch.zeros((n, d + pad), device="cuda")
    wide[:, :d] = torch.from_numpy(x).cuda()
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, projection=P, n_threads=4)
    pq = ra.Pq(P, q)
    got = pq.quantize_batch_device(wide[:, :d]).cpu().numpy()
    assert got.tobytes() == want.tobytes(), (pq.last_encode_kernel(), M, K, dsub, n, pad)
    codes = rng.integers(0, K, (n, M)).astype(np.uint8)
    want_r = orc.reconstruct_batch(q, codes, projection=P)
    got_r = pq.reconstruct_batch_device(torch.from_numpy(codes).cuda()).cpu().numpy()
    assert got_r.tobytes() == want_r.tobytes(), ("reconstruct", M, K, dsub, n)
    plain = ra.Pq(None, q)
    assert plain.reconstruct_batch_device(torch.from_numpy(codes).cuda()).cpu().numpy().tobytes() == orc.reconstruct_batch(q, codes).tobytes()


@pytest.mark.parametrize("seed", range(20))
def test_random_lookup_and_adc_match_oracle(ra, seed):
    """Row lookups with and without scales (one- and two-pass forms) and ADC tables / scans for 1 .. 9 queries, random shapes:
    `reconstruct_batch(codes.select(rows)) * scales`, `sum_m table[m][code]` in subquantizer order."""
    import torch
    import synth
    rng = np.random.default_rng(33_000 + seed)
    dsub = [2, 3, 4, 5, 8, 10, 16, 20, 24][rng.integers(9)]
    M = int(rng.integers(1, 33))
    K = [2, 16, 100, 256, 256][rng.integers(5)]
    N = int(rng.integers(1, 50_000))
    n = int(rng.integers(1, 20_000))
    d = M * dsub
    q = rng.standard_normal((M, K, dsub)).astype(np.float32)
    P = synth.orthonormal(33_100 + seed, d) if rng.random() < 0.4 else None
    codes = rng.integers(0, K, (N, M)).astype(np.uint8)
    rows = rng.integers(0, N, n).astype(np.int64)
    scales = (rng.random(N).astype(np.float32) + np.float32(0.5)) if rng.random() < 0.6 else None
    pq = ra.Pq(P, q)
    cd = torch.from_numpy(codes).cuda()
    want = orc.reconstruct_batch(q, codes[rows], projection=P)
    if scales is not None:
        want = (want * scales[rows][:, None]).astype(np.float32)
    for two_pass in (0, 1):
        ra.set_option("lookup_two_pass", two_pass)
        got = pq.reconstruct_rows_device(cd, torch.from_numpy(rows).cuda(),
                                         scales=None if scales is None else torch.from_numpy(scales).cuda()).cpu().numpy()
        assert got.tobytes() == want.tobytes(), ("lookup", two_pass, M, K, dsub, N, n, P is not None, scales is not None)
    ra.set_option("lookup_two_pass", 2)
    nq = int(rng.integers(1, 10))
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    tabs = pq.adc_tables_device(torch.from_numpy(qs).cuda())
    want_t = orc.adc_tables(q, qs, projection=P)
    assert tabs.cpu().numpy().tobytes() == want_t.tobytes(), ("adc tables", M, K, dsub, nq)
    dist = pq.adc_scan_device(cd, tabs).cpu().numpy()
    assert dist.tobytes() == orc.adc_scan(want_t, codes).tobytes(), ("adc scan", M, K, dsub, N, nq)


@pytest.mark.parametrize("seed", range(12))
def test_random_kmeans_iterations_match_oracle(ra, seed):
    """kmeans_iteration for all subquantizers (kmeans.rs:289-327): assignment, row-ordered update, exact loss -- random shapes,
    empty clusters included (more centroids than distinct rows in some seeds)."""
    import torch
    rng = np.random.default_rng(34_000 + seed)
    dsub = [1, 2, 3, 4, 8, 10, 20, 33][rng.integers(8)]
    M = int(rng.integers(1, 9))
    K = [2, 7, 16, 64, 256, 300][rng.integers(6)]
    n = int(rng.integers(K, 40_000))
    x = rng.standard_normal((n, M * dsub)).astype(np.float32)
    if rng.random() < 0.3:
        x[: n // 2] = x[n // 2: n // 2 + n // 2][: n // 2]          # duplicated rows: exact ties between assignments
    q0 = rng.standard_normal((M, K, dsub)).astype(np.float32)
    if rng.random() < 0.5:
        q0 = np.stack([x[(np.arange(K) * 31) % n, m * dsub:(m + 1) * dsub] for m in range(M)])
    iters = int(rng.integers(1, 4))
    want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=iters, n_threads=4)
    got_q, got_loss = ra.kmeans_iterations(q0, torch.from_numpy(x).cuda(), n_iterations=iters)
    assert got_q.tobytes() == want_q.tobytes(), (M, K, dsub, n, iters)
    assert got_loss.tobytes() == want_loss.tobytes(), (M, K, dsub, n, iters)


def test_zz_the_seeds_reach_every_kernel_family(ra):
    assert {"k_encode_small16", "k_encode_vor2", "k_encode_mfma_lds3"} <= SEEN, SEEN
