"""The small-codebook encode kernel (K <= 64; kernels_smallk.hip.h): x read once, centroids on the scalar
path, lane-local first-minimum scan.  Same codes as the oracle (and as the MFMA kernels) for every
instantiated (padded K, dsub) pair, ragged row counts, strided rows and special values.
Reference shape: benches/pq.rs:9-10 (d = 128, M = 16, K = 16)."""
import numpy as np
import pytest

import synth
from oracle import pq_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import os
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    reductive_amd.lib()
    return reductive_amd


@pytest.mark.parametrize("K", [1, 2, 16, 17, 32, 40, 64])
@pytest.mark.parametrize("dsub", [2, 4, 6, 8, 10, 12, 16, 20, 24, 32])
def test_every_instantiation_matches_oracle(ra, K, dsub):
    import torch
    M = {2: 37, 4: 16, 6: 9, 8: 16, 10: 7, 12: 5, 16: 8, 20: 15, 24: 3, 32: 4}[dsub]
    n = 3000 + 7 * K + dsub          # ragged: not a multiple of 64
    q = synth.normalish(7300 + K + dsub, (M, K, dsub))
    x = synth.normalish(7400 + K + dsub, (n, M * dsub))
    want = orc.quantize_batch(q, x, n_threads=8)
    pq = ra.Pq(None, q)
    pq.set_encode_variant(6)         # auto takes this kernel for K <= 16, dsub <= 8 only; 6 forces it up to K = 64
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda())
    assert pq.last_encode_kernel() == "k_encode_smallk"
    assert got.cpu().numpy().tobytes() == want.tobytes()
    auto = ra.Pq(None, q)
    assert auto.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy().tobytes() == want.tobytes()
    # auto: K <= 16 with sub-vectors of 2 floats (or 4 floats and >= 48 subquantizers) -> the pair kernel; other K <= 16, dsub <= 8 -> this one
    # K <= 32 with 4-, 8- or 16-float sub-vectors -> the 16x16x4 kernel; other K <= 16, dsub <= 8 -> the scalar-path kernel
    pair = K <= 16 and (dsub == 2 or (dsub == 4 and M >= 48))
    s16 = K <= 32 and dsub in (4, 8, 12, 16, 20, 24, 32) and not pair
    assert (auto.last_encode_kernel() == "k_encode_pair16") == pair
    assert (auto.last_encode_kernel() == "k_encode_small16") == s16
    assert (auto.last_encode_kernel() == "k_encode_smallk") == (K <= 16 and dsub <= 8 and not pair and not s16)
    pq4 = ra.Pq(None, q)
    pq4.set_encode_variant(4)        # the MFMA kernel on the same input
    got4 = pq4.quantize_batch_device(torch.from_numpy(x).cuda())
    assert pq4.last_encode_kernel().startswith("k_encode_mfma")
    assert got4.cpu().numpy().tobytes() == want.tobytes()


def test_reference_bench_shape_special_values_and_strides(ra):
    import torch
    M, K, dsub = 16, 16, 8           # benches/pq.rs:9-10
    d = M * dsub
    q = synth.normalish(7500, (M, K, dsub))
    q[3, 5] = q[3, 2]                # duplicate centroids: the lower index wins
    x = synth.normalish(7501, (100_001, d))
    x[10, 3] = np.nan
    x[11, 100] = np.inf
    x[12] = -np.inf
    x[13] *= np.float32(1e19)
    x[14] = 0.0
    x[15, 24:32] = q[3, 2]           # exact tie between centroids 2 and 5 of sub-vector 3
    x[100_000, 0] = np.nan
    pq = ra.Pq(None, q)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_small16"
    assert got.tobytes() == want.tobytes()
    assert want[15, 3] == 2
    pq6 = ra.Pq(None, q)
    pq6.set_encode_variant(6)        # the scalar-path kernel on the same special values
    assert pq6.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy().tobytes() == want.tobytes()
    assert pq6.last_encode_kernel() == "k_encode_smallk"
    pq7 = ra.Pq(None, q)
    pq7.set_encode_variant(7)        # the pair kernel on the same special values (NaN / Inf reach both halves of a pair)
    assert pq7.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy().tobytes() == want.tobytes()
    assert pq7.last_encode_kernel() == "k_encode_pair16"
    # strided rows, 4-byte aligned only (row stride d + 3), codes into a wider matrix
    wide = torch.zeros((5000, d + 3), device="cuda")
    wide[:, :d] = torch.from_numpy(x[:5000]).cuda()
    out = torch.zeros((5000, M + 5), device="cuda", dtype=torch.uint8)
    pq.quantize_batch_device(wide[:, :d], out=out[:, :M])
    assert pq.last_encode_kernel() == "k_encode_smallk"      # rows not 16-byte aligned: the scalar-path kernel
    assert out[:, :M].cpu().numpy().tobytes() == want[:5000].tobytes() and int(out[:, M:].sum()) == 0
    # host-buffer entry point and wider index types
    assert pq.quantize_batch(x[:20000], dtype=np.uint32).tolist() == want[:20000].astype(np.uint32).tolist()


def test_non_finite_codebook_falls_back_and_stays_exact(ra):
    import torch
    M, K, dsub = 4, 16, 8
    q = synth.normalish(7600, (M, K, dsub))
    q[1, 7, 2] = np.inf
    x = synth.normalish(7601, (4097, M * dsub))
    pq = ra.Pq(None, q)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=4)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert pq.last_encode_kernel() not in ("k_encode_smallk", "k_encode_pair16")
    assert got.tobytes() == want.tobytes()


def test_kmeans_assignment_uses_it_and_stays_bit_identical(ra):
    """kmeans::cluster_assignments inside kmeans_iteration (kmeans.rs:319) at K = 16."""
    import torch
    n, M, K, dsub = 20000, 16, 16, 8
    x = synth.normalish(7700, (n, M * dsub))
    q0 = np.stack([x[np.arange(K) * 97 + m, m * dsub:(m + 1) * dsub] for m in range(M)])
    got_q, got_loss = ra.kmeans_iterations(q0, torch.from_numpy(x).cuda(), n_iterations=2)
    want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=2, n_threads=8)
    assert got_q.tobytes() == want_q.tobytes() and got_loss.tobytes() == want_loss.tobytes()


@pytest.mark.parametrize("M,K,dsub,n", [(16, 16, 8, 70001), (15, 16, 8, 4097), (1, 16, 8, 333), (7, 5, 4, 10000), (33, 16, 2, 5001),
                                        (48, 16, 16, 20000), (9, 1, 16, 64), (2, 13, 2, 31), (64, 16, 4, 3000), (5, 16, 16, 1)])
def test_pair_kernel_two_subquantizers_per_tile(ra, M, K, dsub, n):
    """kernels_pair16.hip.h (variant 7 / auto for K <= 16): odd M (the last pair is half empty), K < 16 (padding centroids),
    every sub-vector length, row counts that are not multiples of 32, rows that coincide with centroids, NaN / Inf / huge
    rows in both halves of a pair -- codes equal the oracle's."""
    import torch
    q = synth.normalish(7800 + M + K + dsub, (M, K, dsub))
    x = synth.normalish(7900 + M + K + dsub, (n, M * dsub))
    if n > 40:
        x[3, 0] = np.nan                              # first sub-vector (half 0 of pair 0)
        x[4, M * dsub - 1] = np.inf                   # last sub-vector
        x[5] *= np.float32(1e19)                      # huge norms everywhere
        x[6, :dsub] = q[0, K - 1]                     # a row that IS a centroid: distance 0 up to rounding
        if M > 1:
            x[7, dsub:2 * dsub] = q[1, 0]
    pq = ra.Pq(None, q)
    pq.set_encode_variant(7)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    got = pq.quantize_batch_device(xd).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_pair16"
    assert got.tobytes() == want.tobytes()
    # strided rows (4-byte aligned only) and a wider code matrix
    wide = torch.zeros((n, M * dsub + 3), device="cuda")
    wide[:, :M * dsub] = xd
    out = torch.full((n, M + 2), 255, device="cuda", dtype=torch.uint8)
    pq.quantize_batch_device(wide[:, :M * dsub], out=out[:, :M])
    assert out[:, :M].cpu().numpy().tobytes() == want.tobytes() and int((out[:, M:] != 255).sum()) == 0


def test_pair_kernel_is_refused_outside_its_shapes(ra):
    import torch
    q = synth.normalish(7990, (4, 17, 8))
    pq = ra.Pq(None, q)
    pq.set_encode_variant(7)
    with pytest.raises(Exception):
        pq.quantize_batch_device(torch.from_numpy(synth.normalish(7991, (100, 32))).cuda())


@pytest.mark.parametrize("M,K,dsub,n", [(16, 16, 8, 70001), (2, 16, 8, 333), (6, 16, 8, 5000), (4, 5, 4, 10000), (12, 16, 4, 4097),
                                        (3, 16, 8, 3000), (48, 16, 8, 20000), (16, 32, 8, 9999), (8, 17, 4, 1234), (1, 16, 8, 1),
                                        (5, 30, 4, 777), (14, 16, 8, 6400), (16, 1, 8, 64), (32, 16, 4, 263_000), (1, 3, 4, 100),
                                        (37, 16, 8, 5000), (75, 32, 4, 2000), (48, 16, 16, 20001), (3, 16, 16, 3000), (8, 32, 16, 9999),
                                        (1, 16, 16, 1), (5, 30, 16, 777), (13, 7, 16, 33), (32, 16, 32, 30001), (3, 16, 32, 2000),
                                        (8, 32, 32, 5001), (1, 9, 32, 17), (6, 16, 32, 16), (15, 16, 20, 70001), (15, 32, 20, 5001),
                                        (7, 16, 12, 3333), (3, 9, 24, 999), (25, 16, 12, 40000), (10, 16, 24, 20001), (1, 16, 20, 1),
                                        (4, 16, 20, 33)])
def test_small16_kernel(ra, M, K, dsub, n):
    """kernels_small16.hip.h (variant 10 / auto for K <= 16 with 8-float sub-vectors): one and two centroid tiles, K < 16
    (padding centroids), rows that end inside a 32-float stage, M not a multiple of 4 (byte tail of the code word), ragged
    tiles, several tiles per wave, rows that coincide with centroids, NaN / Inf / huge rows -- codes equal the oracle's."""
    import torch
    q = synth.normalish(8800 + M + K + dsub, (M, K, dsub))
    x = synth.normalish(8900 + M + K + dsub, (n, M * dsub))
    if n > 40:
        x[3, 0] = np.nan
        x[4, M * dsub - 1] = np.inf
        x[5] *= np.float32(1e19)
        x[6, :dsub] = q[0, K - 1]                     # a row that IS a centroid: distance 0 up to rounding
        x[n - 1, (M - 1) * dsub:] = q[M - 1, 0]
        x[n - 2] = -np.inf
        x[20:30] = 0.0
    pq = ra.Pq(None, q)
    pq.set_encode_variant(10)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    got = pq.quantize_batch_device(xd).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_small16"
    assert got.tobytes() == want.tobytes()
    # 16-byte aligned strided rows and a wider code matrix: 4-byte aligned row stride (word stores), then an odd one (byte stores)
    wide = torch.zeros((n, M * dsub + 4), device="cuda")
    wide[:, :M * dsub] = xd
    for extra in (4 + (-M) % 4, 3 + (-M) % 4):
        out = torch.full((n, M + extra), 254, device="cuda", dtype=torch.uint8)
        pq.quantize_batch_device(wide[:, :M * dsub], out=out[:, :M])
        assert pq.last_encode_kernel() == "k_encode_small16"
        assert out[:, :M].cpu().numpy().tobytes() == want.tobytes() and int((out[:, M:] != 254).sum()) == 0


def test_small16_kernel_is_refused_outside_its_shapes(ra):
    import torch
    for shape, cols in (((4, 40, 8), 32), ((4, 16, 10), 40), ((4, 16, 6), 24)):   # K > 32; 10- and 6-float sub-vectors
        pq = ra.Pq(None, synth.normalish(8990, shape))
        pq.set_encode_variant(10)
        with pytest.raises(Exception):
            pq.quantize_batch_device(torch.from_numpy(synth.normalish(8991, (100, cols))).cuda())
    # rows that are only 4-byte aligned
    pq = ra.Pq(None, synth.normalish(8992, (4, 16, 8)))
    pq.set_encode_variant(10)
    wide = torch.zeros((100, 35), device="cuda")
    with pytest.raises(Exception):
        pq.quantize_batch_device(wide[:, :32])
