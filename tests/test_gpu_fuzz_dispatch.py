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


def test_zz_the_seeds_reach_every_kernel_family(ra):
    assert {"k_encode_small16", "k_encode_vor2", "k_encode_mfma_lds3", "k_encode_mfma"} <= SEEN, SEEN
