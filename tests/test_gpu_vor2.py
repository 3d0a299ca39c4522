"""The 2-float sub-vector encode kernel (encode variant 11 / auto for Pq handles with dsub = 2, K <= 256;
kernels_vor2.hip.h): per-cell candidate lists instead of all K distances.  Codes must equal the oracle's for every row --
rows in the fine grid, in the coarse grid, outside both, NaN / Inf / huge rows, rows that are centroids, duplicated centroids.
Reference shape: pq.rs:431-440 (d = 20, M = 10, K = 128).  The table construction itself is checked on the CPU in
tests/test_vor2_tables.py."""
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


@pytest.mark.parametrize("M,K,n", [(10, 128, 200_003), (1, 1, 100), (3, 2, 4097), (7, 17, 10_000), (37, 47, 5_001), (5, 48, 70_000),
                                    (150, 256, 3_000), (64, 255, 9_999), (10, 128, 1), (2, 200, 64),
                                    (150, 256, 1), (150, 64, 513), (100, 200, 511), (66, 256, 65)])   # eight groups or more, a few rows
def test_codes_equal_oracle(ra, M, K, n):
    import torch
    rng = np.random.default_rng(9600 + M + K)
    x = synth.normalish(9601 + M + K + n, (n, 2 * M))
    q = rng.standard_normal((M, K, 2)).astype(np.float32)
    if n > 40:
        x[3, 0] = np.nan
        x[4, 2 * M - 1] = np.inf
        x[5] *= np.float32(1e19)
        x[6] *= np.float32(20.0)                              # coarse grid
        x[7] *= np.float32(500.0)                             # outside both grids
        x[8, :2] = q[0, K - 1]                                # a row that IS a centroid
        x[n - 1] = -np.inf
        x[20:30] = 0.0
    pq = ra.Pq(None, q)
    pq.set_encode_variant(11)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    got = pq.quantize_batch_device(xd).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_vor2"
    assert got.tobytes() == want.tobytes()
    auto = ra.Pq(None, q)
    assert auto.quantize_batch_device(xd).cpu().numpy().tobytes() == want.tobytes()
    assert (auto.last_encode_kernel() == "k_encode_vor2") == (K > 16)      # up to 16 centroids: the pair kernel
    # strided rows (4-byte aligned only) and a wider code matrix
    wide = torch.zeros((n, 2 * M + 3), device="cuda")
    wide[:, 1:2 * M + 1] = xd
    out = torch.full((n, M + 2), 254, device="cuda", dtype=torch.uint8)
    pq.quantize_batch_device(wide[:, 1:2 * M + 1], out=out[:, :M])
    assert out[:, :M].cpu().numpy().tobytes() == want.tobytes() and int((out[:, M:] != 254).sum()) == 0
    assert pq.quantize_batch(x[:5000]).tobytes() == want[:5000].tobytes()          # host-buffer entry point


def test_ties_duplicates_and_uniform_data(ra):
    import torch
    M, K = 4, 64
    g = np.arange(8, dtype=np.float32)
    q = np.tile(np.stack(np.meshgrid(g, g), -1).reshape(1, 64, 2), (M, 1, 1)).astype(np.float32)   # a lattice: exact ties
    q[1, 40] = q[1, 3]                                        # duplicate: the lower index wins
    rng = np.random.default_rng(9700)
    x = (rng.integers(-4, 24, (100_000, 2 * M)).astype(np.float32) * np.float32(0.5))
    x[50_000:] = (rng.random((50_000, 2 * M)).astype(np.float32) - np.float32(0.3)) * np.float32(40.0)
    pq = ra.Pq(None, q)
    want = orc.quantize_batch(q, x, n_threads=8)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_vor2"
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("M,K,n", [(128, 256, 20_001), (300, 256, 3_000), (16, 16, 70_000), (7, 1, 100), (64, 100, 9_999), (1, 256, 1),
                                    (128, 256, 1), (128, 256, 515), (300, 100, 63)])
def test_one_float_sub_vectors(ra, M, K, n):
    """A codebook per dimension (dsub = 1, scalar quantization with learned levels): the candidate-list kernel with the second
    coordinate 0; codes equal the oracle's, special values and out-of-grid rows included."""
    import torch
    rng = np.random.default_rng(9650 + M + K)
    q = rng.standard_normal((M, K, 1)).astype(np.float32)
    x = synth.normalish(9651 + M + K + n, (n, M))
    if n > 40:
        x[3, 0] = np.nan
        x[4, M - 1] = np.inf
        x[5] *= np.float32(1e19)
        x[6] *= np.float32(20.0)
        x[7] *= np.float32(500.0)
        x[8, 0] = q[0, K - 1, 0]
        x[n - 1] = -np.inf
        x[20:30] = 0.0
    pq = ra.Pq(None, q)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    got = pq.quantize_batch_device(xd).cpu().numpy()
    assert pq.last_encode_kernel() == "k_encode_vor2"
    assert got.tobytes() == want.tobytes()
    wide = torch.zeros((n, M + 3), device="cuda")
    wide[:, 1:M + 1] = xd
    out = torch.full((n, M + 2), 254, device="cuda", dtype=torch.uint8)
    pq.quantize_batch_device(wide[:, 1:M + 1], out=out[:, :M])
    assert out[:, :M].cpu().numpy().tobytes() == want.tobytes() and int((out[:, M:] != 254).sum()) == 0
    other = ra.Pq(None, q)
    other.set_encode_variant(4)
    assert other.quantize_batch_device(xd).cpu().numpy().tobytes() == want.tobytes()


def test_opq_rotation_then_candidate_lists(ra):
    """`Pq` with a projection and 2-float sub-vectors: the rotated rows (scratch chunks of the two-kernel OPQ path) go through
    the candidate-list kernel; codes equal the oracle's (pq.rs:276-282)."""
    import torch
    M, K, n = 12, 64, 50_001
    d = 2 * M
    rng = np.random.default_rng(9750)
    q = rng.standard_normal((M, K, 2)).astype(np.float32)
    P = synth.orthonormal(9751, d)
    x = synth.normalish(9752, (n, d))
    x[7] *= np.float32(300.0)
    x[8, 3] = np.nan
    pq = ra.Pq(P, q)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, projection=P, n_threads=8)
    ra.launch_log(reset=True)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    torch.cuda.synchronize()
    assert "k_encode_vor2" in ra.launch_log(reset=True)
    assert got.tobytes() == want.tobytes()


def test_five_million_rows_against_the_mfma_kernel(ra):
    """Size-independent check: the candidate-list kernel and the kernel that evaluates every centroid (variant 4) are two
    independent implementations of cluster_assignment; all n x M codes must be equal."""
    import torch
    M, K, n = 10, 128, 5_000_000
    g = torch.Generator(device="cuda").manual_seed(9800)
    x = torch.empty((n, 2 * M), device="cuda").normal_(generator=g)
    q = x[:K * M].reshape(M, K, 2 * M)[:, :, :2].cpu().numpy().copy()
    a = ra.Pq(None, q)
    b = ra.Pq(None, q)
    b.set_encode_variant(4)
    ca = a.quantize_batch_device(x)
    cb = b.quantize_batch_device(x)
    assert a.last_encode_kernel() == "k_encode_vor2" and b.last_encode_kernel().startswith("k_encode_mfma")
    assert bool((ca == cb).all())


def test_ineligible_codebooks_and_kmeans_handles_use_the_other_kernels(ra):
    import torch
    q = synth.normalish(9900, (4, 64, 2))
    q[2, 5, 1] = np.float32(3e12)                             # beyond the range the tables are built for
    x = synth.normalish(9901, (5000, 8))
    pq = ra.Pq(None, q)
    want = orc.quantize_batch(q, x, n_threads=4)
    assert pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy().tobytes() == want.tobytes()
    assert pq.last_encode_kernel() != "k_encode_vor2"
    pq.set_encode_variant(11)
    with pytest.raises(Exception):
        pq.quantize_batch_device(torch.from_numpy(x).cuda())


def test_handles_created_without_candidate_tables(ra):
    """Context option "candidate_tables" = 0 (include/pqhip.h): no host table build at creation, the same codes from the kernels that
    evaluate every centroid; handles created after the option is set back get their tables again."""
    import torch
    rng = np.random.default_rng(77)
    q = rng.standard_normal((10, 128, 2)).astype(np.float32)
    x = synth.normalish(78, (20_000, 20))
    want = orc.quantize_batch(q, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    ra.set_option("candidate_tables", 0)
    try:
        pq = ra.Pq(None, q)
        assert pq.quantize_batch_device(xd).cpu().numpy().tobytes() == want.tobytes()
        assert pq.last_encode_kernel() != "k_encode_vor2"
        pq.set_encode_variant(11)                                  # asked for by name, no tables: refused like every forced variant that does not apply
        with pytest.raises(Exception):
            pq.quantize_batch_device(xd)
    finally:
        ra.set_option("candidate_tables", 1)
    pq = ra.Pq(None, q)
    assert pq.quantize_batch_device(xd).cpu().numpy().tobytes() == want.tobytes()
    assert pq.last_encode_kernel() == "k_encode_vor2"
