"""k_reconstruct_cg (short sub-vectors: 1 or 2 floats, K <= 256, u8 codes; centroids of a group of subquantizers in LDS) against
the oracle's `reconstruct_batch` (primitives.rs:137-147, 169-172) -- a pure copy, so bit-identical -- in the plain form, the
lookup form with and without scales (one- and two-pass), with strided code and output matrices, code columns that are not word
aligned, ragged last groups, and the reference's index panic for a code >= K / a row index outside the matrix."""
import numpy as np
import pytest

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


SHAPES = [  # d, M, K, n
    (300, 150, 256, 20_011),    # finalfusion's usual quantizer: five groups of 32, 30, ..
    (300, 300, 256, 5_003),     # one float per sub-vector: five groups of 64, the last of 44
    (300, 75, 256, 9_001),      # four floats: stays on k_reconstruct
    (128, 128, 256, 7_777),
    (128, 64, 256, 4_099),
    (128, 32, 255, 4_099),
    (20, 10, 128, 30_001),      # the reference's test shape (pq.rs:431-440): one group
    (64, 32, 128, 10_000),
    (8, 2, 7, 1_000),
    (4, 4, 1, 777),
    (4, 2, 3, 1),
    (512, 256, 16, 3_000),      # many subquantizers, small codebooks: the 64-chunk cap per group
    (1024, 1024, 256, 600),
    (300, 150, 256, 1), (300, 150, 256, 2), (300, 150, 256, 511), (300, 150, 256, 513), (300, 300, 100, 65),
]


@pytest.mark.parametrize("d,M,K,n", SHAPES)
def test_reconstruct_short_sub_vectors(ra, d, M, K, n):
    import torch
    rng = np.random.default_rng(d * 7 + M + K)
    q = rng.standard_normal((M, K, d // M)).astype(np.float32)
    codes = rng.integers(0, K, (n, M)).astype(np.uint8)
    pq = ra.Pq(None, q)
    ra.launch_log(reset=True)
    got = pq.reconstruct_batch_device(torch.from_numpy(codes).cuda()).cpu().numpy()
    assert got.tobytes() == orc.reconstruct_batch(q, codes).tobytes()
    assert ("k_reconstruct_cg" in ra.launch_log()) == (d // M <= 2)


def test_strided_codes_and_output_and_unaligned_code_columns(ra):
    import torch
    rng = np.random.default_rng(5)
    for d, M, K in [(300, 150, 256), (128, 128, 64), (64, 16, 256)]:
        n = 3_001
        q = rng.standard_normal((M, K, d // M)).astype(np.float32)
        pq = ra.Pq(None, q)
        for off, extra in [(0, 3), (1, 2), (3, 4), (2, 0)]:
            wide = rng.integers(0, K, (n, off + M + extra)).astype(np.uint8)
            codes = wide[:, off:off + M]
            cd = torch.from_numpy(wide).cuda()[:, off:off + M]            # row stride != M, first column at byte `off`
            out_w = torch.full((n, d + 8), -7.0, device="cuda", dtype=torch.float32)
            out = out_w[:, 4:4 + d]                                       # 16-byte aligned rows inside a wider matrix
            pq.reconstruct_batch_device(cd, out=out)
            assert out.cpu().numpy().tobytes() == orc.reconstruct_batch(q, np.ascontiguousarray(codes)).tobytes(), (d, M, K, off)
            ow = out_w.cpu().numpy()
            assert (ow[:, :4] == -7.0).all() and (ow[:, 4 + d:] == -7.0).all()


def test_lookup_forms(ra):
    import torch
    rng = np.random.default_rng(6)
    for d, M, K in [(300, 150, 256), (300, 75, 256), (96, 96, 200), (20, 10, 128)]:
        N, n = 40_000, 12_345
        q = rng.standard_normal((M, K, d // M)).astype(np.float32)
        codes = rng.integers(0, K, (N, M)).astype(np.uint8)
        rows = rng.integers(0, N, n).astype(np.int64)
        scales = rng.random(N).astype(np.float32) + np.float32(0.5)
        pq = ra.Pq(None, q)
        cd = torch.from_numpy(codes).cuda()
        base = orc.reconstruct_batch(q, codes[rows])
        for sc in (None, scales):
            want = base if sc is None else (base * sc[rows][:, None]).astype(np.float32)
            for two_pass in (0, 1):
                ra.set_option("lookup_two_pass", two_pass)
                got = pq.reconstruct_rows_device(cd, torch.from_numpy(rows).cuda(),
                                                 scales=None if sc is None else torch.from_numpy(sc).cuda()).cpu().numpy()
                assert got.tobytes() == want.tobytes(), (d, M, K, sc is not None, two_pass)
        ra.set_option("lookup_two_pass", 2)
        bad_rows = rows.copy()
        bad_rows[n // 2] = N
        with pytest.raises(ra.PanicError):
            pq.reconstruct_rows_device(cd, torch.from_numpy(bad_rows).cuda())


def test_code_out_of_range_is_the_reference_panic(ra):
    import torch
    rng = np.random.default_rng(7)
    for d, M, K in [(300, 150, 200), (64, 64, 17), (32, 8, 100)]:
        q = rng.standard_normal((M, K, d // M)).astype(np.float32)
        codes = rng.integers(0, K, (5_000, M)).astype(np.uint8)
        codes[4_321, M - 1] = K
        pq = ra.Pq(None, q)
        with pytest.raises(ra.PanicError):
            pq.reconstruct_batch_device(torch.from_numpy(codes).cuda())
        codes[4_321, M - 1] = K - 1
        got = pq.reconstruct_batch_device(torch.from_numpy(codes).cuda()).cpu().numpy()
        assert got.tobytes() == orc.reconstruct_batch(q, codes).tobytes()


def test_the_kernel_is_the_one_that_ran(ra):
    import torch
    rng = np.random.default_rng(8)
    q = rng.standard_normal((150, 256, 2)).astype(np.float32)
    pq = ra.Pq(None, q)
    cd = torch.from_numpy(rng.integers(0, 256, (1_000, 150)).astype(np.uint8)).cuda()
    ra.launch_log(reset=True)
    pq.reconstruct_batch_device(cd)
    assert "k_reconstruct_cg" in ra.launch_log()
    pq16 = ra.Pq(None, rng.standard_normal((15, 256, 20)).astype(np.float32))
    ra.launch_log(reset=True)
    pq16.reconstruct_batch_device(cd[:, :15].contiguous())
    assert "k_reconstruct_cg" not in ra.launch_log() and "k_reconstruct" in ra.launch_log()
