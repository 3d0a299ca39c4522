"""OPQ rotation + encode fused in one kernel (encode variant 8, kernels_opq_fused2.hip.h; `rx = x.dot(P)` of
pq.rs:276 never written to memory) and the two-kernel path that serves every other shape.  Codes must equal the
oracle's bit for bit, for every input: ragged row counts, ragged last column block, K < 256, special values (the
exact path re-rotates flagged rows with the scalar rule-2 chain)."""
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


SHAPES = [  # n, M, K, dsub -- OPQ shapes WITHOUT a fused instantiation: the auto variant takes rotation -> scratch -> encode
    (50_001, 16, 128, 16),    # d = 256 has a fused instantiation for T = 8 only: K = 128 (T = 4) does not
    (30_000, 24, 200, 10),    # 6 sub-vectors per 64-column block, K not a multiple of 32
    (9_999, 2, 128, 32),      # T = 4
    (20_000, 13, 256, 24),    # d = 312
    (12_345, 12, 100, 4),     # short sub-vectors, T = 4
    (7_777, 26, 256, 12),     # d = 312, dsub 12
    (257, 8, 129, 8),
]


@pytest.mark.parametrize("n,M,K,dsub", SHAPES)
def test_shapes_without_a_fused_kernel_take_the_two_kernel_path(ra, n, M, K, dsub):
    """(These shapes ran the first-generation fused kernel, encode variant 5, until round 4 removed it.)  The launch log
    names what ran: one rotation and one encode kernel, no fused kernel; codes equal the oracle's."""
    import torch
    d = M * dsub
    q = synth.normalish(6100 + d + K, (M, K, dsub))
    P = synth.orthonormal(6101 + d, d)
    x = synth.normalish(6102 + n, (n, d))
    pq = ra.Pq(P, q)
    want = orc.quantize_batch(q, x, projection=P, n_threads=8)
    ra.launch_log(reset=True)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    log = ra.launch_log(reset=True)
    assert "k_rotate_pblock" in log and "k_encode" in log and "fused" not in log, log
    assert got.cpu().numpy().tobytes() == want.tobytes()
    assert pq.quantize_batch(x).tobytes() == want.tobytes()          # host-buffer entry point
    # strided rows (row stride > d, 16-byte aligned)
    wide = torch.zeros((n, d + 12), device="cuda")
    wide[:, :d] = torch.from_numpy(x).cuda()
    assert pq.quantize_batch_device(wide[:, :d]).cpu().numpy().tobytes() == want.tobytes()


def test_retired_variants_are_refused(ra):
    pq = ra.Pq(None, synth.normalish(6310, (15, 256, 20)))
    for v in (3, 5, 12, -1):
        with pytest.raises(ra.PqHipError, match="invalid"):
            pq.set_encode_variant(v)
    import torch
    pq.set_encode_variant(8)     # the fused OPQ kernel on a codebook without a projection
    with pytest.raises(ra.PqHipError, match="unsupported"):
        pq.quantize_batch_device(torch.from_numpy(synth.normalish(6311, (100, 300))).cuda())


# ---- the fused kernel (encode variant 8, kernels_opq_fused2.hip.h): P block AND codebook fragments in LDS, x straight from
# global memory.  Instantiated for (dsub 20, d = 300-like: split, odd, tail) and (dsub 16: d = 256 and d = 320). --------------
def _fused2(ra, q, P):
    pq = ra.Pq(P, q)
    pq.set_encode_variant(8)
    return pq


@pytest.mark.parametrize("n,M,K,dsub", [(200_003, 15, 256, 20), (50_001, 16, 256, 16), (40_000, 20, 256, 16), (31, 15, 256, 20),
                                        (1, 15, 256, 20), (3073, 15, 256, 20), (9000, 15, 250, 20), (6145, 16, 225, 16),
                                        # 32-slot P blocks (round 4): dimensions whose 64-column block does not fit LDS
                                        (60_001, 48, 256, 16), (9_000, 32, 250, 16), (5_001, 24, 256, 16), (3073, 56, 256, 16),
                                        (33, 48, 256, 16)])
def test_fused2_codes_equal_oracle(ra, n, M, K, dsub):
    import torch
    d = M * dsub
    q = synth.normalish(6500 + d + K, (M, K, dsub))
    P = synth.orthonormal(6501 + d, d)
    x = synth.normalish(6502 + n, (n, d))
    pq = _fused2(ra, q, P)
    want = orc.quantize_batch(q, x, projection=P, n_threads=8)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert pq.last_encode_kernel() == "k_opq_encode_fused2"
    assert got.cpu().numpy().tobytes() == want.tobytes()
    assert pq.quantize_batch(x).tobytes() == want.tobytes()          # host-buffer entry point, same kernel
    wide = torch.zeros((n, d + 12), device="cuda")
    wide[:, :d] = torch.from_numpy(x).cuda()
    assert pq.quantize_batch_device(wide[:, :d]).cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("M,dsub", [(15, 20), (48, 16)])
def test_fused2_special_values_take_the_exact_path(ra, M, dsub):
    import torch
    K = 256
    d = M * dsub
    q = synth.normalish(6600, (M, K, dsub))
    P = synth.orthonormal(6601, d)
    x = synth.normalish(6602, (4096, d))
    pick = synth.codes_u8(6603, (512, M), K).astype(np.int64)
    cent = np.concatenate([q[m, pick[:, m]] for m in range(M)], axis=1)
    x[100:612] = (cent.astype(np.float64) @ P.T.astype(np.float64)).astype(np.float32)
    x[700, 3] = np.nan
    x[701, 250] = np.inf
    x[702] = -np.inf
    x[703] *= np.float32(1e19)
    x[704] = 0.0
    x[705] = np.float32(1e-30)
    x[4095, 0] = np.nan
    x[4094, d - 1] = np.nan              # last column of the last column block
    pq = _fused2(ra, q, P)
    with np.errstate(all="ignore"):
        want = orc.quantize_batch(q, x, projection=P, n_threads=8)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert pq.last_encode_kernel() == "k_opq_encode_fused2"
    assert got.tobytes() == want.tobytes()


def test_fused2_refuses_shapes_without_an_instantiation(ra):
    import torch
    for M, K, dsub in ((24, 200, 10), (2, 128, 32), (26, 256, 12), (12, 256, 20)):   # other dsub; T = 4; d = 312; d = 240 (no split)
        d = M * dsub
        pq = _fused2(ra, synth.normalish(6700 + d, (M, K, dsub)), synth.orthonormal(6701 + d, d))
        x = torch.from_numpy(synth.normalish(6702, (100, d))).cuda()
        with pytest.raises(ra.PqHipError, match="unsupported"):
            pq.quantize_batch_device(x)


def test_fused2_one_million_rows_every_code(ra):
    import torch
    M, K, dsub = 15, 256, 20
    d = M * dsub
    q = synth.normalish(6800, (M, K, dsub))
    P = synth.orthonormal(6801, d)
    g = torch.Generator(device="cuda").manual_seed(6802)
    x = torch.empty((1_000_000, d), device="cuda").normal_(generator=g)
    pq = _fused2(ra, q, P)
    got = pq.quantize_batch_device(x).cpu().numpy()
    want = orc.quantize_batch(q, x.cpu().numpy(), projection=P, n_threads=16)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("M,dsub,n", [(48, 16, 3_000_001), (15, 20, 3_000_001)])
def test_fused_and_two_kernel_paths_agree_on_every_code_of_a_large_batch(ra, ctx_options, M, dsub, n):
    """Size-independent check of the fused kernels at batch sizes the oracle cannot walk: the fused launch (64-slot P blocks at
    d = 300, 32-slot at d = 768 -- the size of BASELINE configs[4]) and the chunked rotation -> scratch -> encode path are two
    independent implementations of pq.rs:276-282; all n x M codes must be equal, and both equal the oracle on head and tail."""
    import torch
    K = 256
    d = M * dsub
    q = synth.normalish(6900 + d, (M, K, dsub))
    P = synth.orthonormal(6901 + d, d)
    g = torch.Generator(device="cuda").manual_seed(6902)
    x = torch.empty((n, d), device="cuda").normal_(generator=g)
    pq = ra.Pq(P, q)
    ra.launch_log(reset=True)
    fused = pq.quantize_batch_device(x)
    log = ra.launch_log(reset=True)            # (the codebook's preparation kernels run with the first call)
    assert log.endswith("k_opq_encode_fused2") and "k_rotate" not in log, log
    ctx_options("opq_fused", 0)
    two = pq.quantize_batch_device(x)
    log = ra.launch_log(reset=True)
    assert "k_rotate_pblock" in log and "k_encode_mfma16" in log and "fused" not in log, log
    assert torch.equal(fused, two)
    for r0 in (0, n - 4000):
        want = orc.quantize_batch(q, x[r0:r0 + 4000].cpu().numpy(), projection=P, n_threads=16)
        assert fused[r0:r0 + 4000].cpu().numpy().tobytes() == want.tobytes(), r0
