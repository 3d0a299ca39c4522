"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
include/pqhip.h (via the `Pq` host mirror), against the CPU oracle and the golden fixtures.
Bar: codes bit-exact; PQ reconstructions bit-exact; OPQ reconstructions within 1e-5 relative."""
import ctypes
import json
import os

import numpy as np
import pytest

import synth
from oracle import pq_oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-5   # north_star: reconstructions agree within 1e-5 relative


@pytest.fixture(scope="module")
def ra():
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()    # same image on the GPU box: hipcc is there
    reductive_amd.lib()          # must load: no fallback
    return reductive_amd


def _pq(ra, q, P=None, variant=0):
    pq = ra.Pq(P, q)
    if variant:
        pq.set_encode_variant(variant)
    return pq


# ---- the hardware property the MFMA path rests on --------------------------------------------
@pytest.mark.parametrize("k", [2, 7, 16, 20, 300])
def test_mfma_is_a_k_ordered_fmaf_chain(ra, k):
    from reductive_amd.pq import default_ctx
    bad = ctypes.c_int64(-1)
    rc = ra.lib().pqhip_selftest_mfma_chain(default_ctx().handle, 0, k, 256, 1234 + k,
                                            ctypes.byref(bad))
    assert rc == 0, ra.lib().pqhip_last_hip_error()
    assert bad.value == 0


# ---- the reference's own KATs through the GPU path -------------------------------------------
def test_kat_quantize_and_reconstruct(ra, kats):
    k = kats["pq_predefined_codebook"]
    pq = _pq(ra, np.array(k["quantizers"], np.float32))
    x = np.array(k["vectors"], np.float32)
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        assert pq.quantize_batch(x, dtype=dt).tolist() == k["quantizations"]
    rec = pq.reconstruct_batch(np.array(k["quantizations"], np.uint64))
    assert rec.tolist() == k["reconstructions"]
    assert pq.reconstruct_batch(np.array(k["quantizations"], np.uint8)).tolist() == k["reconstructions"]
    assert pq.quantized_len() == k["quantized_len"]
    assert pq.reconstructed_len() == k["reconstructed_len"]


def test_kat_cluster_assignments(ra, kats):
    k = kats["cluster_assignments"]
    c = np.array(k["centroids"], np.float32)
    x = np.array(k["instances"], np.float32)
    pq = _pq(ra, c[None])
    assert pq.quantize_batch(x)[:, 0].tolist() == k["assignments"]
    assert pq.quantize_batch(np.asfortranarray(x))[:, 0].tolist() == k["assignments"]  # kmeans.rs:397-399


# ---- golden fixtures --------------------------------------------------------------------------
def _golden_cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return sorted(json.load(f).keys())


@pytest.mark.parametrize("name", _golden_cases())
def test_golden_fixture(ra, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    meta = json.load(open(os.path.join(GOLD, "cases.json")))[name]
    data = np.load(os.path.join(GOLD, "cases.npz"))
    q, x, opq, _ = mg.make_inputs(name)
    assert synth.sha(q) == meta["sha_q"] and synth.sha(x) == meta["sha_x"]
    P = data[name + "/projection"] if opq else None
    want = data[name + "/codes"]
    pq = _pq(ra, q, P)
    got = pq.quantize_batch(x)
    assert got.tobytes() == want.tobytes()
    rec = pq.reconstruct_batch(want)
    if opq:
        ref = orc.reconstruct_batch(q, want, projection=P)
        assert np.abs(rec - ref).max() <= REL_TOL * np.abs(ref).max()
    else:
        assert synth.sha(rec) == meta["sha_rec"]


# ---- seeded parity against the oracle, both kernels -------------------------------------------
SHAPES = [  # n, M, K, dsub
    (4113, 15, 256, 20),    # headline shape, ragged n
    (1000, 48, 256, 16),    # d=768 shape
    (999, 10, 128, 2),      # pq.rs:435-436 shape
    (500, 16, 16, 8),       # benches/pq.rs shape
    (300, 5, 40, 7),        # odd dsub, K not a multiple of 32
    (257, 3, 200, 32),      # largest resident fragment set
    (100, 4, 300, 6),       # K > 256 -> wider index type, anchor kernel
    (64, 2, 64, 40),        # wide sub-vectors
    (50, 2, 40, 260),       # dsub > 256 -> anchor kernel
    (33, 1, 1, 5),          # K = 1
    (1, 15, 256, 20),       # single row
]


def _has_mfma16(K, dsub):
    """shapes the 16x16x4 kernel (variant 9, the auto choice there) is instantiated for"""
    return 32 < K <= 256 and dsub % 4 == 0 and dsub <= 32


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("variant", [0, 1, 2, 4, 9])
def test_encode_matches_oracle(ra, shape, variant):
    n, M, K, dsub = shape
    if variant == 9 and not _has_mfma16(K, dsub):
        pytest.skip("no 16x16x4 instantiation for this shape")
    q = synth.normalish(100 + n, (M, K, dsub))
    x = synth.normalish(200 + n, (n, M * dsub))
    dt = np.uint8 if K <= 256 else np.uint16
    if variant >= 2 and (K > 256 or dsub > 32):
        pytest.skip("shape not covered by the MFMA kernels (anchor kernel is used)")
    want = orc.quantize_batch(q, x, dtype=dt)
    pq = _pq(ra, q, variant=variant)
    got = pq.quantize_batch(x, dtype=dt)
    assert got.tobytes() == want.tobytes()
    if variant == 0 and K <= 256 and dsub <= 32:
        # auto: one of the small-codebook kernels where instantiated, else an MFMA kernel
        assert pq.last_encode_kernel().startswith(("k_encode_mfma", "k_encode_smallk", "k_encode_small16", "k_encode_pair16", "k_encode_vor2"))
        assert (pq.last_encode_kernel() == "k_encode_mfma16") == (_has_mfma16(K, dsub) and K > 128 and 12 <= dsub <= 24)
    if variant == 9:
        assert pq.last_encode_kernel() == "k_encode_mfma16"
    if variant == 4:
        assert pq.last_encode_kernel().startswith("k_encode_mfma_lds3")


def test_encode_special_values(ra):
    M, K, dsub = 4, 64, 8
    q = synth.normalish(31, (M, K, dsub))
    x = synth.normalish(32, (200, M * dsub))
    x[3, 1] = np.nan
    x[50, 9] = np.inf
    x[51, 17] = -np.inf
    x[100] = 0
    x[101] = -0.0
    x[150] *= np.float32(1e19)          # xx ~ 1e38 .. overflow region
    x[151] *= np.float32(3e19)          # xx overflows to inf
    x[160] *= np.float32(1e-30)         # subnormal products
    want = orc.quantize_batch(q, x)
    for variant in (0, 1, 2, 4, 9):
        got = _pq(ra, q, variant=variant).quantize_batch(x)
        assert got.tobytes() == want.tobytes(), variant
    # NaN / Inf / huge centroids: codebook leaves the fast path entirely
    q2 = q.copy()
    q2[0, 5, 2] = np.nan
    q2[1, 7, 0] = np.inf
    q2[2, 9] *= np.float32(1e19)
    want2 = orc.quantize_batch(q2, x)
    assert _pq(ra, q2).quantize_batch(x).tobytes() == want2.tobytes()


def test_encode_exact_and_near_ties(ra):
    M, K, dsub = 2, 64, 4
    q = synth.normalish(41, (M, K, dsub))
    q[0, 40] = q[0, 7]                   # exact duplicates: lowest index must win
    q[1, 63] = q[1, 0]
    x = synth.normalish(42, (512, M * dsub))
    x[:64, :dsub] = q[0, 7]              # rows sitting exactly on the duplicated centroid
    x[64:128, dsub:] = q[1, 63]
    mid = (q[0, 3] + q[0, 4]) / 2        # midpoints, nudged by single ulps
    for i in range(128, 192):
        x[i, :dsub] = np.nextafter(mid, np.float32(np.inf if i % 2 else -np.inf))
    big = synth.normalish(43, (64, dsub)) * np.float32(4096)   # |x|^2 >> |c|^2: rounding ties
    x[192:256, :dsub] = big
    want = orc.quantize_batch(q, x)
    assert (want[:64, 0] == 7).all() and (want[64:128, 1] == 0).all()
    for variant in (0, 1, 2, 4, 9):
        assert _pq(ra, q, variant=variant).quantize_batch(x).tobytes() == want.tobytes(), variant


@pytest.mark.parametrize("variant", [9, 4])
def test_exact_path_rows_in_many_tiles_of_a_wave(ra, variant):
    """Headline shape, 600,000 rows (18 row tiles per wave): rows for the exact path -- a NaN, +-Inf, a huge norm, a row
    that sits exactly on a centroid (negative minimum after rounding), a -0 row -- sprinkled over every tile position,
    whole tiles of them, and the ragged last tile.  k_encode_mfma16 only records such rows in its loop (a mask per tile,
    a tile bit in a scalar register) and re-evaluates them after it; every code must equal the oracle's."""
    import torch
    M, K, dsub = 15, 256, 20
    n = 600_000 + 13
    q = synth.normalish(9301, (M, K, dsub))
    x = synth.normalish(9302, (n, M * dsub))
    rng = np.random.RandomState(9303)
    for r in range(0, n, 97):
        x[r, rng.randint(M * dsub)] = np.nan
    for r in range(5, n, 101):
        x[r, rng.randint(M * dsub)] = np.inf if r % 2 else -np.inf
    for r in range(11, n, 89):
        m = rng.randint(M)
        x[r, m * dsub:(m + 1) * dsub] = q[m, rng.randint(K)]
    for r in range(17, n, 103):
        x[r] *= np.float32(3e19)
    for r in range(23, n, 107):
        x[r] = -0.0
    x[32 * 1000:32 * 1003] = np.nan                       # three whole tiles
    x[n - 5:] *= np.float32(1e19)                          # the ragged last tile
    want = orc.quantize_batch(q, x, n_threads=16)
    pq = _pq(ra, q, variant=variant)
    got = pq.quantize_batch_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert pq.last_encode_kernel() == ("k_encode_mfma16" if variant == 9 else "k_encode_mfma_lds3<vec4>")
    assert got.tobytes() == want.tobytes()


def test_reconstruct_exact_and_range_check(ra):
    M, K, dsub = 15, 256, 20
    q = synth.normalish(51, (M, K, dsub))
    codes = synth.codes_u8(52, (3001, M), K)
    pq = _pq(ra, q)
    rec = pq.reconstruct_batch(codes)
    assert rec.tobytes() == orc.reconstruct_batch(q, codes).tobytes()
    assert pq.reconstruct_batch(codes.astype(np.uint32)).tobytes() == rec.tobytes()
    # odd sub-dimension: scalar store path
    q2 = synth.normalish(53, (5, 40, 7))
    c2 = synth.codes_u8(54, (100, 5), 40)
    assert _pq(ra, q2).reconstruct_batch(c2).tobytes() == orc.reconstruct_batch(q2, c2).tobytes()
    bad = c2.copy()
    bad[77, 3] = 40
    with pytest.raises(ra.PanicError):
        _pq(ra, q2).reconstruct_batch(bad)


def test_opq_rotate_encode_and_reconstruct(ra):
    for (n, M, K, dsub) in [(700, 15, 256, 20), (300, 8, 64, 8), (65, 3, 16, 5)]:
        d = M * dsub
        q = synth.normalish(61 + d, (M, K, dsub))
        x = synth.normalish(62 + d, (n, d))
        P = synth.orthonormal(63 + d, d)
        want = orc.quantize_batch(q, x, projection=P)
        pq = _pq(ra, q, P)
        got = pq.quantize_batch(x)
        assert got.tobytes() == want.tobytes()
        ref = orc.reconstruct_batch(q, want, projection=P)
        rec = pq.reconstruct_batch(want)
        assert np.abs(rec - ref).max() <= REL_TOL * np.abs(ref).max()
        assert rec.tobytes() == ref.tobytes()      # stronger than required: same chain rule


@pytest.mark.parametrize("d,M", [(256, 16), (260, 13), (320, 16), (324, 27), (330, 33), (352, 22), (356, 89),
                                 (384, 24), (512, 32), (515, 5), (768, 48), (1024, 64), (1300, 65)])
def test_opq_rotation_dispatch_boundaries(ra, d, M):
    """Every rotation-kernel choice of rotate_dev (pqhip_rotate.hip: k_rotate_pblock8 with a 64-column P block up to
    d = 636, k_rotate_pblock9 where 64-column blocks pad much and, with 32-column blocks, up to d = 1,280, the slab GEMM
    beyond and for d % 4 != 0), on both sides of the rule-2 restart at k = 256, for encode (x.P) and reconstruct (x.P^T)."""
    n, K = 203, 16
    dsub = d // M
    q = synth.normalish(91 + d, (M, K, dsub))
    x = synth.normalish(92 + d, (n, d))
    P = synth.orthonormal(93 + d, d)
    want = orc.quantize_batch(q, x, projection=P)
    pq = _pq(ra, q, P)
    assert pq.quantize_batch(x).tobytes() == want.tobytes()
    ref = orc.reconstruct_batch(q, want, projection=P)
    assert pq.reconstruct_batch(want).tobytes() == ref.tobytes()
    # rotation alone, checked element by element: K = 1 codebook at the origin would hide it, so
    # compare the reconstruction of a second, permuted code matrix as well
    codes2 = want[::-1].copy()
    assert pq.reconstruct_batch(codes2).tobytes() == orc.reconstruct_batch(q, codes2, projection=P).tobytes()


@pytest.mark.parametrize("M,K,dsub", [(5, 16, 103), (150, 16, 20), (64, 32, 64), (8, 16, 250), (750, 4, 4),
                                      (1, 256, 1), (3, 7, 1), (4099, 2, 2), (2, 16, 4100)])
def test_wide_and_narrow_rows(ra, M, K, dsub):
    """Row widths far from the headline shape: d up to 8200, one-column sub-vectors, M beyond the
    gather kernel's per-block code budget -- encode and reconstruct stay exact."""
    n = 131
    d = M * dsub
    q = synth.normalish(95 + d, (M, K, dsub))
    x = synth.normalish(96 + d, (n, d))
    want = orc.quantize_batch(q, x)
    pq = _pq(ra, q)
    assert pq.quantize_batch(x).tobytes() == want.tobytes()
    codes = synth.codes_u8(97 + d, (n, M), K)
    assert pq.reconstruct_batch(codes).tobytes() == orc.reconstruct_batch(q, codes).tobytes()


@pytest.mark.parametrize("shape", [(3000, 3, 1000, 16), (1500, 2, 4096, 8), (700, 1, 257, 20), (900, 4, 300, 6),
                                   (400, 2, 513, 3), (2500, 15, 1024, 20)])
def test_more_than_256_centroids_on_the_matrix_path(ra, shape):
    """K > 256 (u16/u32 codes; k-means with many centroids): groups of 256 centroids through the
    default kernel, merged by 64-bit {distance, index} keys -- codes equal the oracle's, including
    ties across groups, NaN/Inf/huge rows and exact hits (negative-distance slow path)."""
    n, M, K, dsub = shape
    q = synth.normalish(1300 + K, (M, K, dsub))
    x = synth.normalish(1301 + K, (n, M * dsub))
    q[0, K - 1] = q[0, 5]                      # duplicate in the last group: index 5 must win
    q[M - 1, 300 % K] = q[M - 1, 2]
    x[:40, :dsub] = q[0, 5]                    # rows sitting exactly on the duplicated centroid
    x[40:80, (M - 1) * dsub:] = q[M - 1, 2]
    x[100, 0] = np.nan
    x[101, 1 % (M * dsub)] = np.inf
    x[102] *= np.float32(1e19)
    x[103] *= np.float32(3e19)
    want = orc.quantize_batch(q, x, dtype=np.uint32)
    assert (want[:40, 0] == 5).all()
    pq = _pq(ra, q)
    for dt in (np.uint16, np.uint32, np.uint64):
        got = pq.quantize_batch(x, dtype=dt)
        assert got.astype(np.uint32).tobytes() == want.tobytes(), dt
    assert pq.last_encode_kernel() == "k_encode_mfma_lds3<grouped>"
    assert _pq(ra, q, variant=1).quantize_batch(x, dtype=np.uint32).tobytes() == want.tobytes()
    # OPQ in front of it, and the k-means step on top of it
    if M * dsub <= 64:
        P = synth.orthonormal(1302 + K, M * dsub)
        assert _pq(ra, q, P).quantize_batch(x, dtype=np.uint32).tobytes() == \
            orc.quantize_batch(q, x, projection=P, dtype=np.uint32).tobytes()
    xs = synth.normalish(1303 + K, (n, M * dsub))
    wq, wl = orc.kmeans_iterations(q, xs, n_iterations=2, n_threads=8)
    gq, gl = ra.kmeans_iterations(q, xs, n_iterations=2)
    assert gq.tobytes() == wq.tobytes() and gl.tobytes() == wl.tobytes()


@pytest.mark.parametrize("dsub", [33, 36, 40, 41, 47, 48, 50, 56, 57, 60, 63, 64, 65, 72, 80, 81, 96, 100, 112, 127, 128,
                                  129, 144, 150, 176, 192, 200, 255, 256,
                                  257, 300, 320, 321, 400, 512, 513, 600, 767, 768, 1000, 1024, 1025])
def test_wide_subvectors_on_the_matrix_path(ra, dsub):
    """32 < dsub <= 128: the default kernel with one wave per SIMD and 20..64-MFMA chains (DP = 40,
    48, 56, 64, 80, 96, 112, 128 with zero k-padding); 128 < dsub <= 256: k_encode_mfma_wide (groups of <= 128 centroids,
    norms by a pre-pass, keys merged); 256 < dsub <= 1,024: k_encode_mfma_wide2 (several rule-2 blocks per dot product:
    every 256-k block its own chain, block results added in order; groups of 64 / 32 centroids); beyond: the scalar anchor
    -- codes equal the oracle's, K <= 256 and grouped K > 256, special values included; the k-means step on top of it."""
    for (n, M, K) in [(777, 3, 256), (300, 2, 37), (500, 2, 300)]:
        q = synth.normalish(1400 + dsub + K, (M, K, dsub))
        x = synth.normalish(1401 + dsub + K, (n, M * dsub))
        x[5, 3] = np.nan
        x[6] *= np.float32(2e19)
        x[7, :dsub] = q[0, K - 1]
        dt = np.uint8 if K <= 256 else np.uint16
        want = orc.quantize_batch(q, x, dtype=dt)
        pq = _pq(ra, q)
        assert pq.quantize_batch(x, dtype=dt).tobytes() == want.tobytes(), (n, M, K)
        assert pq.last_encode_kernel().startswith(("k_encode_mfma_lds3", "k_encode_mfma16") if dsub <= 128 else
                                                  "k_encode_mfma_wide" if dsub <= 1024 else "k_encode_scalar")
        rec = pq.reconstruct_batch(want)
        assert rec.tobytes() == orc.reconstruct_batch(q, want).tobytes()
    q0, xs = _km_inputs(1200, 2, 16, dsub, 1500 + dsub)
    wq, wl = orc.kmeans_iterations(q0, xs, n_iterations=2)
    gq, gl = ra.kmeans_iterations(q0, xs, n_iterations=2)
    assert gq.tobytes() == wq.tobytes() and gl.tobytes() == wl.tobytes()


def test_strided_host_buffers(ra):
    M, K, dsub = 3, 32, 4
    q = synth.normalish(71, (M, K, dsub))
    big = synth.normalish(72, (40, 40))
    x = big[::2, 5:17]
    want = orc.quantize_batch(q, np.ascontiguousarray(x))
    pq = _pq(ra, q)
    assert pq.quantize_batch(x).tolist() == want.tolist()
    assert pq.quantize_batch(np.asfortranarray(x)).tolist() == want.tolist()
    out = np.zeros((M, x.shape[0]), np.uint16).T
    pq.quantize_batch_into(x, out)
    assert out.tolist() == want.tolist()
    rec = np.zeros((x.shape[0], 2 * M * dsub), np.float32)[:, ::2]
    pq.reconstruct_batch_into(want, rec)
    assert rec.tolist() == orc.reconstruct_batch(q, want).tolist()


def test_empty_batch(ra):
    q = synth.normalish(81, (2, 4, 3))
    pq = _pq(ra, q)
    assert pq.quantize_batch(np.zeros((0, 6), np.float32)).shape == (0, 2)
    assert pq.reconstruct_batch(np.zeros((0, 2), np.uint8)).shape == (0, 6)


def test_row_sharding_over_two_device_slots(ra):
    """SURVEY.md 8e on one GPU: a ctx with two slots on device 0 shards rows across two host
    threads exactly as it would across two GPUs (no collective; disjoint output ranges)."""
    from reductive_amd.pq import _Ctx
    ctx = _Ctx(devices=[0, 0])
    try:
        assert ctx.n_devices == 2
        M, K, dsub = 15, 256, 20
        q = synth.normalish(91, (M, K, dsub))
        x = synth.normalish(92, (20001, M * dsub))
        pq = ra.Pq(None, q, ctx=ctx)
        want = orc.quantize_batch(q, x, n_threads=8)
        assert pq.quantize_batch(x).tobytes() == want.tobytes()
        assert pq.reconstruct_batch(want).tobytes() == orc.reconstruct_batch(q, want).tobytes()
        pq.close()
    finally:
        ctx.close()


def test_concurrent_callers(ra):
    """`Pq<f32>` is Send + Sync in the reference: concurrent quantize_batch calls must work."""
    import threading
    M, K, dsub = 8, 64, 8
    q = synth.normalish(95, (M, K, dsub))
    pq = _pq(ra, q)
    xs = [synth.normalish(96 + i, (3000 + i, M * dsub)) for i in range(4)]
    wants = [orc.quantize_batch(q, x) for x in xs]
    got = [None] * 4

    def run(i):
        got[i] = pq.quantize_batch(xs[i])
    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for g, w in zip(got, wants):
        assert g.tobytes() == w.tobytes()


# ---- device-resident path (what bench.py times) + size-independent properties ------------------
def test_device_resident_and_properties_at_scale(ra):
    import torch
    M, K, dsub = 15, 256, 20
    n = 2_000_000
    q = synth.normalish(101, (M, K, dsub))
    pq = _pq(ra, q)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((n, M * dsub), device="cuda", dtype=torch.float32, generator=g)
    codes = pq.quantize_batch_device(x)
    torch.cuda.synchronize()
    assert pq.last_encode_kernel() == "k_encode_mfma16"
    # (1) sampled rows against the oracle: first / last 32k rows + a strided sample
    idx = torch.cat([torch.arange(0, 32768), torch.arange(n - 32768, n),
                     torch.arange(0, n, 97)]).cuda()
    want = orc.quantize_batch(q, x[idx].cpu().numpy(), n_threads=8)
    assert codes[idx].cpu().numpy().tobytes() == want.tobytes()
    # (2) both kernels agree on a 200k slice
    pq1 = _pq(ra, q, variant=1)
    c1 = pq1.quantize_batch_device(x[:200_000])
    assert torch.equal(c1, codes[:200_000])
    # (3) decode -> encode is the identity on codes (random codebook has no duplicate rows)
    rec = pq.reconstruct_batch_device(codes, check=True)
    again = pq.quantize_batch_device(rec)
    assert torch.equal(again, codes)
    # (4) reconstruct is a gather: checksum of rows == checksum of gathered codebook rows
    qt = torch.from_numpy(q).cuda()
    s_ref = torch.zeros(n, dtype=torch.float64, device="cuda")
    for m in range(M):
        s_ref += qt[m].double().sum(1)[codes[:, m].long()]
    assert torch.allclose(rec.double().sum(1), s_ref, rtol=0, atol=1e-9)
    # (5) row-strided device input (a column window of a wider matrix)
    wide = torch.randn((5000, 2 * M * dsub + 4), device="cuda", generator=g)
    view = wide[:, 4:4 + M * dsub]
    assert torch.equal(pq.quantize_batch_device(view), pq.quantize_batch_device(view.contiguous()))
    # (6) out-of-range code is reported on the device path
    q2 = synth.normalish(102, (4, 40, 8))
    pq2 = _pq(ra, q2)
    bad = torch.full((10, 4), 40, dtype=torch.uint8, device="cuda")
    with pytest.raises(ra.PanicError):
        pq2.reconstruct_batch_device(bad, check=True)


def test_full_size_headline_batch_every_code(ra):
    """BASELINE configs[1] at full size: all 150 M codes of the 10 M x 300 batch against the oracle
    (sharded over the host's cores), plus decode -> encode idempotence and the gather checksum on all rows."""
    import torch
    M, K, dsub = 15, 256, 20
    n = 10_000_000
    q = synth.normalish(43, (M, K, dsub))
    pq = _pq(ra, q)
    g = torch.Generator(device="cuda").manual_seed(42)
    x = torch.empty((n, M * dsub), device="cuda", dtype=torch.float32)
    for r0 in range(0, n, 1 << 20):
        x[r0:r0 + (1 << 20)].normal_(generator=g)
    codes = pq.quantize_batch_device(x)
    rec = pq.reconstruct_batch_device(codes, check=True)
    assert torch.equal(pq.quantize_batch_device(rec), codes)
    qt = torch.from_numpy(q).cuda()
    s_ref = torch.zeros(n, dtype=torch.float64, device="cuda")
    for m in range(M):
        s_ref += qt[m].double().sum(1)[codes[:, m].long()]
    assert torch.allclose(rec.double().sum(1), s_ref, rtol=0, atol=1e-9)
    del rec, s_ref
    got = codes.cpu().numpy()
    cores = os.cpu_count() or 8
    step = 2_500_000                                   # 3 GB of host memory per slice
    for r0 in range(0, n, step):
        want = orc.quantize_batch(q, x[r0:r0 + step].cpu().numpy(), n_threads=cores)
        assert got[r0:r0 + step].tobytes() == want.tobytes(), r0


def test_full_size_configs3_reconstruct_100m_codes(ra):
    """BASELINE configs[3] at full size (VERDICT r1 weakness 2): 100 M u8 code rows -> 120 GB of f32, element offsets
    far beyond 2^32 and a code matrix beyond the Infinity Cache.  Head, middle and tail row ranges and 200 k random
    rows byte for byte against the oracle gather; every row through size-independent properties: the per-row
    checksum (sum over the row of the picked centroids' sums, in f64), and decode -> encode idempotence on the
    ranges (a centroid encodes to itself).  Needs ~125 GB of free HBM."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 135 * (1 << 30):
        pytest.skip("needs 135 GB of free device memory, %.0f GB free" % (free / (1 << 30)))
    M, K, dsub = 15, 256, 20
    d = M * dsub
    n = 100_000_000
    q = synth.normalish(43, (M, K, dsub))
    pq = _pq(ra, q)
    g = torch.Generator(device="cuda").manual_seed(4242)
    codes = torch.randint(0, K, (n, M), device="cuda", dtype=torch.uint8, generator=g)
    out = torch.empty((n, d), device="cuda", dtype=torch.float32)
    assert out.numel() > (1 << 34)
    pq.reconstruct_batch_device(codes, out=out, check=True)
    # sampled rows, byte for byte
    ns = 100_000
    for s0 in (0, n // 2 - ns // 2, n - ns):
        want = orc.reconstruct_batch(q, codes[s0:s0 + ns].cpu().numpy())
        assert out[s0:s0 + ns].cpu().numpy().tobytes() == want.tobytes(), s0
        assert torch.equal(pq.quantize_batch_device(out[s0:s0 + ns]), codes[s0:s0 + ns])
    idx = torch.randint(0, n, (200_000,), device="cuda", generator=g).sort().values
    want = orc.reconstruct_batch(q, codes[idx].cpu().numpy())
    assert out[idx].cpu().numpy().tobytes() == want.tobytes()
    # every row: checksum of the row against the sum of the picked centroids' checksums (f64), in row windows
    qsum = torch.from_numpy(q).cuda().double().sum(2)          # [M][K]
    win = 10_000_000
    for r0 in range(0, n, win):
        c = codes[r0:r0 + win].long()
        ref = torch.zeros(c.shape[0], dtype=torch.float64, device="cuda")
        for m in range(M):
            ref += qsum[m][c[:, m]]
        got = out[r0:r0 + win].sum(1, dtype=torch.float64)
        assert torch.allclose(got, ref, rtol=0, atol=1e-9), r0
        del c, ref, got
    # an out-of-range code in the LAST row of a codebook with K < 256 is still reported at this size
    q2 = synth.normalish(44, (M, 200, dsub))
    pq2 = _pq(ra, q2)
    codes.clamp_(max=199)
    codes[n - 1, M - 1] = 200
    with pytest.raises(ra.PanicError):
        pq2.reconstruct_batch_device(codes, out=out, check=True)
    del out, codes
    torch.cuda.empty_cache()


def test_config0_plumbing_shape_10k_rows(ra):
    """BASELINE configs[0] (the reference's own CPU-runnable case: 10 k random vectors, d = 300, M = 15, K = 256):
    host entry points, codes and reconstructions against the oracle, every element."""
    M, K, dsub = 15, 256, 20
    q = synth.normalish(43, (M, K, dsub))
    x = synth.normalish(42, (10_000, M * dsub))
    pq = _pq(ra, q)
    codes = pq.quantize_batch(x)
    assert codes.dtype == np.uint8 and codes.tobytes() == orc.quantize_batch(q, x).tobytes()
    assert pq.reconstruct_batch(codes).tobytes() == orc.reconstruct_batch(q, codes).tobytes()


@pytest.mark.parametrize("shape", [(15, 20, "opq_fused"), (48, 16, "opq_fused")])
def test_opq_chunk_boundaries_of_a_multi_chunk_batch(ra, ctx_options, shape):
    """The TWO-KERNEL OPQ paths (rotation -> leased scratch -> encode; gather -> scratch -> rotation) walk a large batch in
    scratch chunks that are whole rounds of the rotation grid (1,179,648 rows on a 256-CU device, pqhip_opq.hip
    opq_chunk_rows): 2.5 M rows = two full chunks and a remainder.  The paths are FORCED -- the default at d = 300 and, since
    round 4, at d = 768 / M = 48 (the size of BASELINE configs[4]) is the fused kernel (one launch, no chunks), so the context
    options "opq_fused" / "opq_gather_rotation" are cleared -- and the launch log must show one rotation and one encode launch
    per chunk.  (u16 / u32 codes, K > 256, odd sub-vectors and dimensions without a fused instantiation take this path by default.)  Codes and un-rotated reconstructions around both
    chunk boundaries, at the head and at the tail against the oracle; every row through encode(decode(codes)) == codes."""
    import torch
    M, dsub, opt = shape
    K = 256
    d, n = M * dsub, 2_500_000
    q = synth.normalish(43 + M, (M, K, dsub))
    P = synth.orthonormal(44 + M, d)
    pq = _pq(ra, q, P)
    if opt:
        ctx_options(opt, 0)
    ctx_options("opq_gather_rotation", 0)
    g = torch.Generator(device="cuda").manual_seed(47)
    x = torch.randn((n, d), device="cuda", dtype=torch.float32, generator=g)
    ra.launch_log(reset=True)
    codes = pq.quantize_batch_device(x)
    log = ra.launch_log(reset=True)
    assert "k_rotate_pblock" in log and " x3" in log and "k_encode_mfma16 x3" in log and "fused" not in log, log
    rec = pq.reconstruct_batch_device(codes, check=True)
    log = ra.launch_log(reset=True)
    assert "k_reconstruct x3" in log and "k_rotate_pblock" in log and "gather" not in log, log
    cores = min(os.cpu_count() or 8, 16)
    chunk = 1_179_648
    ns = 6000 if d == 300 else 2000
    for r0 in (0, chunk - ns // 2, 2 * chunk - ns // 2, n - ns):
        xs = x[r0:r0 + ns].cpu().numpy()
        want = orc.quantize_batch(q, xs, projection=P, n_threads=cores)
        assert codes[r0:r0 + ns].cpu().numpy().tobytes() == want.tobytes(), r0
        ref = orc.reconstruct_batch(q, want, projection=P)
        got = rec[r0:r0 + ns].cpu().numpy()
        assert np.abs(got - ref).max() <= REL_TOL * np.abs(ref).max(), r0
    del x
    # size-independent: the reconstruction of a code row encodes back to that code row (a rotated centroid is its own
    # nearest centroid after the inverse rotation up to rounding; compare where the round trip is well separated)
    again = pq.quantize_batch_device(rec)
    assert (again != codes).float().mean().item() < 1e-4


def test_opq_one_million_rows_every_code_and_reconstruction(ra):
    """BASELINE configs[2] shape at 1 M rows: rotation + encode codes and the un-rotated
    reconstructions, every element, against the oracle on all host cores."""
    import torch
    M, K, dsub = 15, 256, 20
    d, n = M * dsub, 1_000_000
    q = synth.normalish(43, (M, K, dsub))
    P = synth.orthonormal(44, d)
    pq = _pq(ra, q, P)
    g = torch.Generator(device="cuda").manual_seed(46)
    x = torch.randn((n, d), device="cuda", dtype=torch.float32, generator=g)
    codes = pq.quantize_batch_device(x)
    rec = pq.reconstruct_batch_device(codes, check=True)
    cores = os.cpu_count() or 8
    want = orc.quantize_batch(q, x.cpu().numpy(), projection=P, n_threads=cores)
    assert codes.cpu().numpy().tobytes() == want.tobytes()
    ref = orc.reconstruct_batch(q, want[:100_000], projection=P)
    got = rec[:100_000].cpu().numpy()
    assert np.abs(got - ref).max() <= REL_TOL * np.abs(ref).max()
    assert got.tobytes() == ref.tobytes()


def test_statistical_roundtrip_loss(ra, kats):
    """pq.rs:431-440 analogue: a trained 7-bit quantizer on U[0,1) 256x20 reconstructs with mean
    Euclidean loss < 0.08 (training itself is out of scope: a few Lloyd steps in numpy)."""
    st = kats["statistical"]
    n, d, M, K = st["n"], st["d"], st["n_subquantizers"], 1 << st["n_bits"]
    dsub = d // M
    x = synth.uniform01(111, (n, d))
    # random-instance initialisation with K DISTINCT rows (kmeans.rs:52-87)
    q = np.stack([x[np.argsort(synth.splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(977 * m)),
                               kind="stable")[:K], m * dsub:(m + 1) * dsub]
                  for m in range(M)]).astype(np.float32)
    for _ in range(10):
        codes = orc.quantize_batch(q, x)
        for m in range(M):
            for j in range(K):
                sel = codes[:, m] == j
                if sel.any():
                    q[m, j] = x[sel, m * dsub:(m + 1) * dsub].mean(0)
    pq = _pq(ra, q)
    rec = pq.reconstruct_batch(pq.quantize_batch(x))
    loss = np.sqrt(((x - rec) ** 2).sum(1)).mean()
    assert loss < st["loss_bound"]


def test_shape_sweep_all_kernel_instantiations(ra):
    """Every (T, DP, even/odd sub-dimension) instantiation of the MFMA kernels plus odd shapes:
    seeded random (M, K, dsub, n), codes must equal the oracle's for every variant."""
    rng = np.random.RandomState(12345)
    shapes = []
    for K in (1, 2, 31, 32, 33, 64, 65, 128, 129, 255, 256):        # T = 1, 2, 4, 8 with padding
        for dsub in (1, 2, 3, 4, 5, 8, 11, 12, 16, 17, 20, 24, 27, 28, 31, 32):
            shapes.append((int(rng.randint(1, 4)), K, dsub, int(rng.randint(1, 200))))
    rng.shuffle(shapes)
    shapes = shapes[:72]
    for i, (M, K, dsub, n) in enumerate(shapes):
        q = synth.normalish(5000 + i, (M, K, dsub))
        x = synth.normalish(6000 + i, (n, M * dsub))
        want = orc.quantize_batch(q, x)
        for variant in (0, 2, 4) + ((9,) if _has_mfma16(K, dsub) else ()):
            got = _pq(ra, q, variant=variant).quantize_batch(x)
            assert got.tobytes() == want.tobytes(), (M, K, dsub, n, variant)
    # device rows that start off the 16-byte grid (row stride not a multiple of 4 floats): the same
    # kernel serves them, its loads are dword-aligned wide loads
    import torch
    M, K, dsub = 3, 64, 8
    q = synth.normalish(7001, (M, K, dsub))
    wide = torch.from_numpy(synth.normalish(7002, (500, M * dsub + 3))).cuda()
    view = wide[:, 1:1 + M * dsub]
    pq = _pq(ra, q, variant=4)
    got = pq.quantize_batch_device(view)
    assert "vec4" in pq.last_encode_kernel()
    assert _pq(ra, q).quantize_batch_device(view).cpu().numpy().tobytes() == got.cpu().numpy().tobytes()   # auto: k_encode_smallk
    want = orc.quantize_batch(q, view.cpu().numpy())
    assert got.cpu().numpy().tobytes() == want.tobytes()
    pq9 = _pq(ra, q, variant=9)                                   # the 16x16x4 kernel loads single dwords: any 4-byte alignment
    assert pq9.quantize_batch_device(view).cpu().numpy().tobytes() == want.tobytes()
    assert pq9.last_encode_kernel() == "k_encode_mfma16"


def test_cluster_assignments_entry_point(ra, kats):
    """the k-means assignment step (kmeans.rs:133-159) through its own C-ABI
    entry point, usize-wide indices, MFMA kernel with 32-bit codes on the device path."""
    import torch
    k = kats["cluster_assignments"]
    c = np.array(k["centroids"], np.float32)
    x = np.array(k["instances"], np.float32)
    assert ra.cluster_assignments(c, x).tolist() == k["assignments"]
    assert ra.cluster_assignments(c, np.asfortranarray(x), dtype=np.uint32).tolist() == k["assignments"]
    for (K, dim, n) in [(256, 20, 5000), (128, 2, 3000), (300, 8, 500), (17, 33, 257)]:
        cen = synth.normalish(8100 + K, (K, dim))
        xs = synth.normalish(8200 + K, (n, dim))
        want = orc.cluster_assignments(cen, xs)
        assert ra.cluster_assignments(cen, xs).tolist() == want.tolist()
    # device path with 32-bit codes runs the MFMA kernel
    K, dim, n = 256, 20, 100_000
    cen = synth.normalish(8301, (K, dim))
    pq = ra.Pq(None, cen[None])
    xd = torch.from_numpy(synth.normalish(8302, (n, dim))).cuda()
    out = torch.empty((n, 1), dtype=torch.int32, device="cuda")
    rc = ra.lib().pqhip_quantize_batch_f32_dev(pq._cb(), 0, xd.data_ptr(), n, dim, out.data_ptr(), 4, 1,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert pq.last_encode_kernel() == "k_encode_mfma16"
    want = orc.cluster_assignments(cen, xd.cpu().numpy())
    assert out[:, 0].cpu().numpy().tolist() == want.tolist()
    pq.set_encode_variant(4)   # the 32x32x2 kernel with 32-bit codes
    out.zero_()
    rc = ra.lib().pqhip_quantize_batch_f32_dev(pq._cb(), 0, xd.data_ptr(), n, dim, out.data_ptr(), 4, 1,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert pq.last_encode_kernel().startswith("k_encode_mfma_lds3")
    assert out[:, 0].cpu().numpy().tolist() == want.tolist()


def test_device_api_streams_and_opq_scratch_ordering(ra):
    """Device entry points are asynchronous on the caller's stream; two streams sharing one OPQ
    codebook must serialise on its scratch buffer (event-ordered), results unchanged."""
    import torch
    M, K, dsub = 8, 64, 8
    d = M * dsub
    q = synth.normalish(9101, (M, K, dsub))
    P = synth.orthonormal(9102, d)
    pq = _pq(ra, q, P)
    xs = [torch.from_numpy(synth.normalish(9110 + i, (20000 + 777 * i, d))).cuda() for i in range(4)]
    wants = [orc.quantize_batch(q, x.cpu().numpy(), projection=P, n_threads=8) for x in xs]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = []
    torch.cuda.synchronize()
    for i, x in enumerate(xs):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(pq.quantize_batch_device(x))
    torch.cuda.synchronize()
    for o, w in zip(outs, wants):
        assert o.cpu().numpy().tobytes() == w.tobytes()
    # reconstruct through the same scratch
    recs = []
    for i, o in enumerate(outs):
        with torch.cuda.stream(streams[i % 2]):
            recs.append(pq.reconstruct_batch_device(o, check=(i == 3)))
    torch.cuda.synchronize()
    for r, w in zip(recs, wants):
        ref = orc.reconstruct_batch(q, w, projection=P)
        assert np.abs(r.cpu().numpy() - ref).max() <= REL_TOL * np.abs(ref).max()


def test_trained_codebook_and_opq_at_scale(ra):
    """Realistic data: U[0,1) vectors, a codebook refined by Lloyd steps (assignment on the GPU,
    centroid update in numpy -- training itself is out of scope), 400k x 300 PQ and 150k x 300
    OPQ batches compared code-for-code with the oracle (natural near-ties included)."""
    M, K, dsub = 15, 256, 20
    d = M * dsub
    x = synth.uniform01(9201, (400_000, d))
    q = np.stack([x[m * 1000:m * 1000 + K, m * dsub:(m + 1) * dsub] for m in range(M)]).astype(np.float32).copy()
    for _ in range(3):
        codes = ra.Pq(None, q).quantize_batch(x[:50_000])
        for m in range(M):
            sub = x[:50_000, m * dsub:(m + 1) * dsub]
            for j in np.unique(codes[:, m]):
                q[m, j] = sub[codes[:, m] == j].mean(0)
    want = orc.quantize_batch(q, x, n_threads=16)
    for variant in (0, 4):
        got = _pq(ra, q, variant=variant).quantize_batch(x)
        assert got.tobytes() == want.tobytes(), variant
    P = synth.orthonormal(9202, d)
    xo = x[:150_000]
    want_o = orc.quantize_batch(q, xo, projection=P, n_threads=16)
    pq = _pq(ra, q, P)
    assert pq.quantize_batch(xo).tobytes() == want_o.tobytes()
    rec = pq.reconstruct_batch(want_o[:20_000])
    ref = orc.reconstruct_batch(q, want_o[:20_000], projection=P)
    assert np.abs(rec - ref).max() <= REL_TOL * np.abs(ref).max()


# ---- randomized dispatch coverage ---------------------------------------------------------------
def _fuzz_cases(count):
    """Deterministic pseudo-random shapes (splitmix-style LCG, no numpy RNG stream dependence)."""
    state = [0x9E3779B97F4A7C15]

    def nxt(lo, hi):
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        return lo + (state[0] >> 33) % (hi - lo + 1)

    ks = [1, 2, 5, 16, 31, 32, 33, 64, 100, 255, 256, 257, 300]
    cases = []
    for i in range(count):
        M, dsub = nxt(1, 40), nxt(1, 36)
        if i % 7 == 0:
            dsub = 4 * nxt(1, 9)                      # keep the 16-byte paths well represented
        K = ks[nxt(0, len(ks) - 1)]
        n = nxt(1, 700)
        opq = bool(nxt(0, 1)) and M * dsub <= 640
        cases.append((i, M, K, dsub, n, opq, nxt(0, 4), nxt(0, 3), nxt(0, 3)))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(64), ids=lambda c: "i%d-M%d-K%d-ds%d-n%d-opq%d-p%d-o%d-c%d" % c)
def test_randomized_shapes_host_and_device_entry_points(ra, case):
    """Random (M, K, dsub, n, OPQ) through the host entry points, and -- for u8 codes -- through
    the device entry points with padded row strides and base pointers off the 16-byte grid, so that
    every vector/scalar-load dispatch decision of the library is exercised against the oracle."""
    import torch
    i, M, K, dsub, n, opq, pad, off, cpad = case
    d = M * dsub
    q = synth.normalish(1000 + i, (M, K, dsub))
    x = synth.normalish(2000 + i, (n, d))
    P = synth.orthonormal(3000 + i, d) if opq else None
    dt = np.uint8 if K <= 256 else np.uint16
    want = orc.quantize_batch(q, x, projection=P, dtype=dt)
    ref = orc.reconstruct_batch(q, want, projection=P)
    pq = _pq(ra, q, P)
    assert pq.quantize_batch(x, dtype=dt).tobytes() == want.tobytes()
    assert pq.reconstruct_batch(want).tobytes() == ref.tobytes()
    if K > 256:
        return
    dev = torch.device("cuda:0")
    xs = d + pad
    xbuf = torch.zeros(off + n * xs, dtype=torch.float32, device=dev)
    xv = xbuf[off:off + n * xs].view(n, xs)[:, :d]
    xv.copy_(torch.from_numpy(x))
    cbuf = torch.full((n, M + cpad), 0xEE, dtype=torch.uint8, device=dev)
    cv = cbuf[:, :M]
    pq.quantize_batch_device(xv, out=cv)
    torch.cuda.synchronize()
    assert cv.cpu().numpy().tobytes() == want.tobytes()
    if cpad:
        assert (cbuf[:, M:] == 0xEE).all().item()          # padding bytes untouched
    obuf = torch.full((off + n * xs,), -7.0, dtype=torch.float32, device=dev)
    ov = obuf[off:off + n * xs].view(n, xs)[:, :d]
    pq.reconstruct_batch_device(cv, out=ov, check=True)
    torch.cuda.synchronize()
    assert ov.cpu().numpy().tobytes() == ref.tobytes()
    if pad:
        assert (obuf[off:off + n * xs].view(n, xs)[:, d:] == -7.0).all().item()
    assert (obuf[:off] == -7.0).all().item()


# ---- "next" row: the k-means step of training (kmeans.rs:166-198, 308-360) ----------------------
KM_SHAPES = [  # n, M, K, dsub
    (5000, 15, 256, 20),    # headline codebook shape
    (3000, 4, 16, 8),
    (777, 3, 5, 7),         # odd sub-dimension, K not a power of two
    (4097, 1, 256, 32),     # plain k-means (M = 1)
    (2000, 2, 300, 6),      # K > 256: 32-bit codes, anchor assignment kernel
    (500, 2, 8, 40),        # dsub > 32: anchor assignment kernel
    (64, 1, 64, 3),         # as many centroids as instances
    (10, 2, 16, 4),         # fewer instances than centroids: most clusters stay empty
    (1, 1, 2, 1),
]


def _km_inputs(n, M, K, dsub, seed):
    x = synth.normalish(seed, (n, M * dsub))
    pick = synth.codes_u8(seed + 1, (K, 4), 256).astype(np.int64)
    rows = (pick[:, 0] * 65536 + pick[:, 1] * 256 + pick[:, 2]) % n
    q0 = np.stack([x[rows, m * dsub:(m + 1) * dsub] for m in range(M)]).copy()
    if K > n:
        q0 += synth.normalish(seed + 2, q0.shape) * np.float32(0.1)
    return q0, x


@pytest.mark.parametrize("shape", KM_SHAPES)
def test_kmeans_iterations_match_oracle(ra, shape):
    n, M, K, dsub = shape
    q0, x = _km_inputs(n, M, K, dsub, 700 + n)
    for iters in (1, 3):
        want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=iters)
        got_q, got_loss = ra.kmeans_iterations(q0, x, n_iterations=iters)
        assert got_q.tobytes() == want_q.tobytes(), iters
        assert got_loss.tobytes() == want_loss.tobytes(), iters
    # loss not asked for (opq.rs:227-245 discards it): same centroids
    got_q2, none = ra.kmeans_iterations(q0, x, n_iterations=3, want_loss=False)
    assert none is None and got_q2.tobytes() == want_q.tobytes()
    assert q0.tobytes() == _km_inputs(n, M, K, dsub, 700 + n)[0].tobytes()   # inputs untouched


def test_kmeans_many_row_windows(ra, ctx_options):
    """The update runs window by window (overlapped with the assignment of the next window); with
    the window shrunk to 64 rows the carried chains and counts cross dozens of windows."""
    ctx_options("kmeans_window_rows", 64)
    for (n, M, K, dsub) in [(5000, 15, 256, 20), (1999, 3, 5, 7), (130, 2, 300, 6), (700, 2, 4, 68), (900, 1, 3, 300)]:
        q0, x = _km_inputs(n, M, K, dsub, 900 + n)
        want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=2)
        got_q, got_loss = ra.kmeans_iterations(q0, x, n_iterations=2)
        assert got_q.tobytes() == want_q.tobytes()
        assert got_loss.tobytes() == want_loss.tobytes()
        # the lane-per-chain form of the walk (used for very wide sub-vectors) on the same inputs
        ctx_options("kmeans_lane_form", 1)
        got_q, got_loss = ra.kmeans_iterations(q0, x, n_iterations=2)
        ctx_options("kmeans_lane_form", 0)
        assert got_q.tobytes() == want_q.tobytes()
        assert got_loss.tobytes() == want_loss.tobytes()


def test_kmeans_kat_fixed_point_and_three_spheres(ra, kats):
    # kmeans.rs:401-434: started from the expected centroids, the KAT's assignments are what
    # cluster_assignments yields, and update_centroids must reproduce the expected means exactly
    k = kats["update_centroids"]
    x = np.array(k["instances"], np.float32)
    c = np.array(k["expected"], np.float32)
    assert orc.cluster_assignments(c, x).tolist() == k["assignments"]
    q, loss = ra.kmeans_iterations(c[None], x, 1)
    assert q[0].tolist() == k["expected"]
    assert loss[0] == orc.mean_squared_error(c, x, k["assignments"])
    q, _ = ra.kmeans_iterations(c[None], np.asfortranarray(x), 1)            # axis-1 layout
    assert q[0].tolist() == k["expected"]
    # kmeans.rs:436-480 (k_means_3) with our own sample stream
    k = kats["k_means_3"]
    centers = np.array(k["centers"], np.float32)
    pts = np.concatenate([cc + np.float32(k["sigma"]) * synth.normalish(520 + i, (k["n_samples"], 2))
                          for i, cc in enumerate(centers)]).astype(np.float32)
    q, loss = ra.kmeans_iterations(pts[[3, 14, 30]][None], pts, k["iterations"])
    assert sorted(np.rint(q[0]).astype(int).tolist()) == k["expected_rounded_sorted"]
    want_q, want_loss = orc.kmeans_iterations(pts[[3, 14, 30]][None], pts, k["iterations"])
    assert q.tobytes() == want_q.tobytes() and loss.tobytes() == want_loss.tobytes()


def test_kmeans_special_values_and_empty_clusters(ra):
    M, K, dsub, n = 2, 8, 4, 600
    q0, x = _km_inputs(n, M, K, dsub, 810)
    q0[0, 5] = 1e6                       # never chosen -> empty -> zero vector (kmeans.rs:192-196)
    x[7, 1] = np.inf                     # poisons one centroid of m = 0 and its loss
    x[9, 6] = np.nan
    x[11] = -0.0
    want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=2)
    got_q, got_loss = ra.kmeans_iterations(q0, x, n_iterations=2)
    assert got_q.tobytes() == want_q.tobytes()
    assert got_loss.tobytes() == want_loss.tobytes()
    assert np.isnan(want_loss).any()
    one_q, _ = ra.kmeans_iterations(q0, x, n_iterations=1)
    assert (one_q[0, 5] == 0).all()
    # >= 3 iterations replay a captured graph: the non-finite centroids that appear after the first
    # update are then handled through the device-side flag, not by the host switching kernels
    want_q4, want_loss4 = orc.kmeans_iterations(q0, x, n_iterations=5)
    got_q4, got_loss4 = ra.kmeans_iterations(q0, x, n_iterations=5)
    assert got_q4.tobytes() == want_q4.tobytes() and got_loss4.tobytes() == want_loss4.tobytes()


def test_kmeans_graph_replay_equals_eager_loop(ra, ctx_options):
    """Small training sets run all iterations but the last as one replayed hipGraph: same bits as
    the eager loop (context option "kmeans_no_graph") and as the oracle, for u8 and for odd shapes."""
    for (n, M, K, dsub, iters) in [(4000, 15, 256, 20, 12), (777, 3, 5, 7, 6), (2000, 2, 16, 40, 4)]:
        q0, x = _km_inputs(n, M, K, dsub, 2200 + n)
        want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=iters, n_threads=8)
        got_q, got_loss = ra.kmeans_iterations(q0, x, n_iterations=iters)
        assert got_q.tobytes() == want_q.tobytes() and got_loss.tobytes() == want_loss.tobytes()
        ctx_options("kmeans_no_graph", 1)
        eager_q, eager_loss = ra.kmeans_iterations(q0, x, n_iterations=iters)
        ctx_options("kmeans_no_graph", 0)
        assert eager_q.tobytes() == want_q.tobytes() and eager_loss.tobytes() == want_loss.tobytes()


def test_kmeans_device_resident_strided_and_at_scale(ra):
    import torch
    n, M, K, dsub = 200_000, 15, 256, 20
    d = M * dsub
    q0, x = _km_inputs(n, M, K, dsub, 820)
    want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=2, n_threads=16)
    xbuf = torch.zeros((n, d + 4), dtype=torch.float32, device="cuda:0")
    xv = xbuf[:, :d]
    xv.copy_(torch.from_numpy(x))
    got_q, got_loss = ra.kmeans_iterations(q0, xv, n_iterations=2)
    assert got_q.tobytes() == want_q.tobytes()
    assert got_loss.tobytes() == want_loss.tobytes()
    # loss decreases from one Lloyd iteration to the next on this data
    _, l1 = ra.kmeans_iterations(q0, xv, n_iterations=1)
    assert (got_loss <= l1).all()


def test_kmeans_f32_counts_saturate_like_the_reference(ra):
    """kmeans.rs:184: counts are f32 incremented by one, so they stop at 2^24."""
    n = (1 << 24) + 5
    x = np.full((n, 1), 1.0, np.float32)
    x[::2] = 3.0
    q, loss = ra.kmeans_iterations(np.zeros((1, 1, 1), np.float32), x, 1)
    want_q, want_loss = orc.kmeans_iterations(np.zeros((1, 1, 1), np.float32), x, 1)
    assert q.tobytes() == want_q.tobytes() and loss.tobytes() == want_loss.tobytes()


def test_train_pq_statistical_loss(ra, kats):
    """pq.rs:431-440 (quantize_with_type): train on U[0,1) 256 x 20, M = 10, 7 bits... the
    reference's bound on the mean Euclidean reconstruction loss must hold for GPU-trained codebooks."""
    k = kats["statistical"]
    x = synth.uniform01(830, (k["n"], k["d"]))
    pq = ra.train_pq(k["n_subquantizers"], k["n_bits"], 10, 1, x, rng=np.random.default_rng(5))
    codes = pq.quantize_batch(x)
    rec = pq.reconstruct_batch(codes)
    loss = np.sqrt(((x - rec) ** 2).sum(axis=1)).mean()
    assert loss < k["loss_bound"], loss
    with pytest.raises(ra.ReductiveError):
        ra.train_pq(0, 4, 10, 1, x)
    with pytest.raises(ra.ReductiveError):
        ra.train_pq(10, 9, 10, 1, x)          # 2^9 > 256 instances
    with pytest.raises(ra.ReductiveError):
        ra.train_pq(3, 4, 10, 1, x)           # 20 % 3 != 0
    with pytest.raises(ra.ReductiveError):
        ra.train_pq(10, 4, 0, 1, x)
    with pytest.raises(ra.ReductiveError):
        ra.train_pq(10, 4, 10, 0, x)


# ---- "next" row 2: lookup path of a resident quantized matrix (select + reconstruct + rescale) ----
@pytest.mark.parametrize("shape", [(5000, 15, 256, 20, False), (3000, 5, 40, 7, False), (2000, 8, 64, 8, True),
                                   (900, 3, 300, 6, False), (400, 4200, 2, 1, False)])
@pytest.mark.parametrize("two_pass", [0, 1])
def test_lookup_rows_matches_select_reconstruct_scale(ra, ctx_options, shape, two_pass):
    """(two_pass: the one-kernel lookup, and the select-then-reconstruct form that large resident matrices take by default --
    forced here through the context option "lookup_two_pass")"""
    import torch
    ctx_options("lookup_two_pass", two_pass)
    N, M, K, dsub, opq = shape
    d = M * dsub
    q = synth.normalish(1100 + N, (M, K, dsub))
    P = synth.orthonormal(1101 + N, d) if opq else None
    codes = synth.codes_u8(1102 + N, (N, M), min(K, 256))
    scales = np.abs(synth.normalish(1103 + N, (N,))) + np.float32(0.5)
    n = 3333
    pick = synth.codes_u8(1104 + N, (n, 3), 256).astype(np.int64)
    rows = (pick[:, 0] * 65536 + pick[:, 1] * 256 + pick[:, 2]) % N          # repeats and any order
    pq = _pq(ra, q, P)
    want = orc.reconstruct_batch(q, codes[rows], projection=P)
    dev = torch.device("cuda:0")
    tc, tr, ts = torch.from_numpy(codes).to(dev), torch.from_numpy(rows).to(dev), torch.from_numpy(scales).to(dev)
    got = pq.reconstruct_rows_device(tc, tr, check=True).cpu().numpy()
    assert got.tobytes() == want.tobytes()
    got_s = pq.reconstruct_rows_device(tc, tr, scales=ts, check=True).cpu().numpy()
    assert got_s.tobytes() == (want * scales[rows][:, None]).astype(np.float32).tobytes()
    # strided code matrix and output
    cbuf = torch.zeros((N, M + 3), dtype=torch.uint8, device=dev)
    cbuf[:, :M] = tc
    obuf = torch.full((n, d + 4), -3.0, dtype=torch.float32, device=dev)
    pq.reconstruct_rows_device(cbuf[:, :M], tr, scales=ts, out=obuf[:, :d], check=True)
    assert obuf[:, :d].cpu().numpy().tobytes() == got_s.tobytes() and (obuf[:, d:] == -3.0).all().item()
    # ndarray `select` panics on an index out of bounds
    bad = tr.clone()
    bad[17] = N
    with pytest.raises(ra.PanicError):
        pq.reconstruct_rows_device(tc, bad, check=True)
    bad[17] = -1
    with pytest.raises(ra.PanicError):
        pq.reconstruct_rows_device(tc, bad, scales=ts, check=True)
    assert pq.reconstruct_rows_device(tc, tr[:0]).shape == (0, d)
    # interleaved records (codes + scale of a row in one 32/64-byte record: one line per lookup): same bits
    if K <= 256:
        rec, off = ra.Pq.interleave_records(tc, ts)
        got_r = pq.reconstruct_records_device(rec, off, tr, check=True).cpu().numpy()
        assert got_r.tobytes() == got_s.tobytes()
        with pytest.raises(ra.PanicError):
            pq.reconstruct_records_device(rec, off, bad, check=True)


# ---- "next" row: the OPQ training iteration without its LAPACK calls (opq.rs:156-195) ------------
@pytest.mark.parametrize("shape", [(700, 3, 8, 4), (5000, 15, 256, 20), (1030, 2, 37, 7), (300, 1, 300, 16), (513, 4, 16, 33)])
def test_opq_train_step_matches_oracle(ra, shape):
    import torch
    n, M, K, dsub = shape
    d = M * dsub
    q0, x = _km_inputs(n, M, K, dsub, 1600 + n)
    P = synth.orthonormal(1601 + n, d)
    want_q, want_cross = orc.opq_train_step(q0, P, x, n_threads=8)
    xd = torch.from_numpy(x).cuda()
    got_q, got_cross = ra.opq_train_step(q0, P, xd)
    assert got_q.tobytes() == want_q.tobytes()
    assert got_cross.tobytes() == want_cross.tobytes()
    # a padded, strided resident matrix gives the same bits
    wide = torch.zeros((n, d + 5), device="cuda")
    wide[:, :d] = xd
    got_q2, got_cross2 = ra.opq_train_step(q0, P, wide[:, :d])
    assert got_q2.tobytes() == want_q.tobytes() and got_cross2.tobytes() == want_cross.tobytes()


def test_at_dot_b_row_blocks_and_groups(ra, ctx_options):
    """`a.t().dot(&b)`: chains restart every 256 rows, block results are added in row order -- across the launch groups of
    partial matrices too (shrunk to 64 MiB here: 181 row blocks of a 300 x 300 output per group).  Outputs wider than one
    workgroup's 320 x 320 (several output blocks), ragged last tiles, unaligned rows."""
    import torch
    for (n, da, db) in [(1, 3, 5), (255, 7, 7), (256, 64, 65), (257, 20, 300), (3000, 130, 40), (600, 700, 330), (513, 321, 17)]:
        a = synth.normalish(1700 + n, (n, da))
        b = synth.normalish(1701 + n, (n, db))
        got = ra.at_dot_b(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        assert got.tobytes() == orc.at_dot_b(a, b, n_threads=8).tobytes(), (n, da, db)
    ctx_options("cross_product_group_bytes", 64 << 20)
    n = 256 * 700 + 11                                     # four groups of row blocks at d = 300
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn((n, 300), device="cuda", generator=g)
    b = torch.randn((n, 300), device="cuda", generator=g)
    ra.launch_log(reset=True)
    got = ra.at_dot_b(a, b)
    assert ra.launch_log(reset=True) == "k_atb_rowblock x4 + k_atb_fold x4"
    want = orc.at_dot_b(a.cpu().numpy(), b.cpu().numpy(), n_threads=min(os.cpu_count() or 8, 16))
    assert got.tobytes() == want.tobytes()
    # float-tolerance mode: a plain split-K product (parts of many 256-row blocks), within north_star's 1e-5 relative
    ctx_options("cross_product_exact", 0)
    fast = ra.at_dot_b(a, b)
    assert np.abs(fast - want).max() <= REL_TOL * np.abs(want).max()
    assert fast.tobytes() != want.tobytes()                # (it IS another summation order)


def test_opq_train_step_gathers_the_reconstruction_inside_the_cross_product(ra):
    """opq.rs:176-191 without the reconstructed matrix in memory: for sub-vectors of whole 16-byte pieces the cross-product
    kernel assembles the rows of R from the codebook; other sub-vectors reconstruct first.  Same bits either way (the
    oracle reconstructs, then multiplies)."""
    import torch
    for (n, M, K, dsub, gathered) in [(3000, 15, 256, 20, True), (1500, 3, 300, 8, True), (1200, 5, 16, 6, False),
                                      (150_011, 15, 256, 20, True)]:          # 587 row blocks, a ragged last one
        d = M * dsub
        q0, x = _km_inputs(n, M, K, dsub, 1650 + n)
        P = synth.orthonormal(1651 + n, d)
        want_q, want_cross = orc.opq_train_step(q0, P, x, n_threads=min(os.cpu_count() or 8, 16))
        ra.launch_log(reset=True)
        got_q, got_cross = ra.opq_train_step(q0, P, torch.from_numpy(x).cuda())
        log = ra.launch_log(reset=True)
        assert ("k_atb_rowblock<gather>" in log) == gathered and ("k_reconstruct" in log) == (not gathered), log
        assert got_q.tobytes() == want_q.tobytes() and got_cross.tobytes() == want_cross.tobytes()


def test_two_pass_lookup_equals_one_pass_on_a_matrix_beyond_the_infinity_cache(ra, ctx_options):
    """Lookups into a resident code matrix of 24 M rows (360 MB > 256 MB: the two-pass form is the DEFAULT there; DESIGN.md
    section 5, lookup): every output row equals the one-kernel form's bit for bit -- split layout, strided wide matrix and
    interleaved records -- and the oracle's on a sample; an index out of range is reported by the select pass."""
    import torch
    N, M, K, dsub, n = 24_000_000, 15, 256, 20, 1_500_000
    q = synth.normalish(1250, (M, K, dsub))
    pq = _pq(ra, q)
    g = torch.Generator(device="cuda").manual_seed(1251)
    codes = torch.randint(0, K, (N, M), device="cuda", dtype=torch.uint8, generator=g)
    rows = torch.randint(0, N, (n,), device="cuda", dtype=torch.int64, generator=g)
    scales = torch.rand((N,), device="cuda", generator=g) + 0.5
    ra.launch_log(reset=True)
    two = pq.reconstruct_rows_device(codes, rows, scales=scales, check=True)
    assert ra.launch_log(reset=True).endswith("k_select_code_rows16 + k_reconstruct<scaled>")
    ctx_options("lookup_two_pass", 0)
    one = pq.reconstruct_rows_device(codes, rows, scales=scales, check=True)
    assert ra.launch_log(reset=True) == "k_reconstruct<lookup>"
    assert torch.equal(one, two)
    pick = rows[:3000].cpu().numpy()
    want = orc.reconstruct_batch(q, codes[rows[:3000]].cpu().numpy()) * scales[rows[:3000]].cpu().numpy()[:, None]
    assert two[:3000].cpu().numpy().tobytes() == want.astype(np.float32).tobytes()
    del one
    ctx_options("lookup_two_pass", 2)
    rec, off = ra.Pq.interleave_records(codes[:20_000_000], scales[:20_000_000])          # 640 MB of 32-byte records
    r2 = rows % 20_000_000
    got_r = pq.reconstruct_records_device(rec, off, r2, check=True)
    assert "k_select_code_rows16" in ra.launch_log(reset=True)
    ctx_options("lookup_two_pass", 0)
    assert torch.equal(got_r, pq.reconstruct_records_device(rec, off, r2, check=True))
    ctx_options("lookup_two_pass", 2)
    bad = rows.clone()
    bad[n - 1] = N
    with pytest.raises(ra.PanicError):
        pq.reconstruct_rows_device(codes, bad, scales=scales, check=True)
    _ = pick


def test_train_opq_statistical_loss(ra, kats):
    """opq.rs:331-340 (quantize_with_opq): U[0,1) 256 x 20, M = 10, 7 bits, 10 iterations -> mean
    Euclidean reconstruction loss < 0.1 for a GPU-trained OPQ; the projection stays orthonormal."""
    k = kats["opq_statistical"]
    x = synth.uniform01(1800, (k["n"], k["d"]))
    pq = ra.train_opq(k["n_subquantizers"], k["n_bits"], k["n_iterations"], 1, x, rng=np.random.default_rng(7))
    P = pq.projection()
    assert np.abs(P @ P.T - np.eye(k["d"], dtype=np.float32)).max() < 1e-4
    rec = pq.reconstruct_batch(pq.quantize_batch(x))
    loss = np.sqrt(((x - rec) ** 2).sum(axis=1)).mean()
    assert loss < k["loss_bound"], loss
    with pytest.raises(ra.ReductiveError):
        ra.train_opq(3, 4, 10, 1, x)


def test_resident_matrix_handle_feeds_the_training_entry_points(ra):
    """pqhip_matrix_upload_f32: what a host-only binding (Rust) uses to keep the instances in HBM
    across the k-means / OPQ iterations -- strided host matrix in, *_dev entry points on its pointer."""
    from reductive_amd.pq import default_ctx
    L = ra.lib()
    n, M, K, dsub = 3000, 3, 16, 8
    d = M * dsub
    q0, x = _km_inputs(n, M, K, dsub, 1900)
    big = np.zeros((n, 2 * d + 3), np.float32)
    big[:, 1:1 + 2 * d:2] = x                              # column stride 2, row stride 2d + 3
    view = big[:, 1:1 + 2 * d:2]
    h = ctypes.c_void_p()
    rc = L.pqhip_matrix_upload_f32(default_ctx().handle, 0, view.ctypes.data, n, d, view.strides[0] // 4,
                                   view.strides[1] // 4, ctypes.byref(h))
    assert rc == 0 and L.pqhip_matrix_rows(h) == n
    ptr = L.pqhip_matrix_device_ptr(h)
    fp = ctypes.POINTER(ctypes.c_float)
    q = q0.copy()
    loss = np.zeros(M, np.float32)
    rc = L.pqhip_kmeans_iterations_f32_dev(default_ctx().handle, 0, q.ctypes.data_as(fp), M, K, dsub, ptr, n, d, 3,
                                           loss.ctypes.data_as(fp), None)
    assert rc == 0
    wq, wl = orc.kmeans_iterations(q0, x, 3)
    assert q.tobytes() == wq.tobytes() and loss.tobytes() == wl.tobytes()
    P = synth.orthonormal(1901, d)
    q = q0.copy()
    cross = np.zeros((d, d), np.float32)
    rc = L.pqhip_opq_train_step_f32_dev(default_ctx().handle, 0, q.ctypes.data_as(fp), M, K, dsub,
                                        P.ctypes.data_as(fp), ptr, n, d, cross.ctypes.data_as(fp), None)
    assert rc == 0
    wq, wc = orc.opq_train_step(q0, P, x)
    assert q.tobytes() == wq.tobytes() and cross.tobytes() == wc.tobytes()
    L.pqhip_matrix_destroy(h)


@pytest.mark.parametrize("d", [4, 20, 32, 40, 48, 52, 64, 72, 80, 96, 100, 112, 160, 176, 240, 256, 260, 272, 288, 300, 320, 324, 336, 352,
                               512, 600, 636, 640, 704, 768, 1024, 1280, 1284])
def test_rotation_kernel_every_burst_structure(ra, d):
    """k_rotate_pblock8 is compiled for the eight combinations of (rule-2 split d > 256, odd number of full 32-k bursts,
    partial last burst); d = 4 .. 636 walks all of them plus the edges -- no full burst at all (d < 32), one burst, the
    largest d whose P block fits LDS (636) and the first one that falls back (640) -- with enough rows (two row groups,
    14 tiles for some waves, a ragged last tile) that the burst ring crosses tile and workgroup boundaries.  Bit-exact
    against the oracle's rule-2 chains; non-contiguous rows and a wider output too.
    k_rotate_pblock9 (16x16x4; forced here, the default of the gather form) has its own eight combinations -- rule-2
    split, odd number of 16-k bursts, d % 16 != 0 -- times one to four 16-column tiles in the last column block
    (d = 272 / 336: one, 288: two, 300: three); beyond d = 640 it runs 32-column blocks (auto there: 704, 768, 1024, 1280;
    1284 is the first d that falls back to the slab kernel)."""
    import torch
    n = 4608 + 2 * 384 + 17
    x = synth.normalish(2100 + d, (n, d))
    P = synth.orthonormal(2101 + d, d) if d <= 320 else synth.normalish(2101 + d, (d, d)) * np.float32(0.05)
    want = orc.rotate(x, P)
    try:
        for variant in (0, 9, 8):   # auto (k_rotate_pblock8 for plain rotation), the 16x16x4 form, the 32x32x2 form
            ra.set_rotation_variant(variant)
            got = ra.rotate(torch.from_numpy(x).cuda(), P).cpu().numpy()
            assert got.tobytes() == want.tobytes(), variant
            if d % 4 == 0 and d >= 20:
                wide = torch.zeros((n, d + 8), device="cuda")
                wide[:, :d] = torch.from_numpy(x).cuda()
                got2 = ra.rotate(wide[:, :d], P).cpu().numpy()
                assert got2.tobytes() == want.tobytes(), variant
    finally:
        ra.set_rotation_variant(0)


def test_rotate_entry_point_and_gaussian_opq(ra, kats):
    """`instances.dot(&projection)` alone (rule 2, bit-exact) and `GaussianOpq::train_pq_using`
    (gaussian_opq.rs:99-108: mean Euclidean loss < 0.12 on U[0,1) 256 x 20)."""
    import torch
    for (n, d) in [(333, 20), (1000, 300), (70, 515)]:
        x = synth.normalish(2000 + d, (n, d))
        P = synth.orthonormal(2001 + d, d)
        got = ra.rotate(torch.from_numpy(x).cuda(), P).cpu().numpy()
        assert got.tobytes() == orc.rotate(x, P).tobytes()
    k = kats["gaussian_opq_statistical"]
    x = synth.uniform01(2002, (k["n"], k["d"]))
    pq = ra.train_gaussian_opq(k["n_subquantizers"], k["n_bits"], k["n_iterations"], 1, x, rng=np.random.default_rng(3))
    rec = pq.reconstruct_batch(pq.quantize_batch(x))
    loss = np.sqrt(((x - rec) ** 2).sum(axis=1)).mean()
    assert loss < k["loss_bound"], loss


def test_concurrent_training_encode_and_lookup_callers(ra):
    """Training entry points next to serving calls from other host threads on one device: k-means
    (own aux stream), the OPQ step (shared per-device workspaces), encode, lookup -- every result
    still equals the oracle's."""
    import threading
    import torch
    n, M, K, dsub = 6000, 4, 32, 8
    d = M * dsub
    q0, x = _km_inputs(n, M, K, dsub, 2100)
    P = synth.orthonormal(2101, d)
    xd = torch.from_numpy(x).cuda()
    pq = _pq(ra, q0)
    codes_ref = orc.quantize_batch(q0, x)
    codes_d = torch.from_numpy(codes_ref).cuda()
    rows = torch.arange(n - 1, -1, -1, device="cuda")
    want_km = orc.kmeans_iterations(q0, x, 4)
    want_opq = orc.opq_train_step(q0, P, x)
    want_rec = orc.reconstruct_batch(q0, codes_ref[::-1])
    out, errs = {}, []

    def guard(name, fn):
        def run():
            try:
                for _ in range(3):
                    out[name] = fn()
            except Exception as e:     # noqa: BLE001 - surfaced below
                errs.append((name, e))
        return run
    jobs = [guard("km", lambda: ra.kmeans_iterations(q0, xd, 4)),
            guard("opq", lambda: ra.opq_train_step(q0, P, xd)),
            guard("opq2", lambda: ra.opq_train_step(q0, P, xd)),
            guard("enc", lambda: pq.quantize_batch(x)),
            guard("look", lambda: pq.reconstruct_rows_device(codes_d, rows, check=True).cpu().numpy())]
    th = [threading.Thread(target=j) for j in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert out["km"][0].tobytes() == want_km[0].tobytes() and out["km"][1].tobytes() == want_km[1].tobytes()
    for k in ("opq", "opq2"):
        assert out[k][0].tobytes() == want_opq[0].tobytes() and out[k][1].tobytes() == want_opq[1].tobytes()
    assert out["enc"].tobytes() == codes_ref.tobytes()
    assert out["look"].tobytes() == want_rec.tobytes()
