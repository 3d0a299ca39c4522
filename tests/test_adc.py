"""asymmetric distance computation over codes (lookup tables + table-sum scan).
Not a function of reductive: it is DEFINED from the reference's vector-to-matrix distance
(linalg.rs:118-148, the quantity `kmeans::cluster_assignment` minimises, kmeans.rs:111-126), so its
anchor to the reference is the identity  argmin_j tables[m][j] == quantize_vector(query)[m]  and the
reference's own KAT codebook (pq.rs:378-407).  CPU: the oracle's definition.  GPU: HIP path vs oracle,
bit for bit."""
import numpy as np
import pytest

import synth
from oracle import pq_oracle as orc


# ---- CPU: the oracle's definition ---------------------------------------------------------------
def test_tables_minimise_to_the_reference_kat_quantizations(kats):
    k = kats["pq_predefined_codebook"]
    q = np.array(k["quantizers"], np.float32)
    for v, want in zip(k["vectors"], k["quantizations"]):
        t = orc.adc_tables(q, np.array(v, np.float32))
        assert t.shape == q.shape[:2]
        assert [orc.first_min(t[m]) for m in range(q.shape[0])] == want
        # the table entry is the plain squared distance (the KAT holds 0.2 / 0.5: compare to rounding)
        for m in range(q.shape[0]):
            sub = np.array(v, np.float64)[m * q.shape[2]:(m + 1) * q.shape[2]]
            assert np.abs(t[m] - ((q[m].astype(np.float64) - sub) ** 2).sum(1)).max() <= 1e-6
    # scan over the KAT's own codes: distance of vector i to the reconstruction of codes i
    codes = np.array(k["quantizations"], np.uint8)
    rec = np.array(k["reconstructions"], np.float64)
    for i, v in enumerate(k["vectors"]):
        dist = orc.adc_scan(orc.adc_tables(q, np.array(v, np.float32)), codes)
        assert np.abs(dist - ((rec - np.array(v, np.float64)) ** 2).sum(1)).max() <= 1e-6


@pytest.mark.parametrize("M,K,dsub,opq", [(15, 256, 20, False), (15, 256, 20, True), (10, 128, 2, False), (3, 7, 5, True)])
def test_oracle_tables_and_scan_properties(M, K, dsub, opq):
    d = M * dsub
    q = synth.normalish(9300 + d, (M, K, dsub))
    P = synth.orthonormal(9301 + d, d) if opq else None
    ys = synth.normalish(9302 + d, (6, d))
    t = orc.adc_tables(q, ys, projection=P)
    assert t.shape == (6, M, K)
    for i in range(6):
        assert [orc.first_min(t[i, m]) for m in range(M)] == orc.quantize_vector(q, ys[i], projection=P).tolist()
    codes = synth.codes_u8(9303 + d, (500, M), K)
    dist = orc.adc_scan(t, codes)
    # against the reconstructed vectors in float64: ADC == |y' - reconstruct(codes)|^2 up to rounding
    yr = orc.rotate(ys, P) if opq else ys
    rec = orc.reconstruct_batch(q, codes).astype(np.float64)
    ref = ((rec[None, :, :] - yr[:, None, :].astype(np.float64)) ** 2).sum(-1)
    assert np.abs(dist - ref).max() <= 1e-4 * ref.max()
    # sequential f32 order over m, literally
    s = np.zeros(500, np.float32)
    for m in range(M):
        s = s + t[2, m, codes[:, m]]
    assert s.tobytes() == dist[2].tobytes()
    with pytest.raises(IndexError):
        bad = codes.astype(np.uint16)
        bad[3, 1] = K
        orc.adc_scan(t[0], bad)


# ---- GPU ----------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ra():
    import os
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    reductive_amd.lib()
    return reductive_amd


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,dsub,opq,n", [(15, 256, 20, False, 100003), (15, 256, 20, True, 5000), (48, 256, 16, False, 20011),
                                            (1, 256, 8, False, 4099), (2, 16, 4, False, 1000), (3, 7, 5, True, 777),
                                            (100, 64, 2, False, 3000), (101, 32, 2, False, 3000), (16, 16, 8, False, 100),
                                            (7, 200, 3, False, 1), (96, 256, 8, False, 50000)])
def test_gpu_tables_and_scan_match_oracle(ra, M, K, dsub, opq, n):
    import torch
    d = M * dsub
    q = synth.normalish(9400 + d + K, (M, K, dsub))
    P = synth.orthonormal(9401 + d, d) if opq else None
    pq = ra.Pq(P, q)
    ys = synth.normalish(9402 + d, (3, d))
    want_t = orc.adc_tables(q, ys, projection=P)
    t = pq.adc_tables_device(torch.from_numpy(ys).cuda())
    assert t.cpu().numpy().tobytes() == want_t.tobytes()
    t1 = pq.adc_tables_device(torch.from_numpy(ys[1]).cuda())
    assert t1.cpu().numpy().tobytes() == want_t[1].tobytes()
    codes = synth.codes_u8(9403 + d, (n, M), K)
    cd = torch.from_numpy(codes).cuda()
    want = orc.adc_scan(want_t, codes)
    got = pq.adc_scan_device(cd, t, check=True)
    assert got.cpu().numpy().tobytes() == want.tobytes()
    got1 = pq.adc_scan_device(cd, t1, check=True)
    assert got1.cpu().numpy().tobytes() == want[1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,dsub,n,nq", [(15, 256, 20, 70001, 8), (15, 256, 20, 5003, 13), (15, 256, 20, 1025, 4), (48, 256, 16, 9001, 9),
                                           (3, 7, 5, 777, 21), (100, 64, 2, 3000, 8), (16, 16, 8, 100, 5), (96, 256, 8, 4100, 6)])
def test_gpu_multi_query_scan_matches_oracle(ra, M, K, dsub, n, nq):
    """several queries per pass over the resident codes (8 or 4 tables interleaved in LDS, the rest
    one by one) -- every (query, row) distance equals the oracle's sequential f32 sum, whatever mix of passes serves nq
    queries; strided / unaligned code rows and the range flag behave as in the single-query form."""
    import torch
    d = M * dsub
    q = synth.normalish(9800 + d + K, (M, K, dsub))
    pq = ra.Pq(None, q)
    ys = synth.normalish(9801 + d + nq, (nq, d))
    t = pq.adc_tables_device(torch.from_numpy(ys).cuda())
    want_t = orc.adc_tables(q, ys)
    assert t.cpu().numpy().tobytes() == want_t.tobytes()
    wide = synth.codes_u8(9802 + d, (n + 3, M + 5), K)
    wd = torch.from_numpy(wide).cuda()
    for r0, c0 in ((0, 0), (1, 3), (3, 5)):                  # aligned and unaligned first rows, row stride M + 5
        view = wd[r0:r0 + n, c0:c0 + M]
        got = pq.adc_scan_device(view, t, check=True)
        assert got.shape == (nq, n)
        assert got.cpu().numpy().tobytes() == orc.adc_scan(want_t, np.ascontiguousarray(wide[r0:r0 + n, c0:c0 + M])).tobytes()
    if K < 256:
        bad = wd[:n, :M].contiguous()
        bad[n - 1, M - 1] = K
        with pytest.raises(ra.PanicError, match="index out of bounds"):
            pq.adc_scan_device(bad, t, check=True)


@pytest.mark.gpu
def test_gpu_scan_unaligned_strided_and_range(ra):
    import torch
    M, K, dsub = 15, 100, 4
    q = synth.normalish(9500, (M, K, dsub))
    pq = ra.Pq(None, q)
    y = synth.normalish(9501, (M * dsub,))
    t = pq.adc_tables_device(torch.from_numpy(y).cuda())
    want_t = orc.adc_tables(q, y)
    assert t.cpu().numpy().tobytes() == want_t.tobytes()
    wide = synth.codes_u8(9502, (30001, M + 6), K)          # row stride 21, codes in columns 5..19
    wd = torch.from_numpy(wide).cuda()
    for r0 in (0, 1, 2, 3):                                  # every byte alignment of the first row
        view = wd[r0:, 5:5 + M]
        assert view.stride(0) == M + 6 and view.data_ptr() % 4 == (wd.data_ptr() + r0 * (M + 6) + 5) % 4
        got = pq.adc_scan_device(view, t, check=True)
        assert got.cpu().numpy().tobytes() == orc.adc_scan(want_t, wide[r0:, 5:5 + M]).tobytes()
    tight = np.ascontiguousarray(wide[:, 5:5 + M])
    td = torch.from_numpy(tight).cuda()
    for r0 in (0, 1, 2, 3, 29990):
        got = pq.adc_scan_device(td[r0:], t, check=True)
        assert got.cpu().numpy().tobytes() == orc.adc_scan(want_t, tight[r0:]).tobytes()
    bad = td.clone()
    bad[17, 3] = K
    with pytest.raises(ra.PanicError, match="index out of bounds"):
        pq.adc_scan_device(bad, t, check=True)
    pq.adc_scan_device(td, t, check=True)                    # flag consumed
    with pytest.raises(ra.PanicError, match="Quantization length"):
        pq.adc_scan_device(td[:, :M - 1], t)


@pytest.mark.gpu
def test_gpu_scan_wide_index_type_and_big_tables(ra):
    import torch
    M, K, dsub = 6, 700, 4                                   # 32-bit codes, generic kernel
    q = synth.normalish(9600, (M, K, dsub))
    pq = ra.Pq(None, q)
    y = synth.normalish(9601, (2, M * dsub))
    want_t = orc.adc_tables(q, y)
    t = pq.adc_tables_device(torch.from_numpy(y).cuda())
    assert t.cpu().numpy().tobytes() == want_t.tobytes()
    x = synth.normalish(9602, (5000, M * dsub))
    codes = orc.quantize_batch(q, x, dtype=np.uint32)
    got = pq.adc_scan_device(torch.from_numpy(codes.astype(np.int32)).cuda(), t, check=True)
    assert got.cpu().numpy().tobytes() == orc.adc_scan(want_t, codes).tobytes()
    # the nearest code row under ADC is the row's own quantization: dist(x_i, codes_i) <= dist(x_i, codes_j)
    ti = pq.adc_tables_device(torch.from_numpy(x[:8]).cuda())
    dd = pq.adc_scan_device(torch.from_numpy(codes.astype(np.int32)).cuda(), ti).cpu().numpy()
    assert (dd.argmin(1) == np.arange(8)).all() or all(dd[i, i] == dd[i].min() for i in range(8))
    # 32-bit codes with the table in LDS (k_adc_scan_wide: M K <= 40,960 entries) and beyond it (generic kernel); code rows
    # inside a wider matrix (row stride > M, 4-byte aligned only); a code >= K raises the range flag
    for (M2, K2, n2) in [(15, 1024, 70001), (15, 2048, 5003), (7, 300, 2049), (48, 1024, 3000)]:
        g = torch.Generator(device="cuda").manual_seed(9610 + K2)
        tab = torch.rand((3, M2, K2), device="cuda", generator=g)
        wide = torch.randint(0, K2, (n2, M2 + 3), device="cuda", dtype=torch.int32, generator=g)
        view = wide[:, 1:1 + M2]
        pq2 = ra.Pq(None, synth.normalish(9611 + K2, (M2, K2, 2)))
        got2 = pq2.adc_scan_device(view, tab, check=True).cpu().numpy()
        want2 = orc.adc_scan(tab.cpu().numpy(), view.cpu().numpy().astype(np.uint32))
        assert got2.tobytes() == want2.tobytes(), (M2, K2)
        wide[n2 // 2, 2] = K2
        with pytest.raises(ra.PanicError):
            pq2.adc_scan_device(view, tab, check=True)


@pytest.mark.gpu
def test_gpu_scan_at_scale_head_and_tail(ra):
    """30 M code rows (450 MB: beyond the Infinity Cache): head and tail rows equal the oracle, and the
    whole vector obeys the size-independent bound  min_m-sum <= dist <= max_m-sum."""
    import torch
    M, K, dsub = 15, 256, 20
    q = synth.normalish(9700, (M, K, dsub))
    pq = ra.Pq(None, q)
    y = synth.normalish(9701, (M * dsub,))
    t = pq.adc_tables_device(torch.from_numpy(y).cuda())
    want_t = orc.adc_tables(q, y)
    n = 30_000_000
    g = torch.Generator(device="cuda").manual_seed(9702)
    codes = torch.randint(0, K, (n, M), device="cuda", dtype=torch.uint8, generator=g)
    dist = pq.adc_scan_device(codes, t, check=True)
    for s0 in (0, n // 2, n - 200_000):
        assert dist[s0:s0 + 200_000].cpu().numpy().tobytes() == orc.adc_scan(want_t, codes[s0:s0 + 200_000].cpu().numpy()).tobytes()
    lo, hi = float(want_t.min(1).sum()), float(want_t.max(1).sum())
    assert float(dist.min()) >= lo * (1 - 1e-5) and float(dist.max()) <= hi * (1 + 1e-5)
