"""GPU tests of the drop-in boundary's concurrency contract (include/pqhip.h: "all entry points are
re-entrant and may be called concurrently from many threads"; SURVEY.md 8b Ownership / Threading;
`Pq<f32>` is Send + Sync, src/pq/pq.rs:28-32): the leased scratch pool of the OPQ / K > 256 device
paths, the per-stream range flags, and the caller's current device being left alone."""
import ctypes
import threading

import numpy as np
import pytest

import synth
from oracle import pq_oracle as orc

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


@pytest.fixture(scope="module")
def ra():
    import os
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    reductive_amd.lib()
    return reductive_amd


def _run_threads(jobs):
    errs = []

    def wrap(fn):
        def run():
            try:
                fn()
            except Exception as e:    # noqa: BLE001 - surfaced below
                errs.append(e)
        return run
    th = [threading.Thread(target=wrap(j)) for j in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_four_threads_opq_device_entry_points_growing_batches(ra):
    """4 host threads x own stream, one OPQ codebook, batches that grow from call to call (so the
    scratch a call needs is larger than what earlier calls left in the pool): codes and
    reconstructions equal the oracle's in every call.  Before the pool, a thread could launch on a
    scratch pointer another thread had just freed, and two streams could wait on the same stale event."""
    import torch
    M, K, dsub = 8, 64, 8
    d = M * dsub
    q = synth.normalish(5101, (M, K, dsub))
    P = synth.orthonormal(5102, d)
    pq = ra.Pq(P, q)
    sizes = [1500, 9000, 40000, 150000]
    xs = {n: synth.normalish(5200 + n, (n, d)) for n in sizes}
    want = {n: orc.quantize_batch(q, xs[n], projection=P, n_threads=8) for n in sizes}
    want_rec = {n: orc.reconstruct_batch(q, want[n], projection=P) for n in sizes}
    got = {}

    def worker(t):
        def run():
            st = torch.cuda.Stream()
            order = sizes[t % 4:] + sizes[:t % 4]           # every thread grows at a different time
            with torch.cuda.stream(st):
                for rep in range(3):
                    for n in order:
                        xd = torch.from_numpy(xs[n]).cuda()
                        codes = pq.quantize_batch_device(xd)
                        rec = pq.reconstruct_batch_device(codes, check=True)
                        st.synchronize()
                        got[(t, rep, n)] = (codes.cpu().numpy(), rec.cpu().numpy())
        return run
    _run_threads([worker(t) for t in range(4)])
    assert len(got) == 4 * 3 * 4
    for (t, rep, n), (codes, rec) in got.items():
        assert codes.tobytes() == want[n].tobytes(), (t, rep, n)
        assert np.abs(rec - want_rec[n]).max() <= REL_TOL * np.abs(want_rec[n]).max(), (t, rep, n)


def test_four_threads_grouped_codebook_growing_batches(ra):
    """Same for a K > 256 codebook (64-bit partial-minimum keys in a leased buffer, 32-bit codes)."""
    import torch
    M, K, dsub = 3, 700, 8
    d = M * dsub
    q = synth.normalish(5301, (M, K, dsub))
    pq = ra.Pq(None, q)
    sizes = [700, 6000, 33000, 90000]
    xs = {n: synth.normalish(5400 + n, (n, d)) for n in sizes}
    want = {n: orc.quantize_batch(q, xs[n], dtype=np.uint32, n_threads=8) for n in sizes}
    got = {}

    def worker(t):
        def run():
            st = torch.cuda.Stream()
            order = sizes[t % 4:] + sizes[:t % 4]
            with torch.cuda.stream(st):
                for rep in range(3):
                    for n in order:
                        codes = pq.quantize_batch_device(torch.from_numpy(xs[n]).cuda())
                        st.synchronize()
                        got[(t, rep, n)] = codes.cpu().numpy()
        return run
    _run_threads([worker(t) for t in range(4)])
    assert pq.last_encode_kernel() == "k_encode_mfma_lds3<grouped>"
    for (t, rep, n), codes in got.items():
        assert codes.astype(np.uint32).tobytes() == want[n].tobytes(), (t, rep, n)


def test_host_and_device_entry_points_mixed_on_one_opq_codebook(ra):
    """A host-buffer call (library streams) and device calls (caller streams) on the same OPQ codebook
    at the same time."""
    import torch
    M, K, dsub = 6, 32, 10
    d = M * dsub
    q = synth.normalish(5501, (M, K, dsub))
    P = synth.orthonormal(5502, d)
    pq = ra.Pq(P, q)
    xh = synth.normalish(5503, (120000, d))
    xdev = synth.normalish(5504, (70000, d))
    want_h = orc.quantize_batch(q, xh, projection=P, n_threads=8)
    want_d = orc.quantize_batch(q, xdev, projection=P, n_threads=8)
    out = {}

    def host():
        for _ in range(2):
            out["h"] = pq.quantize_batch(xh)

    def dev():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(4):
                c = pq.quantize_batch_device(torch.from_numpy(xdev).cuda())
                st.synchronize()
                out["d"] = c.cpu().numpy()
    _run_threads([host, dev, dev])
    assert out["h"].tobytes() == want_h.tobytes()
    assert out["d"].tobytes() == want_d.tobytes()


def test_range_flag_is_per_stream(ra):
    """A code >= K seen on one stream must be reported to THAT stream's check and must not be consumed
    or cleared by a check on another stream (primitives.rs:146: the panic belongs to the call)."""
    import torch
    M, K, dsub = 4, 10, 4
    q = synth.normalish(5601, (M, K, dsub))
    pq = ra.Pq(None, q)
    good = torch.from_numpy(synth.codes_u8(5602, (5000, M), K)).cuda()
    bad = good.clone()
    bad[4321, 2] = K          # out of range
    s_good, s_bad = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s_bad):
        pq.reconstruct_batch_device(bad, check=False)
    s_bad.synchronize()
    with torch.cuda.stream(s_good):
        pq.reconstruct_batch_device(good, check=True)        # must neither raise nor clear the other stream's flag
    with torch.cuda.stream(s_bad):
        with pytest.raises(ra.PanicError, match="index out of bounds"):
            pq.reconstruct_batch_device(good, check=True)    # the earlier violation on this stream is still pending
        pq.reconstruct_batch_device(good, check=True)        # consumed exactly once


def test_entry_points_leave_the_callers_current_device_alone(ra):
    """hipSetDevice is per thread; every entry point restores the device the caller's thread had
    (torch.cuda.current_device() must not change under a torch caller)."""
    import torch
    L = ra.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0
    before = cur.value
    pq = ra.Pq(synth.orthonormal(5701, 12), synth.normalish(5702, (3, 16, 4)))
    x = torch.from_numpy(synth.normalish(5703, (1000, 12))).cuda()
    codes = pq.quantize_batch_device(x)
    pq.reconstruct_batch_device(codes, check=True)
    pq.quantize_batch(x.cpu().numpy())
    ra.kmeans_iterations(synth.normalish(5704, (3, 16, 4)), x, 2)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == before
    assert torch.cuda.current_device() == before
    assert L.pqhip_version() == 100


@pytest.mark.parametrize("n_slots,M,K,dsub,n", [(2, 48, 256, 16, 40_003), (8, 48, 256, 16, 70_001), (8, 15, 256, 20, 33_000),
                                               (5, 15, 256, 20, 300_007), (8, 4, 16, 8, 4_097)])
def test_library_sharder_over_many_device_slots(ra, n_slots, M, K, dsub, n):
    """SURVEY.md 8e through the LIBRARY's sharder (pqhip_ctx_create(devices) + host-buffer entry points:
    contiguous row shards, one host thread per slot, codes land in one host array) with up to 8 slots on the
    one GPU of the box -- the BASELINE configs[4] shape (d = 768, M = 48) included: every row equals the
    oracle's, wide and strided outputs too."""
    from reductive_amd.pq import _Ctx
    ctx = _Ctx(devices=[0] * n_slots)
    try:
        assert ctx.n_devices == n_slots
        d = M * dsub
        q = synth.normalish(5800 + M, (M, K, dsub))
        x = synth.normalish(5801 + n, (n, d))
        pq = ra.Pq(None, q, ctx=ctx)
        want = orc.quantize_batch(q, x, n_threads=8)
        assert pq.quantize_batch(x).tobytes() == want.tobytes()
        wide = np.zeros((n, M + 3), np.uint16)               # strided, wider index type
        pq.quantize_batch_into(x, wide[:, 1:1 + M])
        assert wide[:, 1:1 + M].tolist() == want.astype(np.uint16).tolist() and not wide[:, 0].any() and not wide[:, 1 + M:].any()
        rec = pq.reconstruct_batch(want)
        assert rec.tobytes() == orc.reconstruct_batch(q, want).tobytes()
        bad = want.copy()
        bad[n - 1, M - 1] = K if K < 256 else 0
        if K < 256:
            with pytest.raises(ra.PanicError, match="index out of bounds"):
                pq.reconstruct_batch(bad)                    # the violation sits in the LAST shard
        pq.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("M,K,dsub,opq", [(15, 256, 20, False), (6, 32, 10, True), (3, 700, 8, False), (2, 65536, 4, False)])
def test_device_entry_points_take_every_index_width(ra, M, K, dsub, opq):
    """traits.rs:77-88 is generic over the index type I: the DEVICE entry points take 1-, 2-, 4- and 8-byte code matrices
    like the host ones (round 3 refused 2 and 8: VERDICT r3 missing #6).  2- / 8-byte codes equal the u8 / u32 ones value
    for value, reconstruct from them gives the same rows, an index type too narrow for K is EINDEX_WIDTH, and a 2- or
    8-byte code >= K -- including values that would wrap to a valid index when narrowed -- raises the range flag."""
    import torch
    d, n = M * dsub, 5000
    q = synth.normalish(7100 + K, (M, K, dsub))
    P = synth.orthonormal(7101, d) if opq else None
    x = torch.from_numpy(synth.normalish(7102 + K, (n, d))).cuda()
    pq = ra.Pq(P, q)
    want = orc.quantize_batch(q, x.cpu().numpy(), projection=P, n_threads=8, dtype=np.uint32).astype(np.int64)
    for dt in (torch.uint8, torch.int16, torch.int32, torch.int64):
        bits = 8 * torch.empty((), dtype=dt).element_size()
        out = torch.zeros((n, M + 3), dtype=dt, device="cuda")[:, :M]          # a strided view of a wider matrix
        if K - 1 > (1 << bits) - 1:
            with pytest.raises(ra.PanicError, match="Cannot store centroids"):
                pq.quantize_batch_device(x, out=out)
            continue
        got = pq.quantize_batch_device(x, out=out)
        torch.cuda.synchronize()
        mask = (1 << bits) - 1 if bits < 64 else -1
        assert ((got.cpu().numpy().astype(np.int64) & mask) == want).all(), dt
        rec = pq.reconstruct_batch_device(got, check=True).cpu().numpy()
        ref = orc.reconstruct_batch(q, want.astype(np.uint32), projection=P)
        assert np.abs(rec - ref).max() <= REL_TOL * max(1e-30, np.abs(ref).max())
        if P is None:
            assert rec.tobytes() == ref.tobytes()
    # out-of-range codes in the wide types: K itself, and 2^32 + 1 (a valid index once truncated to 32 bits)
    for dt, bad in ((torch.int16, K if K < 32768 else None), (torch.int64, K), (torch.int64, (1 << 32) + 1)):
        if bad is None or (dt == torch.int16 and K > 32767):
            continue
        codes = torch.from_numpy(want[:64]).to(dt).cuda()
        codes[63, M - 1] = bad
        with pytest.raises(ra.PanicError, match="index out of bounds"):
            pq.reconstruct_batch_device(codes, check=True)
        codes[63, M - 1] = 0
        pq.reconstruct_batch_device(codes, check=True)                          # the flag was cleared by the check
