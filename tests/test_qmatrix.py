"""quantized-matrix storage chunk -> device.  The chunk layout is restated from
memory (the consumer is out of tree): parity UNPINNED by construction -- these tests pin the reader
against our own writer, the header arithmetic, and the in-tree surface (`Pq::new` panics)."""
import io
import struct

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


@pytest.mark.parametrize("opq,norms", [(False, False), (True, False), (False, True), (True, True)])
def test_chunk_round_trip_and_header(ra, opq, norms):
    from reductive_amd import qmatrix
    M, K, dsub, N = 5, 37, 4, 1001
    d = M * dsub
    q = synth.normalish(9800, (M, K, dsub))
    P = synth.orthonormal(9801, d) if opq else None
    codes = synth.codes_u8(9802, (N, M), K)
    nr = synth.uniform01(9803, (N,)) + np.float32(0.5) if norms else None
    blob = qmatrix.dumps(ra.Pq(P, q), codes, nr)
    ident, length = struct.unpack("<IQ", blob[:12])
    assert ident == 4 and length == len(blob) - 12     # 4 = QuantizedArray in finalfusion's chunk identifiers
    assert struct.unpack("<IIIIIQII", blob[12:48]) == (int(opq), int(norms), M, d, K, N, 1, 10)
    assert len(blob) == 48 + (d * d * 4 if opq else 0) + M * K * dsub * 4 + (N * 4 if norms else 0) + N * M
    pq, c2, n2 = qmatrix.read_chunk(io.BytesIO(blob))
    assert pq == ra.Pq(P, q) and c2.tobytes() == codes.tobytes()
    assert (n2 is None) == (nr is None) and (nr is None or n2.tobytes() == nr.tobytes())
    # unaligned stream position: the writer pads, the reader skips the same bytes
    b = io.BytesIO()
    b.write(b"xyz")
    qmatrix.write_chunk(b, ra.Pq(P, q), codes, nr, stream_offset=3)
    b.seek(3)
    pq3, c3, _ = qmatrix.read_chunk(b, stream_offset=3)
    assert pq3 == pq and c3.tobytes() == codes.tobytes()
    # a chunk in the middle of a file (behind magic / header / vocabulary bytes): the padding follows the ABSOLUTE
    # position, which writer and reader take from the stream itself
    for lead in (5, 6, 7, 8):
        b = io.BytesIO()
        b.write(b"v" * lead)
        qmatrix.write_chunk(b, ra.Pq(P, q), codes, nr)
        pad = (-(lead + 48)) % 4
        assert len(b.getvalue()) == lead + len(blob) + pad
        b.seek(lead)
        pq4, c4, n4 = qmatrix.read_chunk(b)
        assert pq4 == pq and c4.tobytes() == codes.tobytes() and (nr is None or n4.tobytes() == nr.tobytes())


def test_malformed_chunks_are_refused(ra):
    from reductive_amd import qmatrix
    q = synth.normalish(9810, (2, 4, 3))
    blob = bytearray(qmatrix.dumps(ra.Pq(None, q), synth.codes_u8(9811, (10, 2), 4)))
    with pytest.raises(qmatrix.FormatError, match="truncated"):
        qmatrix.read_chunk(io.BytesIO(bytes(blob[:-1])))
    bad = bytearray(blob); bad[0] = 2
    with pytest.raises(qmatrix.FormatError, match="identifier"):
        qmatrix.read_chunk(io.BytesIO(bytes(bad)))
    bad = bytearray(blob); bad[40] = 2          # quantized type id
    with pytest.raises(qmatrix.FormatError, match="element types"):
        qmatrix.read_chunk(io.BytesIO(bytes(bad)))
    bad = bytearray(blob); bad[4] ^= 1          # chunk length
    with pytest.raises(qmatrix.FormatError, match="length"):
        qmatrix.read_chunk(io.BytesIO(bytes(bad)))
    with pytest.raises(ra.PanicError, match="Quantization length"):
        qmatrix.dumps(ra.Pq(None, q), np.zeros((3, 5), np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("opq", [False, True])
def test_load_to_device_lookup_and_scan(ra, tmp_path, opq):
    import torch
    from reductive_amd import qmatrix
    M, K, dsub, N = 15, 256, 20, 200_000
    d = M * dsub
    q = synth.normalish(9820, (M, K, dsub))
    P = synth.orthonormal(9821, d) if opq else None
    codes = synth.codes_u8(9822, (N, M), K)
    norms = synth.uniform01(9823, (N,)) + np.float32(0.5)
    path = str(tmp_path / "emb.qa")
    with open(path, "wb") as f:
        qmatrix.write_chunk(f, ra.Pq(P, q), codes, norms)
    qm = qmatrix.QuantizedMatrix.load(path)
    assert len(qm) == N and qm.codes.is_cuda
    rows = np.array([0, N - 1, 17, 17, 123456, 5], np.int64)
    got = qm.embeddings(rows).cpu().numpy()
    want = orc.reconstruct_batch(q, codes[rows], projection=P) * norms[rows][:, None]
    if opq:
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    else:
        assert got.tobytes() == want.astype(np.float32).tobytes()
    y = synth.normalish(9824, (d,))
    dist = qm.distances(torch.from_numpy(y).cuda()).cpu().numpy()
    assert dist.tobytes() == orc.adc_scan(orc.adc_tables(q, y, projection=P), codes).tobytes()
