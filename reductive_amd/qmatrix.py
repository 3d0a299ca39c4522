"""A quantized embedding matrix kept resident in HBM: SURVEY.md 8f rank 3 (file / wire format -> direct
device upload) feeding rank 2 (lookup) and rank 4 (ADC scan).

The consumer of reductive's `Pq` -- finalfusion's quantized embedding storage -- is NOT in
/root/reference; its chunk layout is restated here from memory of finalfusion's public format
description, so this reader is UNPINNED by construction: no reference-held fixture exists for it and
none can be produced in this image (no Rust toolchain).  What is pinned is the in-tree surface it feeds
(`Pq::new` pq.rs:38-61, `projection()` :108-110, `subquantizers()` :191-193, `n_quantizer_centroids()`
:103-105) and the round trip through our own writer.

Storage chunk, little endian:
    u32 chunk identifier (4 = QuantizedArray; 3 is BucketSubwordVocab)     u64 chunk length in bytes (of what follows)
    u32 projection (0/1)   u32 norms (0/1)   u32 quantized_len M   u32 reconstructed_len d
    u32 n_centroids K      u64 n_embeddings N
    u32 quantized type id (1 = u8)   u32 reconstructed type id (10 = f32)
    zero padding up to a multiple of 4 bytes of the ABSOLUTE position in the file (the chunk follows the magic,
    the header chunk and usually a vocabulary chunk, so its start is not aligned in general: writer and reader take the
    position from f.tell(), or from `stream_offset` for streams that cannot tell)
    [d x d] f32 projection (if flagged)    [M x K x d/M] f32 quantizers
    [N] f32 norms (if flagged)             [N x M] u8 quantized embeddings
"""
import io
import struct

import numpy as np

from .pq import Pq, PanicError

CHUNK_QUANTIZED_ARRAY = 4
TYPE_U8, TYPE_F32 = 1, 10


class FormatError(ValueError):
    pass


def _position(f, stream_offset):
    """absolute position of the next byte of `f` in its file (f.tell() when the stream supports it)"""
    if stream_offset is not None:
        return stream_offset
    try:
        return f.tell()
    except (OSError, AttributeError, io.UnsupportedOperation):
        return 0


def write_chunk(f, pq, codes, norms=None, stream_offset=None):
    """Serialise (pq, codes [N, M] u8, norms [N] f32 or None) as one storage chunk at the stream's current position."""
    stream_offset = _position(f, stream_offset)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    M, K, dsub = pq.subquantizers().shape
    if codes.ndim != 2 or codes.shape[1] != M:
        raise PanicError("Quantization length does not match number of subquantizers")
    if norms is not None:
        norms = np.ascontiguousarray(norms, dtype=np.float32)
        if norms.shape != (codes.shape[0],):
            raise FormatError("one norm per embedding expected")
    P = pq.projection()
    head = struct.pack("<IIIIIQII", int(P is not None), int(norms is not None), M, M * dsub, K, codes.shape[0],
                       TYPE_U8, TYPE_F32)
    pad = (-(stream_offset + 12 + len(head))) % 4
    body = [head, b"\0" * pad]
    if P is not None:
        body.append(np.ascontiguousarray(P, dtype="<f4").tobytes())
    body.append(np.ascontiguousarray(pq.subquantizers(), dtype="<f4").tobytes())
    if norms is not None:
        body.append(norms.astype("<f4").tobytes())
    body.append(codes.tobytes())
    payload = b"".join(body)
    f.write(struct.pack("<IQ", CHUNK_QUANTIZED_ARRAY, len(payload)))
    f.write(payload)


def read_chunk(f, stream_offset=None, ctx=None):
    """Parse the storage chunk that starts at the stream's current position -> (Pq, codes [N, M] u8, norms [N] f32 or
    None); host arrays."""
    stream_offset = _position(f, stream_offset)
    def take(n):
        b = f.read(n)
        if len(b) != n:
            raise FormatError("truncated quantized-array chunk")
        return b
    ident, length = struct.unpack("<IQ", take(12))
    if ident != CHUNK_QUANTIZED_ARRAY:
        raise FormatError("not a quantized-array chunk (identifier %d)" % ident)
    proj, has_norms, M, d, K, N, qt, rt = struct.unpack("<IIIIIQII", take(36))
    if qt != TYPE_U8 or rt != TYPE_F32:
        raise FormatError("unsupported element types (%d, %d): u8 codes and f32 reconstructions only" % (qt, rt))
    if M == 0 or d == 0 or K == 0 or d % M != 0 or K > 256:
        raise FormatError("inconsistent quantizer shape M=%d d=%d K=%d" % (M, d, K))
    pad = (-(stream_offset + 12 + 36)) % 4
    take(pad)
    need = (d * d * 4 if proj else 0) + M * K * (d // M) * 4 + (N * 4 if has_norms else 0) + N * M
    if length != 36 + pad + need:
        raise FormatError("chunk length %d does not match its header (%d)" % (length, 36 + pad + need))
    P = np.frombuffer(take(d * d * 4), "<f4").reshape(d, d).astype(np.float32) if proj else None
    q = np.frombuffer(take(M * K * (d // M) * 4), "<f4").reshape(M, K, d // M).astype(np.float32)
    norms = np.frombuffer(take(N * 4), "<f4").astype(np.float32) if has_norms else None
    codes = np.frombuffer(take(N * M), np.uint8).reshape(N, M).copy()
    return Pq(P, q, ctx=ctx), codes, norms


class QuantizedMatrix:
    """Codes (+ norms) resident in HBM next to the device codebook: the lookup and scan consumer."""

    def __init__(self, pq, codes, norms=None, device="cuda:0"):
        import torch
        self.pq = pq
        self.codes = torch.as_tensor(np.ascontiguousarray(codes, dtype=np.uint8)).to(device)
        self.norms = None if norms is None else torch.as_tensor(np.ascontiguousarray(norms, dtype=np.float32)).to(device)
        if self.codes.dim() != 2 or self.codes.shape[1] != pq.quantized_len():
            raise PanicError("Quantization length does not match number of subquantizers")

    @classmethod
    def load(cls, path_or_file, device="cuda:0", ctx=None):
        """file -> device: header on the host, the three payload arrays straight into device tensors."""
        f = open(path_or_file, "rb") if isinstance(path_or_file, str) else path_or_file
        try:
            pq, codes, norms = read_chunk(f, ctx=ctx)
        finally:
            if isinstance(path_or_file, str):
                f.close()
        return cls(pq, codes, norms, device=device)

    def __len__(self):
        return self.codes.shape[0]

    def embeddings(self, rows, out=None):
        """`reconstruct_batch(codes.select(Axis(0), rows)) * norms.select(rows)` in one pass over HBM."""
        import torch
        rows = torch.as_tensor(rows, dtype=torch.int64, device=self.codes.device)
        return self.pq.reconstruct_rows_device(self.codes, rows, scales=self.norms, out=out)

    def distances(self, queries):
        """asymmetric squared distances of the query vector(s) to every (un-normalised) code row."""
        return self.pq.adc_scan_device(self.codes, self.pq.adc_tables_device(queries))


def dumps(pq, codes, norms=None):
    b = io.BytesIO()
    write_chunk(b, pq, codes, norms)
    return b.getvalue()
