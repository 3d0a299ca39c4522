import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
from oracle import pq_oracle as orc
import reductive_amd as ra
for d, M in [(515, 5), (1300, 65), (103, 1), (206, 2), (330, 33), (66, 2), (70, 2)]:
    n, K = 203, 16
    dsub = d // M
    q = synth.normalish(91 + d, (M, K, dsub)); x = synth.normalish(92 + d, (n, d)); P = synth.orthonormal(93 + d, d)
    codes = orc.quantize_batch(q, x, projection=P)
    pq0 = ra.Pq(None, q)
    r0 = pq0.reconstruct_batch(codes); w0 = orc.reconstruct_batch(q, codes)
    bad0 = np.argwhere(r0 != w0)
    pq = ra.Pq(P, q)
    r1 = pq.reconstruct_batch(codes); w1 = orc.reconstruct_batch(q, codes, projection=P)
    bad1 = np.argwhere(r1 != w1)
    print(d, M, dsub, "PQ gather mismatches", len(bad0), bad0[:5].tolist(), "OPQ mismatches", len(bad1), bad1[:5].tolist(),
          "enc ok", (pq.quantize_batch(x) == codes).all(), flush=True)
