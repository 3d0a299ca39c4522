"""Regenerates tests/golden/cases.npz from the CPU oracle (oracle/pq_oracle.c).

The reference is Rust and cannot run in this image, so these vectors are produced by OUR
restatement of its algorithm (CANON-F32), which is itself pinned to the reference's KATs in
tests/golden/reference_kats.json.  Inputs are regenerated from integer-hash seeds
(tests/synth.py) and verified by sha256; only the expected OUTPUTS are stored.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402
from oracle import pq_oracle as orc  # noqa: E402

# name: (n, M, K, dsub, dist, opq)
CASES = {
    "d300_m15_k256": (1000, 15, 256, 20, "normal", False),
    "d768_m48_k256": (200, 48, 256, 16, "normal", False),
    "d20_m10_k128_uniform": (256, 10, 128, 2, "uniform", False),
    "d128_m16_k16": (100, 16, 16, 8, "normal", False),
    "d6_m2_k2": (64, 2, 2, 3, "normal", False),
    "d35_m5_k40_odd": (300, 5, 40, 7, "normal", False),
    "opq_d300_m15_k256": (500, 15, 256, 20, "normal", True),
    "opq_d64_m8_k64": (300, 8, 64, 8, "normal", True),
}


def make_inputs(name):
    n, M, K, dsub, dist, opq = CASES[name]
    seed = int.from_bytes(name.encode(), "little") % (1 << 31)
    gen = synth.normalish if dist == "normal" else synth.uniform01
    q = gen(seed + 1, (M, K, dsub))
    x = gen(seed + 2, (n, M * dsub))
    return q, x, opq, seed


def main():
    out = {}
    meta = {}
    for name in CASES:
        q, x, opq, seed = make_inputs(name)
        P = synth.orthonormal(seed + 3, q.shape[0] * q.shape[2]) if opq else None
        codes = orc.quantize_batch(q, x, projection=P)
        rec = orc.reconstruct_batch(q, codes, projection=P)
        out[name + "/codes"] = codes
        if P is not None:
            out[name + "/projection"] = P       # QR goes through LAPACK: store, do not regenerate
        meta[name] = {"sha_q": synth.sha(q), "sha_x": synth.sha(x), "sha_rec": synth.sha(rec),
                      "sha_codes": synth.sha(codes)}
    np.savez_compressed(os.path.join(HERE, "cases.npz"), **out)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
