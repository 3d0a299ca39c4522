#!/usr/bin/env python3
"""Latency of the host-resident entry point for small batches (numpy in, numpy out; d = 300, M = 15, K = 256), next to
the CPU oracle on one thread -- the number behind the shim's GPU-dispatch threshold.
usage: python tools/small_batch_latency.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth, reductive_amd
from oracle import pq_oracle as orc          # the checker / CPU stand-in, not the product

M, K, dsub = 15, 256, 20
q = synth.normalish(43, (M, K, dsub))
P = synth.orthonormal(44, M * dsub)
for name, proj in (("pq", None), ("opq", P)):
    pq = reductive_amd.Pq(proj, q)
    for n in (1, 32, 256, 1024, 4096, 16384, 65536):
        x = synth.normalish(100 + n, (n, M * dsub))
        pq.quantize_batch(x)
        reps = 200 if n <= 4096 else 30
        t = time.perf_counter()
        for _ in range(reps):
            codes = pq.quantize_batch(x)
        gpu_us = (time.perf_counter() - t) / reps * 1e6
        t = time.perf_counter()
        want = orc.quantize_batch(q, x, projection=proj, n_threads=1)
        cpu_us = (time.perf_counter() - t) * 1e6
        rec = pq.reconstruct_batch(codes)
        t = time.perf_counter()
        for _ in range(reps):
            rec = pq.reconstruct_batch(codes)
        rec_us = (time.perf_counter() - t) / reps * 1e6
        print(json.dumps({"codebook": name, "rows": n, "quantize_batch_us": round(gpu_us, 1), "oracle_1_thread_us": round(cpu_us, 1),
                          "reconstruct_batch_us": round(rec_us, 1), "identical": bool(codes.tobytes() == want.tobytes())}), flush=True)
    pq.close()
