"""k_reconstruct: workgroups per CU (diagnostic build, PQHIP_DEBUG_REC_WGS) at several matrix sizes; run once per value:
PQHIP_LIB=.../libpqhip_diag.so PQHIP_DEBUG_REC_WGS=<w> python tools/rec_wgs_sweep.py"""
import os
import numpy as np, torch
import reductive_amd as ra

M, K, dsub = 15, 256, 20
d = M * dsub
rng = np.random.default_rng(1)
pq = ra.Pq(None, rng.standard_normal((M, K, dsub), dtype=np.float32))
for n in (10_000_000, 40_000_000, 100_000_000):
    codes = torch.randint(0, K, (n, M), device="cuda", dtype=torch.uint8)
    out = torch.empty((n, d), device="cuda", dtype=torch.float32)
    for _ in range(2):
        pq.reconstruct_batch_device(codes, out=out, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pq.reconstruct_batch_device(codes, out=out, check=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("wgs/CU %s rows %d  %.3f ms  %.3f of HBM" % (os.environ.get("PQHIP_DEBUG_REC_WGS", "auto"), n, ms, n * (4 * d + M) / ms * 1e3 / 8e12), flush=True)
    del codes, out
    torch.cuda.empty_cache()
