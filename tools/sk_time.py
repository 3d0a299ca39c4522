"""Time quantize_batch_device at the small-codebook bench shape (A/B of library builds via PQHIP_LIB).  prints ms per call."""
import sys, time
import numpy as np, torch
import reductive_amd as ra

d, M, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 16, 16)
n = 10_000_000
rng = np.random.default_rng(1)
cb = rng.standard_normal((M, K, d // M), dtype=np.float32)
pq = ra.Pq(None, cb)
x = torch.randn(n, d, device="cuda", dtype=torch.float32)
out = torch.empty(n, M, device="cuda", dtype=torch.uint8)
for _ in range(3):
    pq.quantize_batch_device(x, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pq.quantize_batch_device(x, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("%.4f ms  %.3e vec/s  %.3f of HBM" % (ms, n / ms * 1e3, n * (4 * d + M) / ms * 1e3 / 8e12))
