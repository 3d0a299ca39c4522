#!/usr/bin/env python3
"""Time `a.t().dot(&b)` (pqhip_at_dot_b_f32_dev: k_atb_rowblock + k_atb_fold) on resident matrices, exact and
float-tolerance mode.  usage: python tools/atb_time.py [rows] [d]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import reductive_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.empty((n, d), device="cuda").normal_(generator=g)
b = torch.empty((n, d), device="cuda").normal_(generator=g)
for exact in (1, 0):
    ra.set_option("cross_product_exact", exact)
    ra.at_dot_b(a, b)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        ra.launch_log(reset=True)
        t = time.perf_counter()
        c = ra.at_dot_b(a, b)
        ts.append(time.perf_counter() - t)
    ms = min(ts) * 1e3
    print("rows %d d %d %s: %.2f ms (%.1f TFLOP/s on 2 n d^2) [%s]" % (n, d, "exact" if exact else "float-tolerance", ms, 2.0 * n * d * d / ms / 1e9,
                                                                    ra.launch_log(reset=True)), flush=True)
