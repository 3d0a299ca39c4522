#!/usr/bin/env python3
"""Time of the rotation kernel alone (pqhip_rotate_f32_dev: out = x . P) for a row count, 10 launches (HIP events).
usage: python tools/rot_time.py [rows] [d]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_179_648
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
P = synth.orthonormal(44, d)
x = torch.randn((rows, d), device="cuda")
reductive_amd.rotate(x, P); torch.cuda.synchronize()
ms = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = reductive_amd.rotate(x, P); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    del out
ms.sort()
print("rows %d d %d: median %.3f ms, min %.3f ms, %.1f TFLOP/s on 2 d^2" % (rows, d, ms[5], ms[0], 2.0 * d * d * rows / ms[5] / 1e9))
