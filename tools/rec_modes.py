#!/usr/bin/env python3
"""Driver for counter passes over the slow / fast placements of k_reconstruct (see tools/rec_variance2.py): for each
row count two outputs A and B alive at once, 4 launches into each, launch times to stdout (dispatch order = order here).
usage: python tools/rec_modes.py rows [rows ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd
M, K, dsub = 15, 256, 20
d = M * dsub
pq = reductive_amd.Pq(None, synth.normalish(43, (M, K, dsub)))
g = torch.Generator(device="cuda").manual_seed(42)
for rows in [int(x) for x in sys.argv[1:]]:
    src = torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8, generator=g)
    a = torch.empty((rows, d), device="cuda", dtype=torch.float32)
    b = torch.empty((rows, d), device="cuda", dtype=torch.float32)
    for name, dst in (("A", a), ("B", b)):
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); pq.reconstruct_batch_device(src, out=dst, check=False); e1.record(); torch.cuda.synchronize()
            ms.append(round(e0.elapsed_time(e1), 3))
        print(json.dumps({"rows": rows, "alloc": name, "ms": ms}), flush=True)
    del src, a, b; torch.cuda.empty_cache()
