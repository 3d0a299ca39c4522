#!/usr/bin/env python3
"""k_reconstruct launch time over the number of code rows, next to torch's fill_ of the same output buffer (the
plain streaming-store rate of that very allocation).  usage: [PQHIP_DEBUG_REC_WGS=k] python tools/rec_rows_sweep.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd
M, K, dsub = 15, 256, 20
d = M * dsub
pq = reductive_amd.Pq(None, synth.normalish(43, (M, K, dsub)))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    ms.sort(); return ms[len(ms) // 2]
g = torch.Generator(device="cuda").manual_seed(42)
for rows in [int(x) for x in (sys.argv[1:] or "2000000 5000000 10000000 20000000 30000000 50000000 70000000 100000000".split())]:
    src = torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8, generator=g)
    dst = torch.empty((rows, d), device="cuda", dtype=torch.float32)
    rec = timed(lambda: pq.reconstruct_batch_device(src, out=dst, check=False))
    flat = dst.view(-1)
    parts = [flat[i:i + (1 << 30)] for i in range(0, flat.numel(), 1 << 30)]       # 4 GB pieces: 32-bit indexing in torch's kernel
    def fill():
        for p in parts: p.fill_(1.0)
    fl = timed(fill, 5)
    print(json.dumps({"rows": rows, "reconstruct_ms": round(rec, 3), "reconstruct_TBps": round(rows * 1215 / rec / 1e9, 3),
                      "fill_ms": round(fl, 3), "fill_TBps": round(rows * 1200 / fl / 1e9, 3)}), flush=True)
    del src, dst, flat, parts; torch.cuda.empty_cache()
