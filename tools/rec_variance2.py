#!/usr/bin/env python3
"""Follow-up of tools/rec_variance.py (same launch, 21.0 vs 18.8 ms depending on WHICH allocation the output is):
two 120 GB outputs alive at once, and for every allocation the plain store rate (torch fill_), a plain load rate
(torch sum over a view) and the reconstruct launch, so that a placement effect of the memory system shows in all
three and a kernel effect only in the last.
usage: python tools/rec_variance2.py > gpurun_out/<tag>/rv2.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd

M, K, dsub, rows = 15, 256, 20, 100_000_000
d = M * dsub
pq = reductive_amd.Pq(None, synth.normalish(43, (M, K, dsub)))


def timed(fn, reps=8):
    ms = []
    fn(); torch.cuda.synchronize()
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2]


g = torch.Generator(device="cuda").manual_seed(42)
src = torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8, generator=g)


def measure(name, dst):
    flat = dst.view(-1)
    nb = flat.numel() * 4
    rec = timed(lambda: pq.reconstruct_batch_device(src, out=dst, check=False))
    fill = timed(lambda: flat.fill_(1.0), 5)
    q = flat[: flat.numel() // 4]
    rd = timed(lambda: q.sum(), 5)
    print(json.dumps({"alloc": name, "ptr": hex(dst.data_ptr()), "reconstruct_ms": round(rec, 3), "reconstruct_TBps": round(rows * 1215 / rec / 1e9, 3),
                      "fill_TBps": round(nb / fill / 1e9, 3), "sum_30GB_TBps": round(nb / 4 / rd / 1e9, 3)}), flush=True)


a = torch.empty((rows, d), device="cuda", dtype=torch.float32)
b = torch.empty((rows, d), device="cuda", dtype=torch.float32)
measure("A (1st, B alive)", a)
measure("B (2nd, A alive)", b)
measure("A again", a)
del a; torch.cuda.empty_cache()
c = torch.empty((rows, d), device="cuda", dtype=torch.float32)
measure("C (after A freed, B alive)", c)
del b; torch.cuda.empty_cache()
e = torch.empty((rows, d), device="cuda", dtype=torch.float32)
measure("E (after B freed, C alive)", e)
del c, e; torch.cuda.empty_cache()
# halves of one 240 GB allocation
big = torch.empty((2 * rows, d), device="cuda", dtype=torch.float32)
measure("big[:half]", big[:rows])
measure("big[half:]", big[rows:])
print(json.dumps({"mem_get_info": torch.cuda.mem_get_info()}))
