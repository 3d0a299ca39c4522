#!/usr/bin/env python3
"""Why does the SAME k_reconstruct launch over 100 M codes take 18.7 ms in tools/box_calibration.py and 21.1 ms in
bench.py's configs[3] leg on the same box?  One process, one codebook, the launch timed (HIP events, 12 launches,
median) under different histories of the process's device memory:
  fresh      first big allocation of the process
  again      buffers freed (cache emptied) and allocated again
  after_small  after a 12 GB + 3 x 4 GB allocate / free cycle (what bench.py's encode / OPQ legs leave behind)
  held       while 16 GB of other allocations stay alive (the library's scratch pool, cached blocks)
  offset_k   output placed k x 64 KiB further into one oversized allocation (placement / channel-hash sensitivity)
usage: python tools/rec_variance.py > gpurun_out/<tag>/rec_variance.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd

M, K, dsub, rows = 15, 256, 20, 100_000_000
d = M * dsub
pq = reductive_amd.Pq(None, synth.normalish(43, (M, K, dsub)))


def timed(fn, reps=12):
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    return sorted(ms)


def codes():
    g = torch.Generator(device="cuda").manual_seed(42)
    return torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8, generator=g)


def leg(name, dst, src, note=None):
    pq.reconstruct_batch_device(src, out=dst, check=False); torch.cuda.synchronize()
    ms = timed(lambda: pq.reconstruct_batch_device(src, out=dst, check=False))
    print(json.dumps({"leg": name, "median_ms": round(ms[len(ms) // 2], 3), "min_ms": round(ms[0], 3), "max_ms": round(ms[-1], 3),
                      "TBps": round(rows * 1215 / ms[len(ms) // 2] / 1e9, 3), "dst_ptr": hex(dst.data_ptr()), "src_ptr": hex(src.data_ptr()),
                      "note": note}), flush=True)


src = codes()
dst = torch.empty((rows, d), device="cuda", dtype=torch.float32)
leg("fresh", dst, src)
del dst, src; torch.cuda.empty_cache()
src = codes(); dst = torch.empty((rows, d), device="cuda", dtype=torch.float32)
leg("again", dst, src)
del dst, src; torch.cuda.empty_cache()

junk = [torch.empty(12 << 30, dtype=torch.uint8, device="cuda")] + [torch.empty(4 << 30, dtype=torch.uint8, device="cuda") for _ in range(3)]
for j in junk:
    j.zero_()
torch.cuda.synchronize()
del junk; torch.cuda.empty_cache()
src = codes(); dst = torch.empty((rows, d), device="cuda", dtype=torch.float32)
leg("after_small", dst, src)
del dst, src; torch.cuda.empty_cache()

held = [torch.empty(4 << 30, dtype=torch.uint8, device="cuda") for _ in range(4)]
for j in held:
    j.zero_()
src = codes(); dst = torch.empty((rows, d), device="cuda", dtype=torch.float32)
leg("held", dst, src, "16 GB of other allocations alive")
del dst, src, held; torch.cuda.empty_cache()

# codes allocated AFTER the output (the other order of the two buffers in the address space)
dst = torch.empty((rows, d), device="cuda", dtype=torch.float32); src = codes()
leg("dst_first", dst, src)
del dst, src; torch.cuda.empty_cache()

src = codes()
big = torch.empty(rows * d + (1 << 22), device="cuda", dtype=torch.float32)
for kk in (0, 1, 3, 8, 33):
    off = kk * (64 << 10) // 4
    leg("offset_%d" % kk, big[off:off + rows * d].view(rows, d), src, "output starts %d x 64 KiB into the allocation" % kk)
